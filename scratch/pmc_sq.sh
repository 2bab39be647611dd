#!/bin/bash
# usage: scratch/pmc_sq.sh <tag> <kernel substring> <python script...> : SQ / LDS / L2 counters of one kernel (no FETCH/WRITE passes), each pass under a hard limit
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; KSUB=$2; shift; shift
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH SQ_IFETCH" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL ${PMC_LIMIT:-150} rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python3 "$@" > $OUT/p$i.log 2>&1 || echo "pass $i: rc=$?"
done
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(list)
dur = []
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KSUB" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$OUT/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KSUB" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("kernel %s: n=%d median %.4f ms" % ("$KSUB", len(dur), sorted(dur)[len(dur) // 2] if dur else 0))
for k, v in sorted(acc.items()):
    print("%-26s n=%d avg=%.5g" % (k, len(v), sum(v) / len(v)))
PY
