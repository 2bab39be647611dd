#!/bin/bash
# usage: scratch/ab_pmc.sh <tag> <rows> <variant filter> : kernel stats + PMC passes of scratch/bin/headline_ab, one table per kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; ROWS=$2; FILT=$3
OUT=gpurun_out/abpmc_$TAG
rm -rf $OUT && mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- scratch/bin/headline_ab $ROWS 5 $FILT > $OUT/stats.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
         "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
         "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- scratch/bin/headline_ab $ROWS 2 $FILT > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "tile_kernel" in k or "rw_ceiling" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "tile_kernel" in k or "rw_ceiling" in k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k in sorted(acc):
    d = sorted(dur.get(k, [0]))
    print("== %s   n=%d min %.4f ms median %.4f ms" % (k, len(d), d[0], d[len(d) // 2]))
    for c, v in sorted(acc[k].items()):
        print("   %-30s n=%d avg=%.6g" % (c, len(v), sum(v) / len(v)))
PY
