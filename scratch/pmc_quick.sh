#!/bin/bash
# usage: scratch/pmc_quick.sh <tag> <kernel substring> <python script...> : ONE counter pass (instruction mix + LDS) of one kernel, per-wave averages
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; KSUB=$2; shift; shift
OUT=gpurun_out/pmcq_$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL ${PMC_LIMIT:-100} rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p -- python3 "$@" > $OUT/p.log 2>&1
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KSUB" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = sum(acc["SQ_WAVES"]) / max(len(acc["SQ_WAVES"]), 1)
print("$TAG: kernel $KSUB, %d launches, %.0f waves; per wave:" % (len(acc["SQ_WAVES"]), w), " ".join("%s=%.0f" % (k.replace("SQ_", ""), sum(v) / len(v) / w) for k, v in sorted(acc.items()) if k != "SQ_WAVES"))
PY
