#!/bin/bash
# usage: scratch/prof_any.sh <tag> <python script> : rocprofv3 kernel stats of a script -> gpurun_out/prof_<tag>/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 "$@" > $OUT/run.log 2>&1
cat $OUT/run.log | tail -20
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-90s calls=%-5s avg=%10.1f us total=%10.1f us %s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3, r["Percentage"]))
PY
