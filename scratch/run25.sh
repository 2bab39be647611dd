#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 100 python3 scratch/one_shape.py mean100 | tail -1
rm -rf gpurun_out/r2x; timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2x -- python3 scratch/one_shape.py mean100 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/r2x/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-60s calls %4s avg %9.1f us tot %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
