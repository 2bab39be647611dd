#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2u
timeout 900 python -m pytest tests/test_gpu_aggregate.py -m gpu -q -x > gpurun_out/r2u/agg.txt 2>&1
echo "agg rc=$?"; grep -v "^  File \"/usr" gpurun_out/r2u/agg.txt | tail -5
timeout 300 python scratch/longw.py > gpurun_out/r2u/longw.txt 2>&1; cat gpurun_out/r2u/longw.txt
rm -rf gpurun_out/r2u/prof; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2u/prof -- python3 scratch/longw.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/r2u/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-60s calls %4s avg %9.1f us min %9.1f max %9.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
