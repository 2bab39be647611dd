#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2v
export BOWGPU_TRACE_ROUTE=1
timeout 1500 python -m pytest tests -m gpu -q -x -s -k "not fullsize" > gpurun_out/r2v/route.txt 2>&1
echo "rc=$?"; tail -2 gpurun_out/r2v/route.txt
grep "bowgpu route" gpurun_out/r2v/route.txt | sed 's/n=[0-9]* W=[0-9]* //; s/first_ts=.*//' | sort | uniq -c | sort -rn | head -30
python scratch/general_bench.py 2>&1 | tail -14
