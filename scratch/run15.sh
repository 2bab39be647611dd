#!/bin/bash
cd $GRAFT_REPO_ROOT
export BOWGPU_TRACE_ROUTE=1
timeout -s KILL 600 python -m pytest tests -m gpu -q -x -s -k "not fullsize" > gpurun_out/r2w_route.txt 2>&1
tail -1 gpurun_out/r2w_route.txt
echo "route lines: $(grep -c 'bowgpu route' gpurun_out/r2w_route.txt)"
grep "bowgpu route" gpurun_out/r2w_route.txt | sed 's/n=[0-9]* W=[0-9]* //; s/first_ts=.*//' | sort | uniq -c | sort -rn | head -8
