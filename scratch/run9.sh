#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2t
for i in 1 2; do
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r2t/full$i.txt 2>&1
echo "run $i rc=$?"
grep -v "^  File \"/usr" gpurun_out/r2t/full$i.txt | head -40
done
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_gpu_aggregate.py -m gpu -q -x 2>&1 | tail -1; done
