#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2t
( time timeout 2000 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > gpurun_out/r2t/pytest.txt 2>&1
tail -12 gpurun_out/r2t/pytest.txt
python scratch/general_bench.py > gpurun_out/r2t/general_bench.txt 2>&1; cat gpurun_out/r2t/general_bench.txt
