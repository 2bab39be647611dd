#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2r
( time timeout 1500 python -m pytest tests/test_gpu_aggregate.py tests/test_gpu_fuzz.py tests/test_gpu_sharded.py -m gpu -q -x 2>&1 | tail -8 ) > gpurun_out/r2r/pytest.txt 2>&1
tail -8 gpurun_out/r2r/pytest.txt
python scratch/general_bench.py > gpurun_out/r2r/general_bench.txt 2>&1; cat gpurun_out/r2r/general_bench.txt
python scratch/longw.py > gpurun_out/r2r/longw.txt 2>&1; cat gpurun_out/r2r/longw.txt
python scratch/configs.py > gpurun_out/r2r/configs.txt 2>&1; cat gpurun_out/r2r/configs.txt
