#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2o
( time timeout 1500 python -m pytest tests/test_gpu_callers.py tests/test_gpu_sharded.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -q -x -k "interp or Interp" 2>&1 | tail -25 ) > gpurun_out/r2o/pytest.txt 2>&1
tail -12 gpurun_out/r2o/pytest.txt
python scratch/interp_wall.py > gpurun_out/r2o/interp_wall.txt 2>&1; cat gpurun_out/r2o/interp_wall.txt
bash scratch/pmc_any.sh r2o_wave2 interp_wave2 scratch/interp_only.py 2>&1 | tail -32
