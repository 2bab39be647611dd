#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06d
O=gpurun_out/r06d
timeout 2400 python3 -m pytest tests/test_gpu_aggregate.py tests/test_gpu_callers.py tests/test_gpu_multi.py tests/test_gpu_fuzz.py tests/test_gpu_sharded.py -x -q 2>&1 | tail -15 > $O/pytest.txt
timeout 300 python3 scratch/longw_kinds.py strict > $O/longw_kinds_strict.txt 2>&1
timeout 300 python3 scratch/callers_wall.py > $O/callers_wall.txt 2>&1
SWEEP_ROWS=128,144,160,192,224,256 SWEEP_ROUTES=0 timeout 600 python3 scratch/midw_sweep.py dense > $O/midw_band_auto.txt 2>&1
tail -n 40 $O/*.txt
