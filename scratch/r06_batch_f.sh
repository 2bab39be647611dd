#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06f
O=gpurun_out/r06f
timeout 300 python3 scratch/longw_kinds.py strict > $O/longw_kinds_strict.txt 2>&1
timeout 900 python3 -m pytest tests/test_gpu_aggregate.py tests/test_gpu_sharded.py -x -q -p no:cacheprovider 2>&1 | tail -4 > $O/pytest.txt
SWEEP_ROWS=144,160,192 SWEEP_ROUTES=0 timeout 300 python3 scratch/midw_sweep.py MinMax FirstLast WAvgStep 2>&1 | grep -v "^[WE]2026" | cut -c1-120 > $O/band.txt
cat $O/*.txt
