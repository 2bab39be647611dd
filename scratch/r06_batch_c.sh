#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
O=gpurun_out/r06c
timeout 2400 python3 -m pytest tests/test_gpu_aggregate.py tests/test_gpu_callers.py tests/test_gpu_multi.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -15 > $O/pytest.txt
timeout 300 python3 scratch/longw_kinds.py strict > $O/longw_kinds_strict.txt 2>&1
timeout 300 python3 scratch/callers_wall.py > $O/callers_wall.txt 2>&1
timeout 300 python3 scratch/whole_wall.py > $O/whole_wall.txt 2>&1
timeout 300 python3 scratch/cfg2_fused.py 1e8 quick > $O/cfg2_fused_fwd.txt 2>&1
CFG2_ORDER=rev timeout 300 python3 scratch/cfg2_fused.py 1e8 quick > $O/cfg2_fused_rev.txt 2>&1
tail -n 40 $O/*.txt
