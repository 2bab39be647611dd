#!/bin/bash
# round 6, first measurement batch: the 129 .. 255 band (device queue vs host queue vs streaming form), the callers, configs[2] fused (padded A/B), strict long windows
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
O=gpurun_out/r06a
SWEEP_ROWS=96,128,144,160,192,224,256,320 SWEEP_HOSTQ=1 timeout 900 python3 scratch/midw_sweep.py > $O/midw_band.txt 2>&1
timeout 300 python3 scratch/whole_wall.py > $O/whole_wall.txt 2>&1
timeout 300 python3 scratch/callers_wall.py > $O/callers_wall.txt 2>&1
timeout 300 python3 scratch/cfg2_fused.py 1e8 > $O/cfg2_fused.txt 2>&1
BOWGPU_LIB=$GRAFT_REPO_ROOT/bow_amd/libbowgpu_fswz.so timeout 300 python3 scratch/cfg2_fused.py 1e8 > $O/cfg2_fused_swz.txt 2>&1
timeout 300 python3 scratch/longw_kinds.py strict > $O/longw_kinds_strict.txt 2>&1
timeout 300 python3 scratch/small_calls.py > $O/small_calls.txt 2> $O/small_calls_err.txt
tail -n 100 $O/*.txt
