#!/bin/bash
# round 6, second batch: tests of what changed since batch a, the band again, strict walks, counters of the kernels that had none
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
timeout 1500 python3 -m pytest tests/test_gpu_multi.py tests/test_gpu_callers.py tests/test_gpu_aggregate.py tests/test_gpu_sharded.py -x -q 2>&1 | tail -15 > $O/pytest.txt
SWEEP_ROWS=128,144,160,192,224,256 SWEEP_ROUTES=0 timeout 600 python3 scratch/midw_sweep.py > $O/midw_band_auto.txt 2>&1
timeout 300 python3 scratch/whole_wall.py > $O/whole_wall.txt 2>&1
timeout 300 python3 scratch/longw_kinds.py strict > $O/longw_kinds_strict.txt 2>&1
bash scratch/pmc_any.sh fused_off0 rolling_fused scratch/fused_one.py 1e8 0 2 > $O/pmc_fused_off0.txt 2>&1
bash scratch/pmc_any.sh fused_off7 rolling_fused scratch/fused_one.py 1e8 7 2 > $O/pmc_fused_off7.txt 2>&1
bash scratch/pmc_any.sh fused7_off0 rolling_fused scratch/fused_one.py 1e8 0 7 > $O/pmc_fused7_off0.txt 2>&1
bash scratch/pmc_any.sh plain_off0 rolling_simple scratch/fused_one.py 1e8 0 2 plain > $O/pmc_plain_off0.txt 2>&1
bash scratch/pmc_any.sh strict_tw long_strict scratch/longw_one.py tw dense strict > $O/pmc_strict_tw.txt 2>&1
bash scratch/pmc_any.sh strict_mean long_strict scratch/longw_one.py mean dense strict > $O/pmc_strict_mean.txt 2>&1
tail -n 60 $O/*.txt
