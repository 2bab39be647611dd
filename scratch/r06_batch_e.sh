#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06e
O=gpurun_out/r06e
timeout 1500 python3 -m pytest tests/test_gpu_aggregate.py -x -q 2>&1 | tail -15 > $O/pytest.txt
for V in "" nocoop coop16; do
  L=$GRAFT_REPO_ROOT/bow_amd/libbowgpu${V:+_$V}.so
  echo "== ${V:-product}" >> $O/midw_mm.txt
  BOWGPU_LIB=$L SWEEP_ROWS=32,48,64,96,128,144,160,192 SWEEP_ROUTES=0 timeout 600 python3 scratch/midw_sweep.py MinMax SumMinMax >> $O/midw_mm.txt 2>&1
done
tail -n 80 $O/*.txt
