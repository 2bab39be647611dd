#!/bin/bash
# Round 6, second call of the collection (the first - scratch/profile_all.sh r06 - hit gpurun's limit after 58 minutes; and the tile kernel's
# extrema changed after it): the counter files that speak about rolling_simple.hip again, the ones the first call did not reach, the band.
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
DST=gpurun_out/profiles_${TAG}b
rm -rf $DST && mkdir -p $DST
(git rev-parse HEAD 2>/dev/null || cat scratch/HEAD_COMMIT 2>/dev/null || echo "(snapshot without .git)") > $DST/${TAG}_commit.txt
: > $DST/${TAG}_pmc_rolling_fused.txt
for V in "0 2" "7 2" "0 7"; do
  set -- $V
  echo "== Interpolate -> Aggregate, offset $1, $2 reducers (configs[2]: 1e8 irregular rows, 30 % nulls, interval 100)" >> $DST/${TAG}_pmc_rolling_fused.txt
  bash scratch/pmc_sq.sh fused_$1_$2 rolling_fused scratch/fused_one.py 1e8 $1 $2 | grep -v "^pass" >> $DST/${TAG}_pmc_rolling_fused.txt
done
echo "== the same rows through bowgpu_rolling_aggregate (no interpolation): rolling_simple_kernel, Mean, offset 0" >> $DST/${TAG}_pmc_rolling_fused.txt
bash scratch/pmc_sq.sh fused_plain rolling_simple scratch/fused_one.py 1e8 0 2 plain | grep -v "^pass" >> $DST/${TAG}_pmc_rolling_fused.txt
: > $DST/${TAG}_pmc_band.txt
for V in "MinMax 160 dense long_queue" "MinMax 160 dense rolling_simple" "MinMax 128 dense rolling_simple" "SumMinMax 224 dense long_short"; do
  set -- $V
  echo "== $1, $2 rows per window, $3: $4" >> $DST/${TAG}_pmc_band.txt
  bash scratch/pmc_sq.sh band_$1_$2_$4 $4 scratch/one_shape.py gen $1 $2 $3 | grep -v "^pass" >> $DST/${TAG}_pmc_band.txt
done
: > $DST/${TAG}_pmc_mid_windows.txt
for V in "WAvgStep 64 dense rolling_tw" "TW4 64 dense rolling_tw" "WAvgStep 64 sparse rolling_twc" "TW4 64 sparse rolling_twc" "TW4 96 sparse rolling_twc" "WAvgStep 192 sparse rolling_twc" "Mean 64 dense rolling_simple" "SumMinMax 64 sparse rolling_twc" "MinMax 64 sparse rolling_simple" "MinMax 128 dense rolling_simple"; do
  set -- $V
  bash scratch/pmc_quick.sh $1_$2_$3 $4 scratch/one_shape.py gen $1 $2 $3 | tail -1 >> $DST/${TAG}_pmc_mid_windows.txt
done
: > $DST/${TAG}_pmc_callers.txt
for K in whole_value col_order_dense fill_kernel; do
  echo "== $K (scratch/callers_one.py: 1e8 rows)" >> $DST/${TAG}_pmc_callers.txt
  bash scratch/pmc_sq.sh callers_$K $K scratch/callers_one.py | grep -v "^pass" >> $DST/${TAG}_pmc_callers.txt
done
timeout -s KILL 400 python3 scratch/multi_wall.py 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_multi_wall.txt
CFG2_ORDER=rev timeout -s KILL 300 python3 scratch/cfg2_fused.py 1e8 quick 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_cfg2_fused_rev.txt
timeout -s KILL 600 python3 scratch/midw_sweep.py 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_midw_sweep.txt
SWEEP_ROWS=128,144,160,192,224,256 SWEEP_HOSTQ=1 timeout -s KILL 700 python3 scratch/midw_sweep.py 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_midw_band.txt
if [ -f bow_amd/libbowgpu_twpad.so ]; then
  (echo "== product build (pads only where both kinds of integral are walked)"
   SWEEP_ROWS=16,32,64,96,128 SWEEP_ROUTES=0 timeout -s KILL 300 python3 scratch/midw_sweep.py dense WAvgStep 2>&1 | grep -v "^[WE]2026"
   echo "== -DBOWGPU_TW_LEAN_PAD=1 (pads for one kind too)"
   BOWGPU_LIB=$GRAFT_REPO_ROOT/bow_amd/libbowgpu_twpad.so SWEEP_ROWS=16,32,64,96,128 SWEEP_ROUTES=0 timeout -s KILL 300 python3 scratch/midw_sweep.py dense WAvgStep 2>&1 | grep -v "^[WE]2026"
   echo "== counters, 64 rows per window: product, then padded"
   bash scratch/pmc_quick.sh WAvgStep_64_dense_product rolling_tw scratch/one_shape.py gen WAvgStep 64 dense | tail -1
   BOWGPU_LIB=$GRAFT_REPO_ROOT/bow_amd/libbowgpu_twpad.so bash scratch/pmc_quick.sh WAvgStep_64_dense_padded rolling_tw scratch/one_shape.py gen WAvgStep 64 dense | tail -1) > $DST/${TAG}_stdout_tw_lean_pad_ab.txt
fi
timeout -s KILL 300 python3 scratch/configs.py 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_configs.txt
ls -la $DST
