#!/bin/bash
# Every profiles/<tag>_* file of a round from ONE script at ONE commit (run on the GPU box through gpurun):
#   scratch/profile_all.sh r04     -> gpurun_out/profiles_r04/  (copy its files to profiles/ as they are)
# 1. scratch/profile_bench.sh   : the bench line un-profiled, rocprofv3 --kernel-trace --stats, the FETCH_SIZE / WRITE_SIZE passes and
#                                 one SQ counter pass of the benched kernel (traffic rows carry the kernel signature + source hash)
# 2. the secondary shapes, un-profiled stdout tables (the scripts print kernel brackets from HIP events and wall times)
# 3. scratch/pmc_quick.sh       : one counter pass (instruction mix, LDS) per mid-window shape
# (--pmc passes never carry another trace domain; every profiled process is python3 itself: no env / shell hop after `--`.  Every
# rocprofv3 pass runs under a hard limit: the profiler now and then hangs at process exit on this pool.)
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
DST=gpurun_out/profiles_$TAG
rm -rf $DST && mkdir -p $DST
(git rev-parse HEAD 2>/dev/null || cat scratch/HEAD_COMMIT 2>/dev/null || echo "(snapshot without .git)") > $DST/${TAG}_commit.txt
bash scratch/profile_bench.sh $TAG > $DST/profile_bench.log 2>&1
cp gpurun_out/prof/${TAG}_* $DST/ 2>/dev/null
cp gpurun_out/prof/bench.json $DST/${TAG}_bench_1e9.json 2>/dev/null
for name in configs general_bench interp_wall fill_wall whole_wall callers_wall longw_kinds midw_sweep small_calls host_resident; do
  PROF=0; [ "$name" = small_calls ] && PROF=1
  BOWGPU_CALL_PROFILE=$PROF timeout -s KILL 600 python3 scratch/$name.py 2> $DST/${TAG}_stderr_${name}.tmp | grep -v "^[WE]2026" > $DST/${TAG}_stdout_${name}.txt
  # (the call profiler's lines - stderr - go to their own file, not into the table)
  grep "^bowgpu call profile" $DST/${TAG}_stderr_${name}.tmp > $DST/${TAG}_stderr_${name}_call_profile.txt; [ -s $DST/${TAG}_stderr_${name}_call_profile.txt ] || rm -f $DST/${TAG}_stderr_${name}_call_profile.txt
  rm -f $DST/${TAG}_stderr_${name}.tmp
done
timeout -s KILL 300 python3 scratch/longw_kinds.py strict 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_longw_kinds_strict.txt
timeout -s KILL 600 python3 scratch/longw_sweep.py 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_longw_sweep.txt
: > $DST/${TAG}_pmc_mid_windows.txt
for V in "WAvgStep 64 dense rolling_tw" "TW4 64 dense rolling_tw" "WAvgStep 64 sparse rolling_twc" "TW4 64 sparse rolling_twc" "TW4 96 sparse rolling_twc" "WAvgStep 192 sparse rolling_twc" "Mean 64 dense rolling_simple" "SumMinMax 64 sparse rolling_twc" "MinMax 64 sparse rolling_simple"; do
  set -- $V
  bash scratch/pmc_quick.sh $1_$2_$3 $4 scratch/one_shape.py gen $1 $2 $3 | tail -1 >> $DST/${TAG}_pmc_mid_windows.txt
done
# 4. the streaming form's slowest instantiation next to its dense twin: which unit is busy (a counter-backed floor)
: > $DST/${TAG}_pmc_long_short_tw.txt
for V in "WeightedAverageStep sparse" "WeightedAverageStep dense" "Mean sparse"; do
  set -- $V
  echo "== $1 $2, 1000 rows per window" >> $DST/${TAG}_pmc_long_short_tw.txt
  bash scratch/pmc_sq.sh lstw_$1_$2 long_short scratch/longw_pmc.py $1 $2 | grep -v "^pass" >> $DST/${TAG}_pmc_long_short_tw.txt
done
# 4b. Interpolate: the launches of a call (at most 5) and the fill kernel's counters
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/iw_$TAG -o iw -- python3 scratch/interp_wall.py 1e8 one > /dev/null 2>&1
cp $(find gpurun_out/iw_$TAG -name "*kernel_stats.csv" | head -1) $DST/${TAG}_kernel_stats_interp_wall.csv 2>/dev/null
bash scratch/pmc_sq.sh iw3_$TAG interp_wave3 scratch/interp_wall.py 1e8 one | grep -v "^pass" > $DST/${TAG}_pmc_interp_wave3_1e8.txt
# 4c. ... and the same call through the diagnostic build without the kernel's run pass (built beforehand, travels with the snapshot:
#     scratch/build_variant.sh xruns interpolate.hip -DBOWGPU_X_SKIP_RUNS): what the pass costs = how far the kernel is from its traffic
if [ -f bow_amd/libbowgpu_xruns.so ]; then
  (echo "== product build"; timeout -s KILL 200 python3 scratch/interp_wall.py 1e8 one 2>&1 | grep -v "^[WE]2026"
   echo "== diagnostic build, no run pass (outputs wrong, same bytes moved)"; BOWGPU_LIB=bow_amd/libbowgpu_xruns.so timeout -s KILL 200 python3 scratch/interp_wall.py 1e8 one 2>&1 | grep -v "^[WE]2026") > $DST/${TAG}_stdout_interp_skip_runs.txt
fi
[ -f bow_amd/libbowgpu_stamps.so ] && BOWGPU_LIB=bow_amd/libbowgpu_stamps.so timeout -s KILL 200 python3 scratch/interp_stamps.py 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_interp_stamps.txt
# 5. the configs[2] pipeline as one call against the two calls; the fused kernel's traffic
timeout -s KILL 300 python3 scratch/cfg2_fused.py 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_cfg2_fused.txt
# 6. round 6: counters of the kernels README quotes a figure for and that had no counter file: the fused kernel at both
#    offsets next to the plain sparse Mean, the strict lane walks, the queue walk behind a tile pass, the streaming form in the band, the callers
: > $DST/${TAG}_pmc_rolling_fused.txt
for V in "0 2" "7 2" "0 7" "7 7"; do
  set -- $V
  echo "== Interpolate -> Aggregate, offset $1, $2 reducers (configs[2]: 1e8 irregular rows, 30 % nulls, interval 100)" >> $DST/${TAG}_pmc_rolling_fused.txt
  bash scratch/pmc_sq.sh fused_$1_$2 rolling_fused scratch/fused_one.py 1e8 $1 $2 | grep -v "^pass" >> $DST/${TAG}_pmc_rolling_fused.txt
done
echo "== the same rows through bowgpu_rolling_aggregate (no interpolation): rolling_simple_kernel, Mean, offset 0" >> $DST/${TAG}_pmc_rolling_fused.txt
bash scratch/pmc_sq.sh fused_plain rolling_simple scratch/fused_one.py 1e8 0 2 plain | grep -v "^pass" >> $DST/${TAG}_pmc_rolling_fused.txt
: > $DST/${TAG}_pmc_long_strict.txt
for V in "mean dense" "tw dense" "tw sparse" "tw4 dense"; do
  set -- $V
  echo "== strict_order, 1000-row windows, $1 $2" >> $DST/${TAG}_pmc_long_strict.txt
  bash scratch/pmc_sq.sh strict_$1_$2 long_strict scratch/longw_one.py $1 $2 strict | grep -v "^pass" >> $DST/${TAG}_pmc_long_strict.txt
done
: > $DST/${TAG}_pmc_band.txt
for V in "MinMax 160 dense long_queue" "MinMax 160 dense rolling_simple" "MinMax 224 dense long_short" "SumMinMax 224 dense long_short" "MinMax 224 dense stream_final"; do
  set -- $V
  echo "== $1, $2 rows per window, $3: $4" >> $DST/${TAG}_pmc_band.txt
  bash scratch/pmc_sq.sh band_$1_$2_$4 $4 scratch/one_shape.py gen $1 $2 $3 | grep -v "^pass" >> $DST/${TAG}_pmc_band.txt
done
: > $DST/${TAG}_pmc_callers.txt
for K in whole_value whole_finish col_order_dense "col_order_kernel" fill_kernel; do
  echo "== $K (scratch/callers_one.py: 1e8 rows)" >> $DST/${TAG}_pmc_callers.txt
  bash scratch/pmc_sq.sh callers_$K $K scratch/callers_one.py | grep -v "^pass" >> $DST/${TAG}_pmc_callers.txt
done
# 6b. VERDICT r05 item 4a: the lean dense form of rolling_tw_kernel with the padded term layout for ONE kind of integral as well (A/B build,
#     travels with the snapshot: scratch/build_variant.sh twpad rolling_tw.hip -DBOWGPU_TW_LEAN_PAD=1)
if [ -f bow_amd/libbowgpu_twpad.so ]; then
  (echo "== product build (pads only where both kinds of integral are walked)"
   SWEEP_ROWS=16,32,64,96,128 SWEEP_ROUTES=0 timeout -s KILL 300 python3 scratch/midw_sweep.py dense WAvgStep 2>&1 | grep -v "^[WE]2026"
   echo "== -DBOWGPU_TW_LEAN_PAD=1 (pads for one kind too)"
   BOWGPU_LIB=$GRAFT_REPO_ROOT/bow_amd/libbowgpu_twpad.so SWEEP_ROWS=16,32,64,96,128 SWEEP_ROUTES=0 timeout -s KILL 300 python3 scratch/midw_sweep.py dense WAvgStep 2>&1 | grep -v "^[WE]2026"
   echo "== counters, 64 rows per window: product, then padded"
   bash scratch/pmc_quick.sh WAvgStep_64_dense_product rolling_tw scratch/one_shape.py gen WAvgStep 64 dense | tail -1
   BOWGPU_LIB=$GRAFT_REPO_ROOT/bow_amd/libbowgpu_twpad.so bash scratch/pmc_quick.sh WAvgStep_64_dense_padded rolling_tw scratch/one_shape.py gen WAvgStep 64 dense | tail -1) > $DST/${TAG}_stdout_tw_lean_pad_ab.txt
fi
# 7. round 6: one call over the device list (the same device listed N times on a one-GPU box): wall by residency and rank count
timeout -s KILL 600 python3 scratch/multi_wall.py 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_multi_wall.txt
CFG2_ORDER=rev timeout -s KILL 300 python3 scratch/cfg2_fused.py 1e8 quick 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_cfg2_fused_rev.txt
SWEEP_ROWS=96,128,144,160,192,224,256,320 SWEEP_HOSTQ=1 timeout -s KILL 900 python3 scratch/midw_sweep.py 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_midw_band.txt
ls -la $DST
