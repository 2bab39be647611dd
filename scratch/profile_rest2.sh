#!/bin/bash
# Round 6, third call of the collection: what the second call (scratch/profile_rest.sh) did not reach - on its box rocprofv3 hung at the
# exit of EVERY pass and each one ran into its 150 s limit (the counters were written before the hang: the files it produced are whole).
# Hard limit per pass 60 s here, one counter pass per callers' kernel.
TAG=${1:-r06}
export PMC_LIMIT=60
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
DST=gpurun_out/profiles_${TAG}c
rm -rf $DST && mkdir -p $DST
(git rev-parse HEAD 2>/dev/null || cat scratch/HEAD_COMMIT 2>/dev/null || echo "(snapshot without .git)") > $DST/${TAG}_commit.txt
timeout -s KILL 400 python3 scratch/multi_wall.py 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_multi_wall.txt
CFG2_ORDER=rev timeout -s KILL 300 python3 scratch/cfg2_fused.py 1e8 quick 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_cfg2_fused_rev.txt
timeout -s KILL 600 python3 scratch/midw_sweep.py 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_midw_sweep.txt
SWEEP_ROWS=128,144,160,192,224,256 SWEEP_HOSTQ=1 timeout -s KILL 700 python3 scratch/midw_sweep.py 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_midw_band.txt
timeout -s KILL 300 python3 scratch/configs.py 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_configs.txt
if [ -f bow_amd/libbowgpu_twpad.so ]; then
  (echo "== product build (pads only where both kinds of integral are walked)"
   SWEEP_ROWS=16,32,64,96,128 SWEEP_ROUTES=0 timeout -s KILL 300 python3 scratch/midw_sweep.py dense WAvgStep 2>&1 | grep -v "^[WE]2026"
   echo "== -DBOWGPU_TW_LEAN_PAD=1 (pads for one kind too)"
   BOWGPU_LIB=$GRAFT_REPO_ROOT/bow_amd/libbowgpu_twpad.so SWEEP_ROWS=16,32,64,96,128 SWEEP_ROUTES=0 timeout -s KILL 300 python3 scratch/midw_sweep.py dense WAvgStep 2>&1 | grep -v "^[WE]2026"
   echo "== counters, 64 rows per window: product, then padded"
   bash scratch/pmc_quick.sh WAvgStep_64_dense_product rolling_tw scratch/one_shape.py gen WAvgStep 64 dense | tail -1
   BOWGPU_LIB=$GRAFT_REPO_ROOT/bow_amd/libbowgpu_twpad.so bash scratch/pmc_quick.sh WAvgStep_64_dense_padded rolling_tw scratch/one_shape.py gen WAvgStep 64 dense | tail -1) > $DST/${TAG}_stdout_tw_lean_pad_ab.txt
fi
: > $DST/${TAG}_pmc_mid_windows.txt
for V in "WAvgStep 64 dense rolling_tw" "TW4 64 dense rolling_tw" "WAvgStep 64 sparse rolling_twc" "TW4 64 sparse rolling_twc" "TW4 96 sparse rolling_twc" "WAvgStep 192 sparse rolling_twc" "Mean 64 dense rolling_simple" "SumMinMax 64 sparse rolling_twc" "MinMax 64 sparse rolling_simple" "MinMax 128 dense rolling_simple" "SumMinMax 224 dense long_short"; do
  set -- $V
  bash scratch/pmc_quick.sh $1_$2_$3 $4 scratch/one_shape.py gen $1 $2 $3 | tail -1 >> $DST/${TAG}_pmc_mid_windows.txt
done
: > $DST/${TAG}_pmc_callers.txt
for K in whole_value col_order_dense fill_kernel; do
  bash scratch/pmc_quick.sh callers_$K $K scratch/callers_one.py | tail -1 >> $DST/${TAG}_pmc_callers.txt
done
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_callers_$C
  timeout -s KILL 60 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_callers_$C -- python3 scratch/callers_one.py > /dev/null 2>&1
  python3 - <<PY >> $DST/${TAG}_pmc_callers.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_callers_$C/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("whole_value", "col_order_dense", "col_order_kernel", "fill_kernel"):
            if k in r["Kernel_Name"]:
                acc[k + ("<Previous>" if k == "fill_kernel" and "0, " in r["Kernel_Name"].split("fill_kernel")[1][:6] else "")].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("$C %-28s n=%d avg=%.5g KB per launch%s" % (k, len(v), sum(v) / len(v), " (x 2 for bytes read: MI355X_MICROARCH.md)" if "$C" == "FETCH_SIZE" else ""))
PY
done
ls -la $DST
