#!/bin/bash
# A/B of rolling_fused_kernel builds (bow_amd/libbowgpu_v*.so) on the configs[2] pipeline: wall / kernel time and the HBM read counter
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for V in "$@"; do
  echo "== $V"
  BOWGPU_LIB=$GRAFT_REPO_ROOT/bow_amd/libbowgpu_$V.so timeout -s KILL 200 python3 scratch/cfg2_fused.py 1e8 quick 2>&1 | grep -v "^[WE]2026"
  export BOWGPU_LIB=$GRAFT_REPO_ROOT/bow_amd/libbowgpu_$V.so
  for CNT in FETCH_SIZE WRITE_SIZE; do
    rm -rf "gpurun_out/fab_${V}_${CNT}"
    timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "gpurun_out/fab_${V}_${CNT}" -- python3 scratch/cfg2_fused.py 1e8 quick > /dev/null 2>&1
    python3 - <<PY
import csv, glob
v = [float(r["Counter_Value"]) for f in glob.glob("gpurun_out/fab_${V}_$CNT/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "rolling_fused" in r["Kernel_Name"]]
print("$V $CNT rolling_fused_kernel: %d launches, avg %.3f GB%s" % (len(v), (sum(v) / max(len(v), 1)) * 1024 / 1e9 * (2 if "$CNT" == "FETCH_SIZE" else 1), " (x2 applied)" if "$CNT" == "FETCH_SIZE" else ""))
PY
  done
  unset BOWGPU_LIB
done
