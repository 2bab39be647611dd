#!/bin/bash
# round 5, first GPU call: the tests of the files touched so far, the same-box A/B of the benched instantiation with and without
# the staging pads (BOWGPU_ROUTE=256 = BOWGPU_ROUTE_SIMPLE_PADDED), then the counter collection of scratch/profile_bench.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests/test_gpu_aggregate.py tests/test_gpu_callers.py -x -q -m gpu > gpurun_out/r05a_tests.log 2>&1
tail -3 gpurun_out/r05a_tests.log
for i in 1 2 3 4; do
  timeout -s KILL 200 python bench.py --steps 20 --warmup 3 --no-cpu --no-pinned --no-parity > gpurun_out/ab_plain_$i.json 2>/dev/null
  BOWGPU_ROUTE=256 timeout -s KILL 200 python bench.py --steps 20 --warmup 3 --no-cpu --no-pinned --no-parity > gpurun_out/ab_padded_$i.json 2>/dev/null
done
python - <<'PY'
import json, glob
for k in ("plain", "padded"):
    for f in sorted(glob.glob("gpurun_out/ab_%s_*.json" % k)):
        try:
            r = json.load(open(f))
            print(k, r["roofline"]["kernel_instance"], "kernel %.4f ms  step %.4f ms  %.1f G rows/s  frac %.4f" % (r["roofline"]["kernel_ms"], r["ms_per_step"], r["value"] / 1e9, r["roofline"]["frac"]))
        except Exception as e:
            print(k, f, "failed", e)
PY
bash scratch/profile_bench.sh r05 > gpurun_out/r05_profile_bench.log 2>&1
tail -40 gpurun_out/r05_profile_bench.log
