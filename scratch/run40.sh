#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 600 python -m pytest tests/test_gpu_aggregate.py tests/test_gpu_callers.py -m gpu -q -x 2>&1 | tail -2
timeout -s KILL 300 python bench.py --no-cpu --no-pinned 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('bench value %.1f G rows/s  ms/step %.3f  kernel %.3f ms  overhead %.0f us' % (d['value']/1e9, d['ms_per_step'], r['kernel_ms'], (d['ms_per_step']-r['kernel_ms'])*1e3))"
timeout -s KILL 300 python scratch/interp_wall.py 2>&1 | head -1
rm -rf gpurun_out/r2y; timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2y -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-pinned > /dev/null 2>&1
grep -h "finish_bitmaps\|preset_bitmaps" gpurun_out/r2y/*/*kernel_stats.csv | cut -c1-120
