#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for m in t512 t384; do
BOWGPU_LIB=$PWD/scratch/bin/libbowgpu_$m.so timeout -s KILL 200 python bench.py --no-cpu --no-pinned --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$m  kernel %.3f ms  frac %.3f  rw ceiling %.3f ms  ratio %.3f  ms/step %.3f' % (r['kernel_ms'], r['frac'], r['stream_rw_ceiling']['ms'], r['stream_rw_ceiling']['frac_of_ceiling'], d['ms_per_step']))"
done; done
BOWGPU_LIB=$PWD/scratch/bin/libbowgpu_t384.so timeout -s KILL 300 python -m pytest tests/test_gpu_aggregate.py -m gpu -q -x 2>&1 | tail -2
