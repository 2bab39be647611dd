#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for m in 400 382 254; do
BOWGPU_LIB=$PWD/scratch/bin/libbowgpu_cap$m.so timeout -s KILL 200 python bench.py --no-cpu --no-pinned --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('cap $m  kernel %.3f ms  frac %.3f  rw ceiling %.3f ms  ratio %.3f  ms/step %.3f' % (r['kernel_ms'], r['frac'], r['stream_rw_ceiling']['ms'], r['stream_rw_ceiling']['frac_of_ceiling'], d['ms_per_step']))"
done; done
