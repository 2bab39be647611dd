#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in mean mean3 mean2; do timeout -s KILL 100 python scratch/one_shape.py $s 2>&1 | tail -1; done
timeout -s KILL 300 python scratch/general_bench.py 2>&1 | tail -11
timeout -s KILL 300 python bench.py --no-cpu --no-pinned 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('bench value %.1f G rows/s  ms/step %.3f  kernel %.3f ms  frac %.3f  rw ceiling ms %.3f  ratio %.3f' % (d['value']/1e9, d['ms_per_step'], r['kernel_ms'], r['frac'], r['stream_rw_ceiling']['ms'], r['stream_rw_ceiling']['frac_of_ceiling']))"
