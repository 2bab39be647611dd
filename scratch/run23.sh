#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -s KILL 600 python -m pytest tests/test_gpu_aggregate.py -m gpu -q -x 2>&1 | grep -v "^  File \"/usr" | tail -3
timeout -s KILL 300 python scratch/general_bench.py 2>&1 | grep -E "simple kernel|Factor|ns timestamps, 1 s windows: Mean"
timeout -s KILL 300 python scratch/configs.py 2>&1 | grep -E "cfg1|cfg2 sparse|cfg3"
timeout -s KILL 300 python bench.py --no-cpu --no-pinned 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('bench value %.1f G rows/s  ms/step %.3f  kernel %.3f ms  frac %.3f  rw ceiling ms %.3f  ratio %.3f' % (d['value']/1e9, d['ms_per_step'], r['kernel_ms'], r['frac'], r['stream_rw_ceiling']['ms'], r['stream_rw_ceiling']['frac_of_ceiling']))"
