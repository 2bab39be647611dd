#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -s KILL 900 python -m pytest tests -m gpu -q -x > gpurun_out/full_final.txt 2>&1; echo "rc=$?"; tail -2 gpurun_out/full_final.txt
timeout -s KILL 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"; python -c "
import json
d=json.loads(open('gpurun_out/bench_final.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value %.1f G rows/s ms/step %.3f kernel %.3f frac %.3f traffic %s rw ratio %.3f pinned %.2f G rows/s cpu %.0f M rows/s' % (d['value']/1e9, d['ms_per_step'], r['kernel_ms'], r['frac'], r['traffic'], r['stream_rw_ceiling']['frac_of_ceiling'], d['host_pinned']['value']/1e9, d['cpu_baseline']['value']/1e6))"
