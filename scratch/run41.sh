#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -s KILL 900 python -m pytest tests -m gpu -q -x > gpurun_out/full_final.txt 2>&1; echo "rc=$?"; tail -1 gpurun_out/full_final.txt
bash scratch/profile_bench.sh r02 > /dev/null 2>&1
python -c "
import json
d=json.loads(open('gpurun_out/prof/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value %.1f  ms/step %.3f kernel %.3f frac %.3f traffic %s rw %.3f ratio %.3f' % (d['value']/1e9, d['ms_per_step'], r['kernel_ms'], r['frac'], r['traffic'], r['stream_rw_ceiling']['ms'], r['stream_rw_ceiling']['frac_of_ceiling']))"
grep -h "rolling_simple\|finish_bitmaps\|preset_bitmaps" gpurun_out/prof/r02_kernel_stats_bench_1e9.csv | cut -c1-150
tail -6 gpurun_out/prof/r02_pmc_counters_bench_1e9.txt
