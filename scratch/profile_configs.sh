#!/bin/bash
# rocprofv3 kernel stats of the secondary configurations (BASELINE.json configs 1-3, time-weighted reducers, Interpolate / fills,
# long windows) -> gpurun_out/prof_cfg/<name>_kernel_stats.csv ; copied to profiles/ as r01_kernel_stats_<name>_1e8.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_cfg
rm -rf $OUT && mkdir -p $OUT
for name in configs general_bench interp_bench longw mode_bench; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 scratch/$name.py > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv
  grep -v "^[WE]2026" $OUT/$name.log > $OUT/${name}_stdout.txt
done
[ -x scratch/bin/copy_ceiling ] && scratch/bin/copy_ceiling > $OUT/copy_ceiling_stdout.txt 2>&1
ls -la $OUT
