#!/bin/bash
# rocprofv3 kernel stats of the secondary configurations (BASELINE.json configs 1-3, time-weighted reducers, Interpolate / fills,
# long windows, small calls, host-resident columns) -> gpurun_out/prof_cfg/<TAG>_kernel_stats_<name>.csv + <TAG>_stdout_<name>.txt,
# which are copied to profiles/ as they are.  Usage: profile_configs.sh r02 [names...]
TAG=${1:-r03}; shift
NAMES=${@:-configs general_bench interp_bench interp_wall longw longw_kinds small_calls host_resident}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_cfg
rm -rf $OUT && mkdir -p $OUT
for name in $NAMES; do
  # the script's own numbers come from an un-profiled run; the kernel stats from a second run of the same script under rocprofv3
  # (BOWGPU_CALL_PROFILE only where the table is ABOUT the per-call split: its stderr lines do not belong in the other tables)
  PROF=0; [ "$name" = small_calls ] && PROF=1
  BOWGPU_CALL_PROFILE=$PROF timeout -s KILL 300 python3 scratch/$name.py 2>&1 | grep -v "^[WE]2026" > $OUT/${TAG}_stdout_${name}.txt
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 scratch/$name.py > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_kernel_stats_${name}.csv
  rm -rf $OUT/$name $OUT/$name.log
done
ls -la $OUT
