#!/bin/bash
# Every profiles/<tag>_* file of a round from ONE script at ONE commit (run on the GPU box through gpurun):
#   scratch/profile_all.sh r03     -> gpurun_out/profiles_r03/  (copy its files to profiles/ as they are)
# 1. scratch/profile_bench.sh   : the bench line un-profiled, rocprofv3 --kernel-trace --stats, the FETCH_SIZE / WRITE_SIZE passes and
#                                 the SQ / L2 counter passes of the benched kernel (traffic rows carry the kernel signature + source hash)
# 2. scratch/profile_configs.sh : the secondary shapes, each un-profiled (stdout) and under --kernel-trace --stats
# 3. scratch/pmc_sq.sh          : counter passes of the Interpolate fill kernel, the time-weighted tile kernel and the long-window
#                                 streaming kernel (1000-row windows: dense / 30 % nulls, extrema / time-weighted; FETCH_SIZE included)
# (--pmc passes never carry another trace domain; every profiled process is python3 itself: no env / shell hop after `--`)
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
DST=gpurun_out/profiles_$TAG
rm -rf $DST && mkdir -p $DST
(git rev-parse HEAD 2>/dev/null || cat scratch/HEAD_COMMIT 2>/dev/null || echo "(snapshot without .git)") > $DST/${TAG}_commit.txt
bash scratch/profile_bench.sh $TAG > $DST/profile_bench.log 2>&1
cp gpurun_out/prof/${TAG}_* $DST/ 2>/dev/null
cp gpurun_out/prof/bench.json $DST/${TAG}_bench_1e9.json 2>/dev/null
bash scratch/profile_configs.sh $TAG configs general_bench interp_wall fill_wall longw longw_kinds longw_sweep small_calls host_resident > $DST/profile_configs.log 2>&1
cp gpurun_out/prof_cfg/${TAG}_* $DST/ 2>/dev/null
bash scratch/pmc_sq.sh w3 interp_wave3 scratch/interp_pmc.py > /dev/null 2>&1
cp gpurun_out/pmc_w3/summary.txt $DST/${TAG}_pmc_interp_wave3_1e8.txt 2>/dev/null
bash scratch/pmc_sq.sh tw rolling_tw scratch/one_shape.py tw_was > /dev/null 2>&1
cp gpurun_out/pmc_tw/summary.txt $DST/${TAG}_pmc_tw_was_1e8.txt 2>/dev/null
for V in "minmax dense" "tw dense" "minmax sparse" "tw sparse"; do
  set -- $V
  bash scratch/pmc_sq.sh lw_$1_$2 long_short scratch/longw_one.py $1 $2 > /dev/null 2>&1
  cp gpurun_out/pmc_lw_$1_$2/summary.txt $DST/${TAG}_pmc_long_short_$1_$2_1e8.txt 2>/dev/null
done
ls -la $DST
