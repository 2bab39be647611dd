#!/bin/bash
# round 6, after rolling_fused.hip changed (the ISA pass: DESIGN 4): the counter file and the timing tables of the fused kernel again.
# The plain-Aggregate section of the counter file speaks about rolling_simple_kernel, which did not change: carried over from the committed file.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r06}
DST=gpurun_out/profiles_${TAG}c
rm -rf $DST && mkdir -p $DST
export PMC_LIMIT=${PMC_LIMIT:-70}
timeout -s KILL 300 python3 scratch/cfg2_fused.py 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_cfg2_fused.txt
CFG2_ORDER=rev CFG2_WARM=0 timeout -s KILL 300 python3 scratch/cfg2_fused.py 1e8 quick 2>&1 | grep -v "^[WE]2026" > $DST/${TAG}_stdout_cfg2_fused_rev.txt
: > $DST/${TAG}_pmc_rolling_fused.txt
fused_case() {
  echo "== Interpolate -> Aggregate, offset $1, $2 reducers (configs[2]: 1e8 irregular rows, 30 % nulls, interval 100)" >> $DST/${TAG}_pmc_rolling_fused.txt
  bash scratch/pmc_sq.sh fused_$1_$2 rolling_fused scratch/fused_one.py 1e8 $1 $2 | grep -v "^pass" >> $DST/${TAG}_pmc_rolling_fused.txt
}
fused_case 0 2; fused_case 7 2; fused_case 0 7
sed -n '/^== the same rows through bowgpu_rolling_aggregate/,$p' profiles/${TAG}_pmc_rolling_fused.txt >> $DST/${TAG}_pmc_rolling_fused.txt
ls -la $DST
