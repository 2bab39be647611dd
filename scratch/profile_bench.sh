#!/bin/bash
# Collects the rocprofv3 evidence that profiles/ holds for bench.py's default workload (run on the GPU box):
#   1. --kernel-trace --stats            -> gpurun_out/prof/stats
#   2. --kernel-trace --pmc FETCH_SIZE   -> gpurun_out/prof/fetch     (separate passes, as MI355X_MICROARCH.md prescribes)
#   3. --kernel-trace --pmc WRITE_SIZE   -> gpurun_out/prof/write
# then scratch/profile_collect.py boils them down to the two CSVs committed under profiles/.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/write.log 2>&1
python3 scratch/profile_collect.py $OUT $TAG
