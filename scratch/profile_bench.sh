#!/bin/bash
# Collects the rocprofv3 evidence that profiles/ holds for bench.py's default workload (run on the GPU box), all in ONE session:
#   1. --kernel-trace --stats                                       -> gpurun_out/prof/stats
#   2. --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, as MI355X_MICROARCH.md prescribes)
#   3. --kernel-trace --pmc <SQ counters> (one pass: waves, cycles, instruction mix)
#   4. scratch/profile_collect.py boils them down to the files committed under profiles/ (kernel signature + the sha of the kernel's
#      machine code in every row - bow_amd/csrc/kernel_sha.py - so that bench.py only quotes traffic that belongs to the code it runs)
#   5. the bench line itself (un-profiled)                          -> gpurun_out/prof/bench.json
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-pinned > $OUT/stats.log 2>&1
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-pinned > $OUT/fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-pinned > $OUT/write.log 2>&1
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/sq1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-pinned > $OUT/sq1.log 2>&1
python3 scratch/profile_collect.py $OUT $TAG > $OUT/collect.log 2>&1
# 5. the bench line itself (un-profiled), LAST: the counter file of THIS collection is in profiles/ by now, so the line carries
#    roofline.traffic the way the driver's run of the committed tree will (bench.py quotes it when the kernel's sha matches)
cp $OUT/${TAG}_pmc_hbm_traffic_bench_1e9.csv profiles/
timeout -s KILL 400 python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/collect.log
echo "== $OUT/bench.json"; cat $OUT/bench.json
