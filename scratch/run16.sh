#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout -s KILL 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/bench_final.json
BOW_BENCH_SINGLE_DEVICE=1 BOW_BENCH_BACKEND=gloo timeout -s KILL 600 python bench.py --gpus 2 --rows 1e8 --steps 5 --warmup 2 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err; echo "bench2 rc=$?"; cut -c1-400 gpurun_out/bench_2rank.json
