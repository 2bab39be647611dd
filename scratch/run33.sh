#!/bin/bash
cd $GRAFT_REPO_ROOT
BOW_BENCH_SINGLE_DEVICE=1 BOW_BENCH_BACKEND=gloo timeout -s KILL 600 python bench.py --gpus 2 --rows 1e8 --steps 5 --warmup 2 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err; echo "bench2 rc=$?"; python -c "
import json
d=json.loads(open('gpurun_out/bench_2rank.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','n_gpus','ms_per_step','exchange_ms','exchanges_per_step')}, d['config']['parallelism'], d['roofline']['kernel_ms'])"
timeout -s KILL 600 python -m pytest tests/test_gpu_sharded.py tests/test_bench_launcher.py -m gpu -q -x 2>&1 | tail -2
