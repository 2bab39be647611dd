#!/bin/bash
cd $GRAFT_REPO_ROOT
BOW_FUZZ_BIG=1 BOW_FUZZ_SEEDS=48 timeout -s KILL 1700 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 4 -k "test_fuzz_aggregate or test_fuzz_agg" > gpurun_out/fuzzbig.txt 2>&1; echo "rc=$?"; grep -v "^  File \"/usr" gpurun_out/fuzzbig.txt | tail -6
