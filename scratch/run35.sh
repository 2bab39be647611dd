#!/bin/bash
cd $GRAFT_REPO_ROOT
BOW_FUZZ_SEEDS=400 timeout -s KILL 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 4 > gpurun_out/fuzz400.txt 2>&1; echo "rc=$?"; grep -v "^  File \"/usr" gpurun_out/fuzz400.txt | tail -12
