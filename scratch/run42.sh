#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -s KILL 600 python -m pytest tests/test_gpu_aggregate.py tests/test_gpu_threads.py -m gpu -q -x 2>&1 | tail -2
timeout -s KILL 300 python scratch/longw_kinds.py 2>&1 | tail -10
