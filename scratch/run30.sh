#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -s KILL 300 python scratch/interp_wall.py 2>&1 | tail -3
timeout -s KILL 300 python -m pytest tests/test_gpu_callers.py -m gpu -q -x 2>&1 | tail -1
