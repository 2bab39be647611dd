#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -s KILL 600 python -m pytest tests/test_gpu_aggregate.py -m gpu -q -x 2>&1 | grep -v "^  File \"/usr" | tail -5
timeout -s KILL 300 python scratch/longw_kinds.py 2>&1 | tail -10
