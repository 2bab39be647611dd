#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -s KILL 600 python -m pytest tests/test_gpu_aggregate.py -m gpu -q -x -k "chunk_edges" 2>&1 | grep -v "^  File \"/usr" | tail -15
