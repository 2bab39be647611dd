#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -s KILL 900 python -m pytest tests/test_bench_launcher.py -m gpu -q -x 2>&1 | grep -v "^  File \"/usr" | tail -15
