#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -s KILL 900 python -m pytest tests -m gpu -q -x > gpurun_out/full_final.txt 2>&1; echo "rc=$?"; grep -v "^  File \"/usr" gpurun_out/full_final.txt | tail -8
