#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2u
export AMD_LOG_LEVEL=1
timeout 600 python -X faulthandler -m pytest tests/test_gpu_aggregate.py -m gpu -q -x -k "mixed_shapes" -s > gpurun_out/r2u/mixed.txt 2>&1
echo "rc=$?"; grep -v "^  File \"/usr" gpurun_out/r2u/mixed.txt | head -40
