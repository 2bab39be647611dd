#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2t
export LIBC_FATAL_STDERR_=1 BOWGPU_ABORT_TRACE=1
timeout 1500 python -m pytest tests -m gpu -q -x -s > gpurun_out/r2t/full1.txt 2>&1
echo "run 1 rc=$?"
grep -v "^  File \"/usr" gpurun_out/r2t/full1.txt | grep -v "^Extension modules" | tail -60
