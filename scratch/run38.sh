#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -s KILL 600 python -m pytest tests/test_gpu_aggregate.py tests/test_gpu_callers.py tests/test_gpu_threads.py -m gpu -q -x 2>&1 | tail -2
timeout -s KILL 300 python scratch/host_resident.py 2>&1 | head -1
BOWGPU_COPY_THREADS=0 timeout -s KILL 300 python scratch/host_resident.py 2>&1 | head -1
BOWGPU_COPY_THREADS=7 timeout -s KILL 300 python scratch/host_resident.py 2>&1 | head -1
