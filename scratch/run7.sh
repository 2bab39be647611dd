#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2s
( timeout 900 python -m pytest tests/test_gpu_aggregate.py -m gpu -q -x -k "pinned or planned or bitmaps" 2>&1 | tail -15 ) > gpurun_out/r2s/pytest.txt 2>&1
tail -15 gpurun_out/r2s/pytest.txt
python scratch/host_resident.py > gpurun_out/r2s/host_resident.txt 2>&1; cat gpurun_out/r2s/host_resident.txt
