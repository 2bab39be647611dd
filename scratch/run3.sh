#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
( time timeout 2000 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > gpurun_out/r2i/pytest.txt 2>&1
tail -25 gpurun_out/r2i/pytest.txt
cat gpurun_out/fullsize_mode.txt
