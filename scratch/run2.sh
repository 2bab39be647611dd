#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 600 scratch/bin/headline_ab 1e9 9 > gpurun_out/r2g/ab_1e9.txt 2>&1
cat gpurun_out/r2g/ab_1e9.txt
