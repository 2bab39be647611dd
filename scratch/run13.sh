#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scratch/profile_bench.sh r02 > /dev/null 2>&1
bash scratch/profile_configs.sh r02 > /dev/null 2>&1
ls gpurun_out/prof gpurun_out/prof_cfg
for f in gpurun_out/prof_cfg/r02_stdout_*.txt; do echo "== $f"; cat $f; done
