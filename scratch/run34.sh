#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scratch/profile_configs.sh r02 interp_bench interp_wall > /dev/null 2>&1
cat gpurun_out/prof_cfg/r02_stdout_interp_wall.txt gpurun_out/prof_cfg/r02_stdout_interp_bench.txt
