#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scratch/profile_configs.sh r02 longw longw_kinds > /dev/null 2>&1
cat gpurun_out/prof_cfg/r02_stdout_longw.txt gpurun_out/prof_cfg/r02_stdout_longw_kinds.txt
