#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2p
python scratch/interp_wall.py > gpurun_out/r2p/interp_wall.txt 2>&1; cat gpurun_out/r2p/interp_wall.txt
bash scratch/pmc_any.sh r2p_wave2 interp_wave2 scratch/interp_only.py 2>&1 | tail -32 | grep -v "^SQ_ACTIVE_INST_F\|SQ_ACTIVE_INST_VM\|SQ_IFETCH\|GRBM\|SQ_INST_LEVEL\|SQ_ACTIVE_INST"
