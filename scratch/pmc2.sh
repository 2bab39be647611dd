#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; ROWS=$2
mkdir -p gpurun_out/pmc_$TAG
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" \
         "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_EXP_GDS SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_$TAG/p$i -- python3 bench.py --rows $ROWS --steps 3 --warmup 1 --no-cpu > gpurun_out/pmc_$TAG/p$i.log 2>&1 || tail -3 gpurun_out/pmc_$TAG/p$i.log
done
python3 scratch/pmc_summary.py gpurun_out/pmc_$TAG | tee gpurun_out/pmc_$TAG/summary.txt
