#!/bin/bash
# usage: scratch/pmc.sh <tag> <rows>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; ROWS=$2
mkdir -p gpurun_out/pmc_$TAG
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
         "FETCH_SIZE GRBM_GUI_ACTIVE" \
         "WRITE_SIZE" \
         "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_$TAG/p$i -- python3 bench.py --rows $ROWS --steps 3 --warmup 1 --no-cpu > gpurun_out/pmc_$TAG/p$i.log 2>&1
done
python3 scratch/pmc_summary.py gpurun_out/pmc_$TAG | tee gpurun_out/pmc_$TAG/summary.txt
