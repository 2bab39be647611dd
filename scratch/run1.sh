#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
scratch/bin/headline_ab 1e9 9 > gpurun_out/r2a/ab_1e9.txt 2>&1
tail -40 gpurun_out/r2a/ab_1e9.txt
(cd /tmp && TMPDIR=/tmp rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r2a/counters.txt 2>&1)
scratch/ab_pmc.sh r2a 1e9 "t512_reg_full,t512_dma_full,t1024_dma_full,rw_ceiling_reg,rw_ceiling_dma_nt,abl" 2>&1 | tail -150
