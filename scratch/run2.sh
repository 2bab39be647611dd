#!/bin/bash
cd $GRAFT_REPO_ROOT
scratch/ab_pmc2.sh r2f 1e9 "t512_reg_full,t512_reg_hotst,t512_reg_al16_nti,rw_reg_st0,rw_reg_st2_al16,rw_reg_st7_lds,rw_reg_st4_none" 2>&1 | tail -80
