#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in A B; do echo "== $v"; BOWGPU_LIB=$PWD/scratch/bin/libbowgpu_$v.so timeout -s KILL 300 python scratch/interp_bench.py 2>&1 | grep -E "^Fill|IsCol|whole"; done; done
