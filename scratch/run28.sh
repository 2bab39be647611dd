#!/bin/bash
cd $GRAFT_REPO_ROOT
BOWGPU_LIB=$PWD/scratch/bin/libbowgpu_stamps.so timeout -s KILL 200 python scratch/interp_stamps.py 2>&1 | tail -8
