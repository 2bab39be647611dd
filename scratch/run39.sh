#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -s KILL 500 python scratch/longw_1e9_check.py 2>&1 | tail -12
