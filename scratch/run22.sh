#!/bin/bash
cd $GRAFT_REPO_ROOT
BOWGPU_CALL_PROFILE=1 timeout -s KILL 200 python scratch/small_calls.py 2>&1 | grep -E "device cols|profile" | head -40
