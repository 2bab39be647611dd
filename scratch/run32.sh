#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -s KILL 300 python scratch/fill_wall.py 2>&1 | tail -6
