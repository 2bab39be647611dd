cd $GRAFT_REPO_ROOT
timeout -s KILL 900 python -m pytest tests -q -m gpu -x -k "whole" 2>&1 | tail -12
python3 scratch/whole_wall.py 2>&1 | grep -v "^[WE]2026"
