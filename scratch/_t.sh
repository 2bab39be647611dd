cd $GRAFT_REPO_ROOT
python3 scratch/small_calls.py 2>&1 | tail -12
timeout -s KILL 2400 python -m pytest tests -q -m gpu -x --ignore=tests/test_gpu_fullsize.py 2>&1 | tail -4
