cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 scratch/interp_wall.py 2>&1 | head -2
timeout -s KILL 900 python -m pytest tests/test_gpu_callers.py tests/test_gpu_fuzz.py tests/test_gpu_fused.py -q -m gpu -x -k "interp or fused or Interp" 2>&1 | tail -3
