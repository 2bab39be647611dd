cd $GRAFT_REPO_ROOT
timeout -s KILL 1500 python -m pytest tests/test_gpu_aggregate.py tests/test_gpu_fuzz.py -q -m gpu -x -k "long or stream or fuzz" 2>&1 | tail -3
python3 scratch/longw_sweep.py 2>&1 | grep -E " 128 rows| 256 rows|  64 rows" | head -20
