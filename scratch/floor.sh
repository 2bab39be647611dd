for cfg in "0 0" "1 8" "1 16" "1 32"; do set -- $cfg
echo "floor persist=$1 waves=$2"; BOWGPU_DBG_STOP=6 BOWGPU_FAST_PERSIST=$1 BOWGPU_FAST_WAVES=$2 python bench.py --rows 1000000000 --steps 5 --warmup 1 --no-cpu 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['roofline']['kernel_ms'], round(d['roofline']['frac'],4))"
done
