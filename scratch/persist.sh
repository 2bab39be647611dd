for cfg in "0 0" "1 8" "1 12" "1 16" "1 20"; do set -- $cfg
echo "persist=$1 waves=$2"; BOWGPU_FAST_PERSIST=$1 BOWGPU_FAST_WAVES=$2 python bench.py --rows 1000000000 --steps 10 --warmup 2 --no-cpu | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['roofline']['kernel_ms'], round(d['roofline']['frac'],4), d['ms_per_step'])"
done
