for st in 0 1; do for occ in 2 3 4; do
echo "strided=$st occ=$occ"; BOWGPU_FAST_STRIDED=$st BOWGPU_FAST_OCC=$occ python bench.py --rows 400000000 --steps 10 --warmup 2 --no-cpu | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'], d['ms_per_step'])"
done; done
