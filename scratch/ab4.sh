#!/bin/bash
# A/B of several builds of the library on one box: scratch/libbowgpu_<name>.so for each name given
for rep in 1 2; do
  for v in "$@"; do
    echo "== $v bench"; BOWGPU_LIB=$PWD/scratch/libbowgpu_$v.so timeout 200 python bench.py --steps 20 --warmup 3 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'])"
  done
done
for v in "$@"; do
  echo "== $v sweep"; BOWGPU_LIB=$PWD/scratch/libbowgpu_$v.so SWEEP_ROUTES=0 SWEEP_ROWS=${AB_ROWS:-16,32,64,96} timeout 300 python scratch/midw_sweep.py dense Mean MinMax SumMinMax WAvgStep TW4 2>&1 | grep -v worst
done
