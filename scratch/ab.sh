#!/bin/bash
# A/B two builds of the library in the same box: scratch/libbowgpu_a.so (A) vs bow_amd/libbowgpu.so (B)
for rep in 1 2; do
  echo "== A"; BOWGPU_LIB=$PWD/scratch/libbowgpu_a.so python bench.py --steps 20 --warmup 3 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline'])"
  echo "== B"; python bench.py --steps 20 --warmup 3 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline'])"
done
echo "== A longw"; BOWGPU_LIB=$PWD/scratch/libbowgpu_a.so python scratch/longw.py
echo "== B longw"; python scratch/longw.py
