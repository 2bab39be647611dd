"""Interpolate(WindowStart, Linear) at 1e8 rows of gen_sparse data, three calls (for the profiler)."""
import sys
sys.path.insert(0, '.')
from bow_amd import capi
n = 100_000_000
ts, val = capi.gen_sparse(0, n, seed=42)
ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
for _ in range(3):
    out = capi.rolling_interpolate([ts, val], 0, 100, ip, out_residency=capi.DEVICE)
capi.synchronize()
print("rows out", out[0].length)
