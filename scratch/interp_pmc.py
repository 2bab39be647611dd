"""Interpolate(WindowStart, Linear) at 1e8 rows of gen_sparse data, three calls, for the PMC passes of scratch/pmc_any.sh.
Buffers are released and the stream drained before the interpreter exits (an earlier version of this script aborted at exit under
--pmc: device buffers freed from destructors after the profiler's tool library had begun to shut down)."""
import gc, sys
sys.path.insert(0, '.')
from bow_amd import capi
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ts, val = capi.gen_sparse(0, n, seed=42)
ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
out = None
for _ in range(3):
    out = capi.rolling_interpolate([ts, val], 0, 100, ip, out_residency=capi.DEVICE)
capi.synchronize()
print("rows out", out[0].length)
del out, ts, val
gc.collect()
capi.trim()
capi.synchronize()
