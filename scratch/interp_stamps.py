"""Where a wavefront of interp_wave3_kernel (BOWGPU_ROUTE=4096: interp_wave2_kernel) spends its cycles (diagnostic build with
-DBOWGPU_STAMPS, BOWGPU_LIB pointing at it: make -C bow_amd/csrc stamps)."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from bow_amd import capi
n = 100_000_000
ts, val = capi.gen_sparse(0, n, seed=42)
ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
out = capi.rolling_interpolate([ts, val], 0, 100, ip, out_residency=capi.DEVICE)
buf = (C.c_uint32 * 16)()
capi.check(capi.lib().bowgpu_debug_status(32, 16, buf, 1))
out = capi.rolling_interpolate([ts, val], 0, 100, ip, out_residency=capi.DEVICE)
capi.check(capi.lib().bowgpu_debug_status(32, 16, buf, 1))
w = np.frombuffer(bytes(buf), dtype=np.uint64)
waves = int(w[7])
wave2 = capi.get_route() & capi.ROUTE_INTERP_WAVE2
names = ["issue round 1", "wait round 1 + phase 1", "column head (bits, carry, rows staged)", "run pass", "flush"] if wave2 else \
        ["round 1 + phase 1", "column head (values, bits, carry, rows staged)", "run pass", "flush", "-"]
tot = sum(int(w[i]) for i in range(5))
print("wavefronts %d, %.0f cycles each" % (waves, tot / max(waves, 1)))
for i, nm in enumerate(names):
    print("  %-42s %8.0f cycles per wavefront  %5.1f %%" % (nm, int(w[i]) / max(waves, 1), 100.0 * int(w[i]) / max(tot, 1)))
