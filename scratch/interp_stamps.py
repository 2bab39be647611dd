"""cycles per phase of interp_wave3_kernel (diagnostic build: make -C bow_amd/csrc stamps; BOWGPU_LIB=bow_amd/libbowgpu_stamps.so):
s_memtime deltas summed over the wavefronts, read back from the status words."""
import os, sys
sys.path.insert(0, '.')
import ctypes as C
import numpy as np
from bow_amd import capi
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ts, val = capi.gen_sparse(0, n, seed=42)
ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
L = capi.lib()
def status(first, cnt, zero):
    out = (C.c_uint32 * cnt)()
    capi.check(L.bowgpu_debug_status(first, cnt, out, zero))
    return np.frombuffer(out, dtype=np.uint32).copy()
capi.rolling_interpolate([ts, val], 0, 100, ip, out_residency=capi.DEVICE)
status(32, 16, 1)
capi.rolling_interpolate([ts, val], 0, 100, ip, out_residency=capi.DEVICE)
w = status(32, 16, 1).view(np.uint64)
waves = int(w[7])
names = ["loads + phase 1", "column head (stage rows)", "run pass", "flush", "-"]
tot = sum(int(x) for x in w[:5])
for i in range(4):
    print("%-28s %8.0f ticks per wavefront and (for 1-3) over both columns  %5.1f %%" % (names[i], int(w[i]) / waves, 100.0 * int(w[i]) / tot))
print("wavefronts", waves, "ticks per wavefront", tot / waves, "(s_memtime: 100 MHz)")
