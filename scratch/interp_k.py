"""wall time of the Interpolate _fill call alone (count made once, outputs allocated once), 1e8 rows; BOWGPU_LIB picks the build"""
import sys, time, ctypes as C
sys.path.insert(0, '.')
from bow_amd import capi
n = 100_000_000
ts, val = capi.gen_sparse(0, n, seed=42)
ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
carr, iarr = capi._cols([ts, val]), capi._interps(ip)
opts = capi.Options(0, 0, 0)
L = capi.lib()
m = C.c_int64(0)
capi.check(L.bowgpu_rolling_interpolate_count(carr, 2, 0, C.c_int64(100), C.byref(opts), iarr, 2, C.byref(m)))
outs = [capi.OutColumn(m.value, capi.DEVICE) for _ in ip]
oarr = (capi.Out * 2)()
def fill():
    for i, o in enumerate(outs):
        oarr[i] = o.c()
    capi.check(L.bowgpu_rolling_interpolate_fill(carr, 2, 0, C.c_int64(100), C.byref(opts), iarr, 2, oarr))
for _ in range(3): fill()
capi.synchronize()
t = []
for _ in range(10):
    t0 = time.perf_counter(); fill(); capi.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
t.sort()
print("%s: fill wall median %.3f ms  min %.3f" % (sys.argv[1] if len(sys.argv) > 1 else "", t[len(t) // 2], t[0]))
