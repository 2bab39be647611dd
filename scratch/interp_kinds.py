"""Interpolate on the configs[2] shape (1e8 rows, 30 % nulls, interval 100) by interpolator of the value column: what the neighbour
lookups of Linear / StepPrevious cost beside the copy of the rows (None)."""
import sys, time
sys.path.insert(0, '.')
import ctypes as C
from bow_amd import capi
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ts, val = capi.gen_sparse(0, n, seed=42)
L = capi.lib()
opts = capi.Options(0, 0, 0)
for kind in ("None", "StepPrevious", "Linear"):
    ip = [{"kind": "WindowStart", "col": 0}, {"kind": kind, "col": 1}]
    carr, iarr = capi._cols([ts, val]), capi._interps(ip)
    m = C.c_int64(0)
    capi.check(L.bowgpu_rolling_interpolate_count(carr, 2, 0, C.c_int64(100), C.byref(opts), iarr, 2, C.byref(m)))
    outs = [capi.OutColumn(m.value, capi.DEVICE) for _ in ip]
    oarr = (capi.Out * 2)()
    def call():
        mm = C.c_int64(0)
        capi.check(L.bowgpu_rolling_interpolate_count(carr, 2, 0, C.c_int64(100), C.byref(opts), iarr, 2, C.byref(mm)))
        for i, o in enumerate(outs):
            oarr[i] = o.c()
        capi.check(L.bowgpu_rolling_interpolate_fill(carr, 2, 0, C.c_int64(100), C.byref(opts), iarr, 2, oarr))
    call(); capi.synchronize()
    t = []
    for _ in range(9):
        t0 = time.perf_counter(); call(); capi.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
    t.sort()
    print("%-13s count + fill %.3f ms  (%d -> %d rows)" % (kind, t[4], n, m.value))
    del outs
