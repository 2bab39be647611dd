"""Interpolate(WindowStart, Linear) on the configs[2] shape (1e8 rows, 30 % nulls, interval 100): wall time of the _count call, of the
_fill call that follows it (outputs allocated once, outside the timing), and of both; capi.ROUTE_INTERP_TILE
switch to the older kernels for an A/B in one process."""
import os, sys, time
sys.path.insert(0, '.')
import ctypes as C
from bow_amd import capi
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ts, val = capi.gen_sparse(0, n, seed=42)
# the column's null count as a Bow knows it (Data().NullN(): what the cgo shim passes; -1 makes every call count the bits first)
val = capi.Column(val.values, val.validity, capi.FLOAT64, 0, n, n - capi.aggregate_whole([ts, val], 0, [("Count", 1)])[0].to_list()[0])
cols = [ts, val]
ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
carr, iarr = capi._cols(cols), capi._interps(ip)
opts = capi.Options(0, 0, 0)
L = capi.lib()
def count():
    m = C.c_int64(0)
    capi.check(L.bowgpu_rolling_interpolate_count(carr, 2, 0, C.c_int64(100), C.byref(opts), iarr, 2, C.byref(m)))
    return m.value
n_out = count()
# (capacity: the rows rounded up to 512, i.e. bitmaps of whole 64-byte blocks as the Arrow allocator hands out - a bitmap that reaches the
# end of its last 32-bit word is written in place; "exact" as the first argument after the row count: exactly n_out rows of capacity)
exact = len(sys.argv) > 2 and sys.argv[2] == "exact"
outs = [capi.OutColumn(n_out if exact else (n_out + 511) // 512 * 512, capi.DEVICE) for _ in ip]
oarr = (capi.Out * 2)()
def fill():
    for i, o in enumerate(outs):
        oarr[i] = o.c()
    capi.check(L.bowgpu_rolling_interpolate_fill(carr, 2, 0, C.c_int64(100), C.byref(opts), iarr, 2, oarr))
def timeit(fn, reps=10):
    fn(); capi.synchronize()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); capi.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
    t.sort()
    return t[len(t) // 2]
for _ in range(20):   # (the first dozen launches of a process run 5 % slower than the rest: clocks, page mappings)
    count(); fill()
capi.synchronize()
# ("one" as the second argument: the default route only - what a profiler pass wants)
only_default = len(sys.argv) > 2 and sys.argv[2] == "one"
for label, mask in (("wave3 (default)", 0), ("wave3, bitmap copies", capi.ROUTE_INTERP_COPIES), ("tile", capi.ROUTE_INTERP_TILE))[:1 if only_default else 3]:
    capi.set_route(mask)
    both = timeit(lambda: (count(), fill()))
    c_ms = timeit(count)
    f_alone = timeit(fill)              # no _count in front: the fill call makes its own pass 1
    print("%-32s %d -> %d rows: count %.3f ms, count+fill %.3f ms (%.1f G rows/s), fill on its own (its own count pass inside) %.3f ms" %
          (label, n, n_out, c_ms, both, n / both / 1e6, f_alone))
