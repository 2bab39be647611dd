"""the smaller callers at 1e8 rows: IsColSorted (bowassertion.go:15-81) and FillLinear (bowfill.go:14-103), wall per call"""
import sys, time
sys.path.insert(0, '.')
from bow_amd import capi
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
def med(fn, reps=9):
    fn(); capi.synchronize()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); capi.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
    return sorted(t)[len(t) // 2]
ts, val = capi.gen_sparse(0, n, seed=42)
valid = capi.aggregate_whole([ts, val], 0, [("Count", 1)])[0].to_list()[0]
valn = capi.Column(val.values, val.validity, capi.FLOAT64, 0, n, n - valid)
dts, dval = capi.gen_dense(0, n, seed=42)
for label, col, gb in (("IsColSorted, Int64 without nulls", ts, 0.8), ("IsColSorted, Float64 with 30 % nulls", valn, 0.8125)):
    w = med(lambda: capi.is_col_sorted(col))
    print("%-40s wall %.3f ms per call  %6.1f G rows/s  %.2f of 8 TB/s on %.2f GB" % (label, w, n / w / 1e6, gb / w / 8, gb))
out = capi.OutColumn((n + 511) // 512 * 512, capi.DEVICE)
import ctypes as C
carr = capi._cols([ts, valn])
def fl():
    o = out.c(); u = C.c_int32(0)
    capi.check(capi.lib().bowgpu_fill_linear(carr, 2, 0, 1, C.byref(o), C.byref(u)))
w = med(fl)
print("%-40s wall %.3f ms per call  %6.1f G rows/s  %.2f of 8 TB/s on %.2f GB moved (kernel %s %.3f ms)" %
      ("FillLinear, 30 % nulls, ref = ts", w, n / w / 1e6, 2.4125 / w / 8, 2.4125, capi.last_kernel_name(), capi.last_kernel_ms()))
def fls():
    o = out.c(); u = C.c_int32(0)
    capi.check(capi.lib().bowgpu_fill_linear_sorted(carr, 2, 0, 1, C.byref(o), C.byref(u)))
w = med(fls)
print("%-40s wall %.3f ms per call  %6.1f G rows/s  %.2f of 8 TB/s on %.2f GB moved (kernel %s %.3f ms)" %
      ("... the ref column checked by the caller", w, n / w / 1e6, 2.4125 / w / 8, 2.4125, capi.last_kernel_name(), capi.last_kernel_ms()))
