"""Interpolate / fills at 1e8 rows (configs[2] shape): wall time per call; run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import sys, time
sys.path.insert(0, '.')
from bow_amd import capi
n = 100_000_000
ts, val = capi.gen_sparse(0, n, seed=42)
ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
for rep in range(3):
    t0 = time.perf_counter()
    filled = capi.rolling_interpolate([ts, val], 0, 100, ip, out_residency=capi.DEVICE)
    capi.synchronize()
    t1 = time.perf_counter()
    print("Interpolate(WindowStart, Linear) I=100: %d -> %d rows  %.2f ms  %.1f Grows/s" % (n, filled[0].length, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9))
    del filled
for method in ("Previous", "Next", "Mean"):
    for rep in range(2):
        t0 = time.perf_counter()
        out, _ = capi.fill(val, method, out_residency=capi.DEVICE)
        capi.synchronize()
        t1 = time.perf_counter()
    print("Fill%s: %.2f ms  %.1f Grows/s" % (method, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9))
    del out
for rep in range(2):
    t0 = time.perf_counter()
    out, _ = capi.fill_linear([ts, val], 0, 1, out_residency=capi.DEVICE)
    capi.synchronize()
    t1 = time.perf_counter()
print("FillLinear: %.2f ms  %.1f Grows/s" % ((t1 - t0) * 1e3, n / (t1 - t0) / 1e9))
t0 = time.perf_counter(); s = capi.is_col_sorted(ts); capi.synchronize(); t1 = time.perf_counter()
t0 = time.perf_counter(); s = capi.is_col_sorted(ts); capi.synchronize(); t1 = time.perf_counter()
print("IsColSorted: %.2f ms" % ((t1 - t0) * 1e3))
aggs = [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1), ("Count", 1), ("WeightedAverageLinear", 1)]
for rep in range(2):
    t0 = time.perf_counter(); outs = capi.aggregate_whole([ts, val], 0, aggs); capi.synchronize(); t1 = time.perf_counter()
print("aggregation.Aggregate (whole frame, 7 reducers over 2 columns): %.2f ms  %.1f Grows/s" % ((t1 - t0) * 1e3, n / (t1 - t0) / 1e9))
