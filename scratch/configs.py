"""Quick timing of BASELINE.json configs 1-3 on one GPU (parity cases, not the bench line)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from bow_amd import capi
n = 100_000_000
def run(label, cols, interval, aggs, bytes_per_row, offset=0, reps=5):
    s0, W = capi.plan_windows(cols[0], interval, offset)
    outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
    ms = []
    for i in range(reps + 1):
        _, info = capi.rolling_aggregate(cols, 0, interval, aggs, offset=offset, outs=outs)
        ms.append(info.kernel_ms)
    k = sum(ms[1:]) / reps
    print("%-58s W=%-9d kernel %.3f ms  %.1f Grows/s  %.0f GB/s (%.1f%% of 8 TB/s) long=%d" % (
        label, W, k, n / k / 1e6, n * bytes_per_row / k / 1e6, n * bytes_per_row / k / 1e6 / 80, info.long_windows))
ts, val = capi.gen_dense(0, n, seed=42)
run("cfg1 dense Sum/Mean/Min/Max (+WindowStart), interval 10", [ts, val], 10,
    [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1)], 16)
run("cfg1 dense WindowStart+Mean, interval 10", [ts, val], 10, [("WindowStart", 0), ("ArithmeticMean", 1)], 16)
run("cfg1 dense WindowStart+Mean, interval 1000 (long windows)", [ts, val], 1000, [("WindowStart", 0), ("ArithmeticMean", 1)], 16)
ts2, val2 = capi.gen_sparse(0, n, seed=42)
# (the column's null count as a Bow knows it - Data().NullN(), what the cgo shim passes; -1 makes every call count the bits first)
val2 = capi.Column(val2.values, val2.validity, capi.FLOAT64, 0, n, n - capi.aggregate_whole([ts2, val2], 0, [("Count", 1)])[0].to_list()[0])
run("cfg2 sparse 30% nulls WindowStart+Mean, interval 100", [ts2, val2], 100, [("WindowStart", 0), ("ArithmeticMean", 1)], 16.125)
ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
filled = capi.rolling_interpolate([ts2, val2], 0, 100, ip, out_residency=capi.DEVICE)
# the call itself, outputs allocated once: count + fill as the shim makes them, back to back (scratch/interp_wall.py has the split)
import ctypes as C
carr, iarr, opts = capi._cols([ts2, val2]), capi._interps(ip), capi.Options(0, 0, 0)
outs = [capi.OutColumn(filled[0].length, capi.DEVICE) for _ in ip]
oarr = (capi.Out * 2)()
def interpolate_call():
    m_ = C.c_int64(0)
    capi.check(capi.lib().bowgpu_rolling_interpolate_count(carr, 2, 0, C.c_int64(100), C.byref(opts), iarr, 2, C.byref(m_)))
    for i, o in enumerate(outs):
        oarr[i] = o.c()
    capi.check(capi.lib().bowgpu_rolling_interpolate_fill(carr, 2, 0, C.c_int64(100), C.byref(opts), iarr, 2, oarr))
interpolate_call(); capi.synchronize()
tt = []
for _ in range(7):
    t1 = time.perf_counter(); interpolate_call(); capi.synchronize(); tt.append(time.perf_counter() - t1)
w = sorted(tt)[3]
print("cfg2 Interpolate(WindowStart, Linear) interval 100: %d -> %d rows, count + fill %.3f ms wall per call (outputs preallocated)  %.1f Grows/s  %.0f GB/s moved (%.1f%% of 8 TB/s)" %
      (n, filled[0].length, w * 1e3, n / w / 1e9, (n * 16.125 + filled[0].length * 16.125) / w / 1e9, (n * 16.125 + filled[0].length * 16.125) / w / 8e10))
del outs
m = filled[0].length
cols2 = [capi.Column(filled[0].values, None, capi.INT64, 0, m, 0), capi.Column(filled[1].values, filled[1].validity, capi.FLOAT64, 0, m, -1)]
run("cfg2 Mean after Linear fill, interval 100", cols2, 100, [("WindowStart", 0), ("ArithmeticMean", 1)], 16.125)
del filled, cols2
cols8 = [ts] + [capi.gen_dense(0, n, seed=42 + k)[1] for k in range(8)]
run("cfg3 8 float64 columns Mean each (+WindowStart), interval 10", cols8, 10,
    [("WindowStart", 0)] + [("ArithmeticMean", 1 + k) for k in range(8)], 72)
