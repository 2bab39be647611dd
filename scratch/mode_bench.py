"""aggregation.Mode: wall time per call (device-resident columns and outputs), 1e8 rows, the window-size classes of mode.hip."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from bow_amd import capi
n = 100_000_000
ts, val = capi.gen_sparse(0, n, seed=42)     # ts = 10 i + U{0..9}; value = U{0..9} + 0.5, 30 % nulls: ties everywhere
def run(label, interval, aggs, reps=3):
    s0, W = capi.plan_windows(ts, interval, 0)
    outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
    capi.rolling_aggregate([ts, val], 0, interval, aggs, outs=outs)
    capi.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        _, info = capi.rolling_aggregate([ts, val], 0, interval, aggs, outs=outs)
    capi.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    print("%-46s W=%-9d queued windows %-8d %.2f ms per call  %.2f Grows/s" % (label, W, info.long_windows, ms, n / ms / 1e6))
W0 = ("WindowStart", 0)
run("Mean alone (for scale), 10 rows / window", 100, [W0, ("ArithmeticMean", 1)])
run("Mode, 10 rows / window (lane per window)", 100, [W0, ("Mode", 1)])
run("Mode, 25 rows / window (lane per window)", 250, [W0, ("Mode", 1)])
run("Mode, 50 rows / window (wavefront per window)", 500, [W0, ("Mode", 1)])
run("Mode, 200 rows / window (wavefront per window)", 2_000, [W0, ("Mode", 1)])
run("Mode, 1000 rows / window (workgroup per window)", 10_000, [W0, ("Mode", 1)])
run("Mode, 7000 rows / window (workgroup per window)", 70_000, [W0, ("Mode", 1)])
run("Mode, 10 000 rows / window (global-memory tables)", 100_000, [W0, ("Mode", 1)])
run("Mode, 1e6 rows / window (global-memory tables)", 10_000_000, [W0, ("Mode", 1)], reps=1)
run("Mode, one window of 1e8 rows (sort path)", 2_000_000_000, [W0, ("Mode", 1)], reps=1)
