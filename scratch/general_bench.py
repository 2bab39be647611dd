"""Throughput of the reducers that take the general kernel (time-weighted, inclusive windows) and of the wave kernel, 1e8 rows."""
import sys, os
sys.path.insert(0, '.')
from bow_amd import capi
n = 100_000_000
ts, val = capi.gen_dense(0, n, seed=42)
ts2, val2 = capi.gen_sparse(0, n, seed=42)
def run(label, cols, interval, aggs, bytes_per_row, route=0, reps=5):
    capi.set_route(route)
    s0, W = capi.plan_windows(cols[0], interval, 0)
    outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
    ms = []
    for i in range(reps + 1):
        _, info = capi.rolling_aggregate(cols, 0, interval, aggs, outs=outs)
        ms.append(info.kernel_ms)
    k = sum(ms[1:]) / reps
    print("%-64s %-22s kernel %.3f ms  %.1f Grows/s  %.1f%% of 8 TB/s" % (label, capi.last_kernel_name(), k, n / k / 1e6, n * bytes_per_row / k / 1e6 / 80))
W0 = ("WindowStart", 0)
run("Mean (simple kernel)", [ts, val], 10, [W0, ("ArithmeticMean", 1)], 16)
run("Mean (wave kernel)", [ts, val], 10, [W0, ("ArithmeticMean", 1)], 16, capi.ROUTE_NO_SIMPLE)
run("Mean (general kernel)", [ts, val], 10, [W0, ("ArithmeticMean", 1)], 16, capi.ROUTE_FORCE_GENERAL)
run("WeightedAverageStep", [ts, val], 10, [W0, ("WeightedAverageStep", 1)], 16)
run("WeightedAverageLinear (inclusive)", [ts, val], 10, [W0, ("WeightedAverageLinear", 1)], 16)
run("IntegralStep + IntegralTrapezoid + Mean", [ts, val], 10, [W0, ("IntegralStep", 1), ("IntegralTrapezoid", 1), ("ArithmeticMean", 1)], 16)
run("sparse 30% nulls: WeightedAverageStep I=100", [ts2, val2], 100, [W0, ("WeightedAverageStep", 1)], 16.125)
run("Mean x Factor(0.5) (wave kernel: factors)", [ts, val], 10, [W0, ("ArithmeticMean", 1, [0.5])], 16)
# nanosecond timestamps (rows span far more than 2^32 from the first window): the kWide variants
import numpy as np
tsn = capi.Column((np.arange(n, dtype=np.int64) * 100_000_000 + 1_700_000_000_000_000_000)).to_device()
run("ns timestamps, 1 s windows: Mean", [tsn, val], 1_000_000_000, [W0, ("ArithmeticMean", 1)], 16)
run("ns timestamps, 1 s windows: WeightedAverageLinear", [tsn, val], 1_000_000_000, [W0, ("WeightedAverageLinear", 1)], 16)
run("ns timestamps: Mean (wave kernel)", [tsn, val], 1_000_000_000, [W0, ("ArithmeticMean", 1)], 16, capi.ROUTE_NO_SIMPLE)
