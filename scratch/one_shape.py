"""One shape, a few calls - the workload of scratch/pmc_any.sh passes.  usage: one_shape.py <longw200|longw1000|tw_was|tw_3int|mean|interp> [rows]"""
import sys
sys.path.insert(0, '.')
from bow_amd import capi
name = sys.argv[1]
n = int(float(sys.argv[2])) if len(sys.argv) > 2 and name != "gen" else 100_000_000
if name == "interp":
    ts, val = capi.gen_sparse(0, n, seed=42)
    ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
    for _ in range(3):
        capi.rolling_interpolate([ts, val], 0, 100, ip, out_residency=capi.DEVICE)
    capi.synchronize()
    sys.exit(0)
if name == "gen":
    # one_shape.py gen <Mean|MinMax|SumMinMax|FirstLast|WAvgStep|TW4> <rows per window> <dense|sparse>: a cell of scratch/midw_sweep.py
    sets = {"Mean": [("WindowStart", 0), ("ArithmeticMean", 1)],
            "MinMax": [("WindowStart", 0), ("Min", 1), ("Max", 1)],
            "SumMinMax": [("WindowStart", 0), ("Sum", 1), ("Min", 1), ("Max", 1)],
            "FirstLast": [("WindowStart", 0), ("First", 1), ("Last", 1)],
            "WAvgStep": [("WindowStart", 0), ("WeightedAverageStep", 1)],
            "TW4": [("WindowStart", 0), ("IntegralStep", 1), ("IntegralTrapezoid", 1), ("WeightedAverageStep", 1), ("WeightedAverageLinear", 1)]}
    aggs, rpw, data = sets[sys.argv[2]], int(sys.argv[3]), sys.argv[4]
    n = 100_000_000
    cols = capi.gen_dense(0, n, seed=42) if data == "dense" else capi.gen_sparse(0, n, seed=3)
    interval = rpw * (1 if data == "dense" else 10)
    s0, W = capi.plan_windows(cols[0], interval, 0)
    outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
    for _ in range(5):
        _, info = capi.rolling_aggregate(list(cols), 0, interval, aggs, outs=outs)
    capi.synchronize()
    print(" ".join(sys.argv[2:]), capi.last_kernel_name(), "kernel %.3f ms" % info.kernel_ms)
    sys.exit(0)
ts, val = capi.gen_dense(0, n, seed=42)
interval, aggs = {"longw200": (200, [("WindowStart", 0), ("ArithmeticMean", 1)]),
                  "longw1000": (1000, [("WindowStart", 0), ("ArithmeticMean", 1)]),
                  "longw1000_5": (1000, [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1)]),
                  "mean": (10, [("WindowStart", 0), ("ArithmeticMean", 1)]),
                  "mean2": (2, [("WindowStart", 0), ("ArithmeticMean", 1)]),
                  "mean100": (100, [("WindowStart", 0), ("ArithmeticMean", 1)]),
                  "mean3": (3, [("WindowStart", 0), ("ArithmeticMean", 1)]),
                  "tw_was": (10, [("WindowStart", 0), ("WeightedAverageStep", 1)]),
                  "tw_3int": (10, [("WindowStart", 0), ("IntegralStep", 1), ("IntegralTrapezoid", 1), ("ArithmeticMean", 1)])}[name]
s0, W = capi.plan_windows(ts, interval, 0)
outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
import time
for _ in range(2):
    _, info = capi.rolling_aggregate([ts, val], 0, interval, aggs, outs=outs)
capi.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    _, info = capi.rolling_aggregate([ts, val], 0, interval, aggs, outs=outs)
capi.synchronize()
print(name, capi.last_kernel_name(), "kernel %.3f ms, wall %.3f ms per call" % (info.kernel_ms, (time.perf_counter() - t0) / 4 * 1e3))
