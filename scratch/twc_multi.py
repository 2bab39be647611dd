"""rolling_twc_kernel with two nullable value columns (kMulti): time-weighted reducers per column, 1e8 rows, 30 % nulls"""
import sys, time
sys.path.insert(0, '.')
from bow_amd import capi
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ts, v1 = capi.gen_sparse(0, n, seed=42)
_, v2 = capi.gen_sparse(0, n, seed=43)
cols = [ts, v1, v2]
for rows in (32, 64, 160):
    for label, aggs in (("WAvgStep x2", [("WindowStart", 0), ("WeightedAverageStep", 1), ("WeightedAverageStep", 2)]),
                        ("TW4 x2", [("WindowStart", 0)] + [(k, c) for c in (1, 2) for k in ("IntegralStep", "IntegralTrapezoid", "WeightedAverageStep", "WeightedAverageLinear")])):
        interval = rows * 10
        s0, W = capi.plan_windows(ts, interval, 0)
        outs = [capi.OutColumn(W + 2, capi.DEVICE) for _ in aggs]
        kms = []
        for _ in range(7):
            _, info = capi.rolling_aggregate(cols, 0, interval, aggs, outs=outs)
            kms.append(info.kernel_ms)
        print("%3d rows/window %-12s kernel %.3f ms (%s)" % (rows, label, sorted(kms)[len(kms) // 2], capi.last_kernel_name()))
