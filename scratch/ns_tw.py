"""nanosecond epochs, 1 s windows, WeightedAverageLinear over 1e8 rows: default route and the float64-timestamps route (for prof_any.sh / A-B)"""
import sys
sys.path.insert(0, '.')
import numpy as np
from bow_amd import capi
n = 100_000_000
_, val = capi.gen_dense(0, n, seed=42)
tsn = capi.Column((np.arange(n, dtype=np.int64) * 100_000_000 + 1_700_000_000_000_000_000)).to_device()
aggs = [("WindowStart", 0), ("WeightedAverageLinear", 1)]
s0, W = capi.plan_windows(tsn, 1_000_000_000, 0)
outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
for label, mask in (("default", 0), ("float64 times", capi.ROUTE_TW_F64)):
    with capi.route(mask):
        ms = []
        for _ in range(6):
            _, info = capi.rolling_aggregate([tsn, val], 0, 1_000_000_000, aggs, outs=outs)
            ms.append(info.kernel_ms)
        print("%-14s %s kernel %.3f ms  %.1f%% of 8 TB/s  checksum %s" % (label, capi.last_kernel_name(), sorted(ms)[2], n * 16 / (sorted(ms)[2] * 1e-3) / 8e12 * 100, capi.checksum64(outs[1].values_dev, W) if hasattr(outs[1], "values_dev") else ""))
