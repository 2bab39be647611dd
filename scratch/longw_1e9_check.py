"""1e9 rows, long windows: the streaming form against the bisection form (WindowStart / Count / Min / Max / First / Last bit for bit,
Sum / Mean / WeightedAverageStep within 1e-12 on this non-cancelling data)."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from bow_amd import capi
n = 1_000_000_000
ts, val = capi.gen_dense(0, n, seed=42)
aggs = [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Count", 1), ("Min", 1), ("Max", 1), ("First", 1), ("Last", 1), ("WeightedAverageStep", 1)]
for interval in (200, 1000, 7777, 3_000_000):
    res = {}
    for mode in ("stream", "classic"):
        capi.set_route(capi.ROUTE_LONG_CLASSIC if mode == "classic" else 0)
        outs, info = capi.rolling_aggregate([ts, val], 0, interval, aggs, out_residency=capi.DEVICE)
        capi.synchronize()
        t0 = time.perf_counter()
        outs, info = capi.rolling_aggregate([ts, val], 0, interval, aggs, out_residency=capi.DEVICE)
        capi.synchronize()
        dt = time.perf_counter() - t0
        res[mode] = [o.host_arrays() for o in outs]
        print("interval %-5d %-8s %s  wall %.2f ms (%.0f G rows/s, outputs allocated inside)  long=%d" % (interval, mode, capi.last_kernel_name(), dt * 1e3, n / dt / 1e9, info.long_windows))
    for (k, _), a, b in zip(aggs, res["stream"], res["classic"]):
        assert np.array_equal(a[1], b[1]), k
        if k in ("WindowStart", "Count", "Min", "Max", "First", "Last"):
            assert np.array_equal(a[0].view(np.uint64), b[0].view(np.uint64)), k
        else:
            x, y = a[0].view(np.float64), b[0].view(np.float64)
            err = np.max(np.abs(x - y) / np.maximum(np.abs(y), 1e-300))
            assert err <= 1e-12, (k, err)
    print("  stream == classic (exact reducers bit for bit, float sums within 1e-12)")
capi.set_route(0)
