"""VERDICT r05 item 2, the regression case of round 5's one-lane-for-every-queued-window attempt: 10-row windows with ONE 1 500-row window in
every 2 000 rows, 1e7 rows, WindowStart + ArithmeticMean.  The call averages 39 rows per window, so the default route keeps the host
in the loop for its queued windows (kQueueMinAvgRows = 64); forced through long_queue_kernel the 1 500-row windows are LISTED by it (longer
than kQueueWalkMaxRows) and reduced by the chunked machinery - not walked by one lane."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from bow_amd import capi
from oracle import pyoracle as orc
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
i = np.arange(n, dtype=np.int64)
blk, r = i // 2000, i % 2000
ts = blk * 510 + np.where(r < 1500, (r * 10) // 1500, 10 + (r - 1500))
val = np.random.default_rng(1).random(n)
cols = [capi.Column(ts, None, capi.INT64).to_device(), capi.Column(val, None, capi.FLOAT64).to_device()]
aggs = [("WindowStart", 0), ("ArithmeticMean", 1)]
W = capi.plan_windows(cols[0], 10, 0)[1]
outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
ref = None
for label, mask in (("default", 0), ("queue through the host (rounds 1 - 5)", capi.ROUTE_QUEUE_HOST), ("long_queue_kernel forced", capi.ROUTE_QUEUE_DEVICE)):
    with capi.route(mask):
        for _ in range(3):
            capi.rolling_aggregate(cols, 0, 10, aggs, outs=outs)
        capi.synchronize()
        t = []
        for _ in range(9):
            t0 = time.perf_counter(); _, info = capi.rolling_aggregate(cols, 0, 10, aggs, outs=outs); t.append((time.perf_counter() - t0) * 1e3)
        chk = capi.checksum64(outs[1].values, W)
        ref = ref or chk
        print("%-40s wall %.3f ms per call (median of 9), kernel bracket %.3f ms, %d windows, %d of them order-free, outputs %s" %
              (label, sorted(t)[4], info.kernel_ms, W, info.long_windows, "identical" if chk == ref else "DIFFER (order-free sums: within the stated bound)"), flush=True)
if n <= 10_000_000:
    exp, _ = orc.aggregate([orc.Column(ts, None, orc.INT64), orc.Column(val, None, orc.FLOAT64)], 0, 10, aggs)
    got = outs[1].host_arrays()[0]
    w = exp[1].values[:W].view(np.float64)
    print("max |GPU - oracle| / |oracle| over %d means: %.3g" % (W, float(np.max(np.abs(got - w) / np.maximum(np.abs(w), 1e-300)))))
