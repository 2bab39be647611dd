import gc, sys, time
sys.path.insert(0, '.')
from bow_amd import capi
n = 100_000_000
ts, val = capi.gen_dense(0, n, seed=42)
for interval in (10, 100, 200, 1000, 10_000, 1_000_000, 10**9):
    aggs = [("WindowStart", 0), ("ArithmeticMean", 1)]
    s0, W = capi.plan_windows(ts, interval, 0)
    outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
    for _ in range(2):
        capi.rolling_aggregate([ts, val], 0, interval, aggs, outs=outs)
    gc.collect()            # (the previous interval's output buffers: their hipFree must not land inside the timed calls)
    capi.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        _, info = capi.rolling_aggregate([ts, val], 0, interval, aggs, outs=outs)
    capi.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("interval %-10d W=%-9d long=%-8d wall %.3f ms  tile kernel %.3f ms  %.1f Grows/s" % (interval, W, info.long_windows, dt * 1e3, info.kernel_ms, n / dt / 1e9))
