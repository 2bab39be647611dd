"""Per-call latency of Rolling.Aggregate at the reference's own benchmark sizes (10 .. 100000 rows), host- and device-resident."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from bow_amd import capi
aggs = [("WindowStart", 0), ("ArithmeticMean", 1)]
for n in (10, 1000, 100_000, 1_000_000):
    ts = np.arange(n, dtype=np.int64); val = np.random.default_rng(0).random(n)
    host = [capi.Column(ts), capi.Column(val)]
    dev = [c.to_device() for c in host]
    for label, cols, res in (("host cols, host outs", host, capi.HOST), ("device cols, device outs", dev, capi.DEVICE)):
        s0, W = capi.plan_windows(cols[0], 10, 0)
        outs = [capi.OutColumn(W, res) for _ in aggs]
        for _ in range(5): capi.rolling_aggregate(cols, 0, 10, aggs, outs=outs)
        t0 = time.perf_counter()
        reps = 200
        for _ in range(reps): capi.rolling_aggregate(cols, 0, 10, aggs, outs=outs)
        dt = (time.perf_counter() - t0) / reps
        print("n=%-8d %-26s %.1f us per call" % (n, label, dt * 1e6))
        # with the plan the host keeps from the constructor (newIntervalRolling computes it once): no round trip for first / last ts
        plan = capi.plan_windows_ex(cols[0], 10, 0)
        for _ in range(5): capi.rolling_aggregate(cols, 0, 10, aggs, outs=outs, plan=plan)
        t0 = time.perf_counter()
        for _ in range(reps): capi.rolling_aggregate(cols, 0, 10, aggs, outs=outs, plan=plan)
        dt = (time.perf_counter() - t0) / reps
        print("n=%-8d %-26s %.1f us per call (planned)" % (n, label, dt * 1e6))
