"""Long windows (interval 1000, 1e8 rows) with reducer sets that select the different instantiations of long_stream_kernel.
`strict` as an argument: the same calls with bowgpu_options.strict_order (every window walked in row order: long_strict_kernel)."""
import gc, sys, time
sys.path.insert(0, '.')
from bow_amd import capi
strict = "strict" in sys.argv[1:]
n = 100_000_000
ts, val = capi.gen_dense(0, n, seed=42)
tss, vals = capi.gen_sparse(0, n, seed=3)
sets = {"Mean": [("WindowStart", 0), ("ArithmeticMean", 1)],
        "First+Last": [("WindowStart", 0), ("First", 1), ("Last", 1)],
        "Sum+Mean+Min+Max": [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1)],
        "WeightedAverageStep": [("WindowStart", 0), ("WeightedAverageStep", 1)],
        "all four time-weighted": [("WindowStart", 0), ("IntegralStep", 1), ("IntegralTrapezoid", 1), ("WeightedAverageStep", 1), ("WeightedAverageLinear", 1)]}
for label, cols, interval in (("dense", [ts, val], 1000), ("30% nulls", [tss, vals], 10000)):
    for name, aggs in sets.items():
        s0, W = capi.plan_windows(cols[0], interval, 0)
        outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
        for _ in range(2):
            capi.rolling_aggregate(cols, 0, interval, aggs, outs=outs, strict_order=strict)
        gc.collect(); capi.synchronize()     # (earlier output buffers are freed outside the timed calls)
        t0 = time.perf_counter()
        ms = []
        for _ in range(7):
            _, info = capi.rolling_aggregate(cols, 0, interval, aggs, outs=outs, strict_order=strict)
            ms.append(info.kernel_ms)
        capi.synchronize()
        dt = (time.perf_counter() - t0) / 7
        br = sorted(ms)[3]      # the median bracket of the seven calls (single calls are now and then 0.1 ms slower on a shared box)
        print("%-10s %-26s W=%-7d %s  bracket %.3f ms (max %.3f)  wall %.3f ms  %.1f Grows/s  %.1f%% of 8 TB/s" %
              (label, name, W, capi.last_kernel_name(), br, max(ms), dt * 1e3, n / dt / 1e9, n * 16 / (br * 1e-3) / 8e12 * 100))
