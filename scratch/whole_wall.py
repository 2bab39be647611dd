"""aggregation.Aggregate over the whole frame (bowgpu_aggregate_whole) at 1e8 rows: wall per call by reducer set, dense and with 30 % nulls"""
import sys, time
sys.path.insert(0, '.')
from bow_amd import capi
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
def med(fn, reps=9):
    fn(); capi.synchronize()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); capi.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
    return sorted(t)[len(t) // 2]
for label, (ts, val) in (("dense", capi.gen_dense(0, n, seed=42)), ("30 % nulls", capi.gen_sparse(0, n, seed=42))):
    for name, aggs in (("Count", [("Count", 1)]), ("Sum+Mean+Min+Max", [("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1)]),
                       ("First+Last", [("First", 1), ("Last", 1)]), ("WeightedAverageStep", [("WeightedAverageStep", 1)]),
                       ("all four time-weighted + Mean", [("IntegralStep", 1), ("IntegralTrapezoid", 1), ("WeightedAverageStep", 1), ("WeightedAverageLinear", 1), ("ArithmeticMean", 1)])):
        w = med(lambda: capi.aggregate_whole([ts, val], 0, aggs))
        need_ts = any(a[0].startswith(("Integral", "Weighted")) for a in aggs)
        gb = n * (16.125 if need_ts else 8.125) / 1e9 if label != "dense" else n * (16 if need_ts else 8) / 1e9
        print("%-11s %-32s wall %.3f ms per call  %6.1f G rows/s  %.2f of 8 TB/s on %.2f GB" % (label, name, w, n / w / 1e6, gb / w / 8, gb))
