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
import numpy as np
def known(cols):
    """the column's null count as a Bow hands it over (Data().NullN(), bowseries.go:67): counted once here, not by every call"""
    ts, val = cols
    bits = val.validity.to_numpy(np.uint8, (n + 7) // 8)
    val.null_count = int(n - int(np.unpackbits(bits, bitorder="little")[:n].sum()))
    return ts, val
for label, (ts, val) in (("dense", capi.gen_dense(0, n, seed=42)), ("30 % nulls", known(capi.gen_sparse(0, n, seed=42))), ("30 %, count unknown", capi.gen_sparse(0, n, seed=42))):
    for name, aggs in (("Count", [("Count", 1)]), ("Sum+Mean+Min+Max", [("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1)]),
                       ("First+Last", [("First", 1), ("Last", 1)]), ("WeightedAverageStep", [("WeightedAverageStep", 1)]),
                       ("all four time-weighted + Mean", [("IntegralStep", 1), ("IntegralTrapezoid", 1), ("WeightedAverageStep", 1), ("WeightedAverageLinear", 1), ("ArithmeticMean", 1)])):
        # the library call alone: descriptors and the one-slot output buffers are built once (what a cgo caller's stack frame holds),
        # not per call by this script's Python (numpy allocations and ctypes marshalling were 20 - 30 us of round 5's figures)
        import ctypes as C
        carr, aarr = capi._cols([ts, val]), capi._aggs(aggs)
        outs = [capi.OutColumn(1) for _ in aggs]
        oarr = (capi.Out * len(aggs))()
        for i, o in enumerate(outs):
            oarr[i] = o.c()
        L = capi.lib()
        def call():
            for i in range(len(aggs)):
                oarr[i].length = 1
            capi.check(L.bowgpu_aggregate_whole(carr, 2, 0, aarr, len(aggs), oarr))
        w = med(call)
        need_ts = any(a[0].startswith(("Integral", "Weighted")) for a in aggs)
        gb = n * (16.125 if need_ts else 8.125) / 1e9 if label != "dense" else n * (16 if need_ts else 8) / 1e9
        print("%-19s %-32s wall %.3f ms per call  %6.1f G rows/s  %.2f of 8 TB/s on %.2f GB" % (label, name, w, n / w / 1e6, gb / w / 8, gb))
