"""The mid-window sweep's shapes checked against the oracle at scale (default 2e7 rows; scratch/midw_sweep.py times them at 1e8 and
checks nothing): regular rows without nulls and irregular rows with 30 % nulls, 16 .. 128 rows per window, every reducer in one call
and the sets of the sweep on their own - bit for bit (the tile kernels serve all of these: long_windows == 0)."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from bow_amd import capi
from oracle import pyoracle as orc
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
dts, dval = orc.gen_dense(0, n, seed=42)
sts, sval, sbm = orc.gen_sparse(0, n, seed=3)
frames = {"dense": ([capi.Column(dts).to_device(), capi.Column(dval).to_device()], [orc.Column(dts, None, orc.INT64), orc.Column(dval, None, orc.FLOAT64)], 1),
          "sparse": ([capi.Column(sts).to_device(), capi.Column(sval, sbm, capi.FLOAT64, 0, n, -1).to_device()],
                     [orc.Column(sts, None, orc.INT64), orc.Column(sval, sbm, orc.FLOAT64)], 10)}
sets = {"all": [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1), ("Count", 1), ("First", 1), ("Last", 1), ("NumRows", 1),
                ("IntegralStep", 1), ("IntegralTrapezoid", 1), ("WeightedAverageStep", 1), ("WeightedAverageLinear", 1)],
        "Mean": [("WindowStart", 0), ("ArithmeticMean", 1)],
        "MinMax": [("WindowStart", 0), ("Min", 1), ("Max", 1)],
        "SumMinMax": [("WindowStart", 0), ("Sum", 1), ("Min", 1), ("Max", 1)],
        "WAvgStep": [("WindowStart", 0), ("WeightedAverageStep", 1)],
        "TW4": [("WindowStart", 0), ("IntegralStep", 1), ("IntegralTrapezoid", 1), ("WeightedAverageStep", 1), ("WeightedAverageLinear", 1)]}
bad = 0
for label, (ccols, ocols, scale) in frames.items():
    for rpw in (16, 24, 32, 48, 64, 96, 127):
        interval = rpw * scale
        for name, aggs in sets.items():
            t0 = time.perf_counter()
            want, _ = orc.aggregate(ocols, 0, interval, aggs)
            got, info = capi.rolling_aggregate(ccols, 0, interval, aggs)
            assert info.long_windows == 0, (label, rpw, name, info.long_windows)
            for (k, _c), g, w in zip(aggs, got, want):
                gv, gb = g.host_arrays()
                gm, wm = g.valid_mask(), w.valid_mask()
                ok = g.length == w.length and np.array_equal(gm, wm) and np.array_equal(gv.view(np.uint64)[gm], w.values[:w.length].view(np.uint64)[wm])
                if not ok:
                    bad += 1
                    print("MISMATCH", label, rpw, name, k)
            print("%-6s %3d rows/window %-9s %s ok (%.1f s)" % (label, rpw, name, capi.last_kernel_name(), time.perf_counter() - t0), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
