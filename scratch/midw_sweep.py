"""The 16 .. 256 rows-per-window band: 1e8 rows, five reducer sets, dense (regular, no nulls) and sparse
(irregular timestamps, 30 % nulls), default route - plus, from 96 rows on, the streaming form and the tile kernels forced, to show
where the threshold belongs.  Prints the bracket of all kernels of a call in ms, the fraction of 8 TB/s that is (algorithmic read
bytes: 16 B per row, + 1/8 B per row for a nullable column) and the kernel that ran."""
import gc, os, sys, time
sys.path.insert(0, '.')
from bow_amd import capi
n = int(float(os.environ.get("SWEEP_N", "1e8")))
dense = capi.gen_dense(0, n, seed=42)
sparse = capi.gen_sparse(0, n, seed=3)
sets = {"Mean": [("WindowStart", 0), ("ArithmeticMean", 1)],
        "MinMax": [("WindowStart", 0), ("Min", 1), ("Max", 1)],
        "SumMinMax": [("WindowStart", 0), ("Sum", 1), ("Min", 1), ("Max", 1)],
        "FirstLast": [("WindowStart", 0), ("First", 1), ("Last", 1)],
        "WAvgStep": [("WindowStart", 0), ("WeightedAverageStep", 1)],
        "TW4": [("WindowStart", 0), ("IntegralStep", 1), ("IntegralTrapezoid", 1), ("WeightedAverageStep", 1), ("WeightedAverageLinear", 1)]}
only = [a for a in sys.argv[1:] if a in ("dense", "sparse")] or ["dense", "sparse"]
want_sets = [a for a in sys.argv[1:] if a in sets] or list(sets)
rows_list = [int(x) for x in os.environ.get("SWEEP_ROWS", "16,24,32,48,64,96,128,192,256").split(",")]
worst = {}
for label, cols, scale, bpr in (("dense", dense, 1, 16.0), ("sparse", sparse, 10, 16.125)):
    if label not in only:
        continue
    for rpw in rows_list:
        interval = rpw * scale
        for name in want_sets:
            aggs = sets[name]
            s0, W = capi.plan_windows(cols[0], interval, 0)
            outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
            routes = [("auto", 0)]
            if label == "sparse" and os.environ.get("SWEEP_ROUTES", "1") == "1":
                routes += [("row-space", capi.ROUTE_TW_ROWS)]   # round 5: without rolling_twc_kernel (the forms of round 4: rolling_tw / rolling_simple / streaming)
            if rpw >= 96 and os.environ.get("SWEEP_ROUTES", "1") == "1":
                routes += [("stream", capi.ROUTE_LONG_STREAM_ALL), ("tiles", capi.ROUTE_NO_LONG_ONLY)]
                if label == "sparse" and os.environ.get("SWEEP_TILES_ROWS") == "1":   # round 6: the tile route WITHOUT rolling_twc_kernel (rolling_simple / rolling_tw + the queue launch) on nullable columns
                    routes += [("tiles-rows", capi.ROUTE_NO_LONG_ONLY | capi.ROUTE_TW_ROWS)]
                if os.environ.get("SWEEP_HOSTQ") == "1":   # round 6 A/B: the tile route with the queued windows through the host (rounds 1 - 5) instead of long_queue_kernel
                    routes += [("tiles-hostq", capi.ROUTE_NO_LONG_ONLY | capi.ROUTE_QUEUE_HOST)]
            line = "%-6s %4d rows/window %-9s" % (label, rpw, name)
            for rname, mask in routes:
                with capi.route(mask):
                    try:
                        for _ in range(2):
                            capi.rolling_aggregate(list(cols), 0, interval, aggs, outs=outs)
                        ms = []
                        capi.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(5):
                            _, info = capi.rolling_aggregate(list(cols), 0, interval, aggs, outs=outs)
                            ms.append(info.kernel_ms)
                        capi.synchronize()
                        wall = (time.perf_counter() - t0) / 5 * 1e3
                        k = sorted(ms)[2]
                        frac = n * bpr / (k * 1e-3) / 8e12
                        kn = capi.last_kernel_name().replace("_kernel", "").replace("rolling_", "r_").replace("long_", "l_")
                        line += "  %s %.3f ms %.2f (%s, long %d, wall %.3f)" % (rname, k, frac, kn, info.long_windows, wall)
                        if rname == "auto":
                            worst[(label, name)] = min(worst.get((label, name), 9.0), frac)
                    except Exception as e:
                        line += "  %s ERR %s" % (rname, str(e)[:40])
            print(line, flush=True)
            del outs
            gc.collect()
print("worst cell per (data, set) on the default route:", ", ".join("%s/%s %.2f" % (k[0], k[1], v) for k, v in sorted(worst.items())))
