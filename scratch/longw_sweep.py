"""Which form for which window length: 1e8 rows, window lengths 64 .. 1e6 rows, three reducer sets, the routes of api.cpp job_run
(auto / streaming for every set / bisection form / tile kernels + cooperative path).  Prints bracket / wall per call in ms (the bracket of the tile route covers the tile kernel only)."""
import gc, os, sys, time
sys.path.insert(0, '.')
from bow_amd import capi
n = 100_000_000
dense = capi.gen_dense(0, n, seed=42)
sparse = capi.gen_sparse(0, n, seed=3)
sets = {"Mean": [("WindowStart", 0), ("ArithmeticMean", 1)],
        "MinMax": [("WindowStart", 0), ("Sum", 1), ("Min", 1), ("Max", 1)],
        "WAvgStep": [("WindowStart", 0), ("WeightedAverageStep", 1)]}
routes = (("auto", 0), ("stream", capi.ROUTE_LONG_STREAM_ALL), ("bisect", capi.ROUTE_LONG_CLASSIC), ("tiles", capi.ROUTE_NO_LONG_ONLY))
only = sys.argv[1:] or ["dense", "sparse"]
for label, cols, scale in (("dense", dense, 1), ("sparse", sparse, 10)):
    if label not in only:
        continue
    for rows_per_window in [int(x) for x in os.environ.get("SWEEP_ROWS", "64,128,256,512,1000,4000,32768,262144,1000000").split(",")]:
        interval = rows_per_window * scale
        for name, aggs in sets.items():
            s0, W = capi.plan_windows(cols[0], interval, 0)
            outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
            line = "%-6s %8d rows/window %-9s" % (label, rows_per_window, name)
            for rname, mask in routes:
                with capi.route(mask):
                    try:
                        for _ in range(2):
                            capi.rolling_aggregate(list(cols), 0, interval, aggs, outs=outs)
                        ms = []
                        capi.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(4):
                            _, info = capi.rolling_aggregate(list(cols), 0, interval, aggs, outs=outs)
                            ms.append(info.kernel_ms)
                        capi.synchronize()
                        wall = (time.perf_counter() - t0) / 4 * 1e3
                        line += "  %s %.3f/%.3f (%s)" % (rname, sorted(ms)[1], wall, capi.last_kernel_name().replace("_kernel", "").replace("rolling_", "r_").replace("long_", "l_"))
                    except Exception as e:
                        line += "  %s ERR %s" % (rname, str(e)[:30])
            print(line, flush=True)
            del outs
            gc.collect()
