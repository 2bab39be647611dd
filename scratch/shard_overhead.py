"""Fixed cost of the sharded protocol on one GPU (world = 1: no collectives): ShardedRolling.step() next to the unsharded call."""
import sys, time
sys.path.insert(0, '.')
import torch
from bow_amd import capi, sharded
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
aggs = [("WindowStart", 0), ("ArithmeticMean", 1)]
r = sharded.ShardedRolling(0, 1, rows, 10, aggs, None, torch, offset=3)
def timeit(fn, reps=10):
    for _ in range(3): fn()
    capi.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    capi.synchronize(); return (time.perf_counter() - t) / reps * 1e3
print("sharded step (world 1): %.3f ms" % timeit(r.step))
ts, val = r.cols
s0, W = capi.plan_windows(ts, 10, 3)
outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
print("unsharded call:         %.3f ms" % timeit(lambda: capi.rolling_aggregate([ts, val], 0, 10, aggs, offset=3, outs=outs)))
p = r.provider
print("  begin (record)        %.3f ms" % timeit(p.begin))
rec = [p.begin()]
print("  finish (pass+stitch)  %.3f ms" % timeit(lambda: p.finish(rec, 0)))
