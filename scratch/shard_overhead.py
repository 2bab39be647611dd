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
print("  first_last_nrows      %.3f ms" % timeit(p.first_last_nrows))
sess = sharded.ShardSession(p, 0, 1, 10)
info = [sess.local_info()]
import numpy as np
f = int(np.frombuffer(info[0][:24], dtype=np.int64)[0])
s0 = sharded.first_window_start(f, 10, 3)
print("  local_info            %.3f ms" % timeit(sess.local_info))
print("  phase1 (aggregate)    %.3f ms" % timeit(lambda: sess.phase1(s0, info)))
c = sess.phase1(s0, info)
print("  phase2                %.3f ms" % timeit(lambda: sess.phase2([c])))
print("  shard_carry_only      %.3f ms" % timeit(lambda: p.shard_carry_only(s0, True)))
