import numpy as np, sys
sys.path.insert(0,'.')
from bow_amd import capi
from oracle import pyoracle as orc
rng = np.random.default_rng(1)
for n, interval, offset in [(1, 10, 0), (7, 3, 1), (2047, 10, 0), (2048, 10, 3), (2049, 10, -4), (6000, 7, 8), (50_000, 10, 0)]:
    for c0 in (-37, 0, 5):
        ts = np.arange(n, dtype=np.int64) + c0
        v = rng.standard_normal(n)
        try:
            outs, info = capi.rolling_aggregate([capi.Column(ts), capi.Column(v)], 0, interval, [("WindowStart",0),("Sum",1)], offset=offset)
            exp,_ = orc.aggregate([orc.Column(ts), orc.Column(v)], 0, interval, [("WindowStart",0),("Sum",1)], offset=offset)
            ok = np.array_equal(outs[1].host_arrays()[0].view(np.uint64), exp[1].values[:exp[1].length].view(np.uint64))
            print(n, interval, offset, c0, "ok" if ok else "MISMATCH", info.s0, info.num_windows)
        except capi.BowGpuError as e:
            print(n, interval, offset, c0, "ERR", e.code, capi.plan_windows(capi.Column(ts), interval, offset))
