"""ONE bowgpu_rolling_aggregate call over the device list (bowgpu_set_devices; bow_amd/csrc/multi.cpp) at 1e8 rows, interval 10, WindowStart +
ArithmeticMean: wall per call by residency and by number of ranks.  On a one-GPU box device 0 is listed N times - the ranks share ONE host
link and ONE device, so this shows what the fan-out COSTS (threads, records, staging, the stitch into the caller's buffers), not what
eight links buy; on a node with several GPUs the list names them (argv[1] = "all")."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from bow_amd import capi
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
ndev = capi.device_count()
dts, dval = capi.gen_dense(0, n, seed=42)
ts = capi.page_aligned(n, np.int64); ts[:] = dts.values.to_numpy(np.int64, n)
val = capi.page_aligned(n, np.float64); val[:] = dval.values.to_numpy(np.float64, n)
aggs = [("WindowStart", 0), ("ArithmeticMean", 1)]
W = n // 10
def best(fn, reps=4):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    return min(t)
ref = None
for label, cols, res in (("device-resident", [dts, dval], capi.DEVICE), ("pageable host", [capi.Column(ts), capi.Column(val)], capi.HOST),
                         ("registered host", [capi.Column(ts).pin(), capi.Column(val).pin()], capi.HOST_PINNED)):
    outs = [capi.OutColumn(W, res) for _ in aggs]
    for ranks in (1, 2, 4, 8):
        ids = [] if ranks == 1 else ([i % ndev for i in range(ranks)] if (len(sys.argv) > 1 and sys.argv[1] == "all" and ndev > 1) else [0] * ranks)
        with capi.devices(ids):
            dt = best(lambda: capi.rolling_aggregate(cols, 0, 10, aggs, outs=outs))
            got = capi.last_call_ranks()
        chk = capi.checksum64(outs[1].values, W) if res == capi.DEVICE else int(np.bitwise_xor.reduce(outs[1].values[:W].view(np.uint64)))
        if ranks == 1:
            ref = chk
        print("%-16s devices %-26s ranks %d: %8.3f ms per call  %7.2f G rows/s  outputs %s" %
              (label, ids or "[one-device path]", got, dt * 1e3, n / dt / 1e9, "identical" if chk == ref else "DIFFER"), flush=True)
    del outs
