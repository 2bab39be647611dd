"""PCIe-inclusive rates: the same Rolling.Aggregate call with host-resident columns, as a cgo caller holding Arrow buffers would make it -
pageable memory (Go heap / malloc: staged through HBM by the runtime's pageable copy), registered memory read in place by the kernels
(BOWGPU_HOST_PINNED: zero-copy), registered memory staged by DMA (ROUTE_PINNED_STAGE).  Reported in DESIGN.md section 6 and as
bench.py's `host_pinned` key; never bench.py's `value`."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from bow_amd import capi
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ts = np.arange(n, dtype=np.int64)
val = np.random.default_rng(0).random(n)
aggs = [("WindowStart", 0), ("ArithmeticMean", 1)]
s0, W = capi.plan_windows(capi.Column(ts), 10, 0)

def run(label, cols, outs, reps=4):
    best = None
    for rep in range(reps):
        t0 = time.perf_counter()
        _, info = capi.rolling_aggregate(cols, 0, 10, aggs, outs=outs)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    print("%-72s %d rows  %.1f ms  %.2f G rows/s  (%.1f GB/s over PCIe)  kernel %.3f ms"
          % (label, n, best * 1e3, n / best / 1e9, (16 * n + 1.6 * n) / best / 1e9, info.kernel_ms))
    return info

run("pageable columns, pageable outputs", [capi.Column(ts), capi.Column(val)], [capi.OutColumn(W, capi.HOST) for _ in aggs])
t0 = time.perf_counter()
cols = [capi.Column(ts).pin(), capi.Column(val).pin()]
outs = [capi.OutColumn(W, capi.HOST_PINNED) for _ in aggs]
print("registering 2 x %.1f GB + outputs: %.1f ms" % (8 * n / 1e9, (time.perf_counter() - t0) * 1e3))
info = run("registered columns read in place (zero-copy), registered outputs", cols, outs)
want = [o.host_arrays()[0].copy() for o in outs]
capi.set_route(capi.ROUTE_PINNED_STAGE)
run("registered columns staged by DMA (ROUTE_PINNED_STAGE), registered outputs", cols, outs)
capi.set_route(0)
for w, o in zip(want, outs):
    assert np.array_equal(w.view(np.uint64), o.host_arrays()[0].view(np.uint64))
dev = [c.to_device() for c in [capi.Column(ts), capi.Column(val)]]
ref, _ = capi.rolling_aggregate(dev, 0, 10, aggs, out_residency=capi.DEVICE)
for w, o in zip(want, ref):
    assert np.array_equal(w.view(np.uint64), o.host_arrays()[0].view(np.uint64))
print("zero-copy == staged == device-resident results: ok")
