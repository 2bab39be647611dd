"""PCIe-inclusive rate: the same Rolling.Aggregate call with HOST-resident (pageable numpy) columns, as a cgo caller holding Go-heap
Arrow buffers would make it.  Reported in DESIGN.md §6; never bench.py's `value`."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from bow_amd import capi
n = 100_000_000
ts = np.arange(n, dtype=np.int64)
val = np.random.default_rng(0).random(n)
aggs = [("WindowStart", 0), ("ArithmeticMean", 1)]
for rep in range(3):
    t0 = time.perf_counter()
    outs, info = capi.rolling_aggregate([capi.Column(ts), capi.Column(val)], 0, 10, aggs)
    t1 = time.perf_counter()
    print("host-resident columns + host outputs: %d rows  %.1f ms  %.2f Grows/s  (%.1f GB/s over PCIe incl. allocation)  kernel %.3f ms"
          % (n, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9, (16 * n + 1.6 * n) / (t1 - t0) / 1e9, info.kernel_ms))
