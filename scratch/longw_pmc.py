"""One long-window shape for the counter passes of scratch/pmc_sq.sh: longw_pmc.py <Mean|WeightedAverageStep|TW4> <dense|sparse> - 1e8 rows, 1000 rows per
window, a few calls (the streaming form: long_short_kernel + long_stream_kernel + stream_final_kernel)."""
import sys
sys.path.insert(0, '.')
from bow_amd import capi
name, data = sys.argv[1], sys.argv[2]
sets = {"Mean": [("WindowStart", 0), ("ArithmeticMean", 1)],
        "WeightedAverageStep": [("WindowStart", 0), ("WeightedAverageStep", 1)],
        "TW4": [("WindowStart", 0), ("IntegralStep", 1), ("IntegralTrapezoid", 1), ("WeightedAverageStep", 1), ("WeightedAverageLinear", 1)]}
n = 100_000_000
cols = list(capi.gen_dense(0, n, seed=42)) if data == "dense" else list(capi.gen_sparse(0, n, seed=3))
interval = 1000 if data == "dense" else 10000
s0, W = capi.plan_windows(cols[0], interval, 0)
outs = [capi.OutColumn(W, capi.DEVICE) for _ in sets[name]]
for _ in range(4):
    _, info = capi.rolling_aggregate(cols, 0, interval, sets[name], outs=outs)
capi.synchronize()
print(name, data, capi.last_kernel_name(), "bracket %.3f ms" % info.kernel_ms)
