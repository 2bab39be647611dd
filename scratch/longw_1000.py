"""1000-row windows over 1e8 dense rows (the long-only pipeline), five calls (for prof_any.sh)."""
import sys
sys.path.insert(0, '.')
from bow_amd import capi
n = 100_000_000
ts, val = capi.gen_dense(0, n, seed=42)
aggs = [("WindowStart", 0), ("ArithmeticMean", 1)]
s0, W = capi.plan_windows(ts, 1000, 0)
outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
for _ in range(5):
    _, info = capi.rolling_aggregate([ts, val], 0, 1000, aggs, outs=outs)
print(info.long_windows, info.kernel_ms)
