"""1000-row windows over 1e8 rows with ONE reducer set (argv[1]: mean | firstlast | minmax | tw | tw4; argv[2]: dense | sparse), five calls
(for pmc_sq.sh / prof_any.sh)."""
import sys
sys.path.insert(0, '.')
from bow_amd import capi
sets = {"mean": [("WindowStart", 0), ("ArithmeticMean", 1)],
        "firstlast": [("WindowStart", 0), ("First", 1), ("Last", 1)],
        "minmax": [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1)],
        "tw": [("WindowStart", 0), ("WeightedAverageStep", 1)],
        "tw4": [("WindowStart", 0), ("IntegralStep", 1), ("IntegralTrapezoid", 1), ("WeightedAverageStep", 1), ("WeightedAverageLinear", 1)]}
aggs = sets[sys.argv[1] if len(sys.argv) > 1 else "tw"]
sparse = len(sys.argv) > 2 and sys.argv[2] == "sparse"
strict = len(sys.argv) > 3 and sys.argv[3] == "strict"   # bowgpu_options.strict_order: long_strict_kernel
n = 100_000_000
ts, val = (capi.gen_sparse if sparse else capi.gen_dense)(0, n, seed=42)
interval = 10000 if sparse else 1000
s0, W = capi.plan_windows(ts, interval, 0)
outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
for _ in range(5):
    _, info = capi.rolling_aggregate([ts, val], 0, interval, aggs, outs=outs, strict_order=strict)
print(capi.last_kernel_name(), info.long_windows, info.kernel_ms)
