"""Where the time of a call with both kinds of integral goes: 1e8 regular rows (or, `sparse` as the first argument, irregular rows with
30 % nulls), 64 rows per window, reducer sets from one kind to all four."""
import sys
sys.path.insert(0, '.')
from bow_amd import capi
n = 100_000_000
sparse = len(sys.argv) > 1 and sys.argv[1] == "sparse"
if sparse:
    sys.argv.pop(1)
cols = capi.gen_sparse(0, n, seed=3) if sparse else capi.gen_dense(0, n, seed=42)
scale = 10 if sparse else 1
sets = {"WAvgStep": ["WeightedAverageStep"], "WAvgLinear": ["WeightedAverageLinear"], "IStep+WAvgStep": ["IntegralStep", "WeightedAverageStep"],
        "ITrap+WAvgLinear": ["IntegralTrapezoid", "WeightedAverageLinear"], "IStep+ITrap": ["IntegralStep", "IntegralTrapezoid"],
        "all four": ["IntegralStep", "IntegralTrapezoid", "WeightedAverageStep", "WeightedAverageLinear"]}
for rpw in (int(a) for a in (sys.argv[1:] or ["64"])):
    for name, kinds in sets.items():
        aggs = [("WindowStart", 0)] + [(k, 1) for k in kinds]
        s0, W = capi.plan_windows(cols[0], rpw * scale, 0)
        outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
        ms = []
        for _ in range(6):
            _, info = capi.rolling_aggregate(list(cols), 0, rpw * scale, aggs, outs=outs)
            ms.append(info.kernel_ms)
        print("%4d rows/window %-18s %d outputs  %.3f ms  %.2f of 8 TB/s  (%s)" % (rpw, name, len(aggs), sorted(ms[1:])[2], n * 16 / (sorted(ms[1:])[2] * 1e-3) / 8e12, capi.last_kernel_name()))
