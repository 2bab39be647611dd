"""ONE shape of bowgpu_rolling_interpolate_aggregate for counter passes (scratch/pmc_any.sh): configs[2]'s frame (irregular rows, 30 % nulls,
interval 100), argv: rows offset set(2|4|7 reducers) [plain = the same rows through bowgpu_rolling_aggregate, no interpolation]"""
import sys
sys.path.insert(0, '.')
from bow_amd import capi
n = int(float(sys.argv[1])); offset = int(sys.argv[2]); nset = int(sys.argv[3]); plain = len(sys.argv) > 4 and sys.argv[4] == "plain"
ts, val = capi.gen_sparse(0, n, seed=42)
valid = capi.aggregate_whole([ts, val], 0, [("Count", 1)])[0].to_list()[0]
val = capi.Column(val.values, val.validity, capi.FLOAT64, 0, n, n - valid)
SETS = {2: [("WindowStart", 0), ("ArithmeticMean", 1)], 4: [("WindowStart", 0), ("ArithmeticMean", 1), ("Count", 1), ("Min", 1)],
        7: [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1), ("First", 1), ("Last", 1)]}
aggs = SETS[nset]
ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
W = capi.plan_windows(ts, 100, offset)[1]
outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
ms = []
for _ in range(6):
    if plain:
        _, info = capi.rolling_aggregate([ts, val], 0, 100, aggs, offset=offset, outs=outs)
    else:
        _, info = capi.rolling_interpolate_aggregate([ts, val], 0, 100, ip, aggs, offset=offset, outs=outs)
    ms.append(info.kernel_ms)
print("%s off=%d %d reducers: kernel %s median %.3f ms" % ("plain Aggregate" if plain else "Interpolate -> Aggregate", offset, nset, capi.last_kernel_name(), sorted(ms)[3]))
