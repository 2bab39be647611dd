"""The callers either side of the path at 1e8 rows, five calls each, for the counter passes (scratch/pmc_sq.sh): whole-frame Aggregate
(dense and 30 % nulls: whole_value_kernel), IsColSorted (a column without nulls: col_order_dense_kernel; with nulls: col_order_kernel),
FillPrevious and FillLinear (fill_kernel)."""
import sys
sys.path.insert(0, '.')
from bow_amd import capi
n = 100_000_000
dts, dval = capi.gen_dense(0, n, seed=42)
ts, val = capi.gen_sparse(0, n, seed=42)
valid = capi.aggregate_whole([ts, val], 0, [("Count", 1)])[0].to_list()[0]
val = capi.Column(val.values, val.validity, capi.FLOAT64, 0, n, n - valid)
aggs = [("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1)]
for _ in range(5):
    capi.aggregate_whole([dts, dval], 0, aggs)
    capi.aggregate_whole([ts, val], 0, aggs)
    capi.is_col_sorted(ts)
    capi.is_col_sorted(val)
    capi.fill(val, "Previous", out_residency=capi.DEVICE)
    capi.fill_linear([ts, val], 0, 1, out_residency=capi.DEVICE)
capi.synchronize()
print("ok")
