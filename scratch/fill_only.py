"""FillPrevious / FillLinear / IsColSorted at 1e8 rows of gen_sparse data, three calls each (for pmc_any.sh)."""
import sys
sys.path.insert(0, '.')
from bow_amd import capi
n = 100_000_000
ts, val = capi.gen_sparse(0, n, seed=42)
for _ in range(3):
    out, _ = capi.fill(val, "Previous", out_residency=capi.DEVICE)
    out, _ = capi.fill_linear([ts, val], 0, 1, out_residency=capi.DEVICE)
    capi.is_col_sorted(ts)
capi.synchronize()
print("ok")
