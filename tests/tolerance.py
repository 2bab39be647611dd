"""The stated float64 tolerance of the ORDER-FREE forms (include/bowgpu.h "THE STATED TOLERANCE"), as a bound
the tests assert - not a relative error, which no reordering of a sum can promise under cancellation.

A window the library reduces order-free (bowgpu_agg_info.long_windows > 0: windows longer than a tile's look-ahead, every
window of a long-only call, a window that three or more shards share) sums the SAME terms as the reference's left-to-right
loop in another, fixed association.  With u = 2^-53, n the window's rows and T_i its terms (the valid values for Sum /
ArithmeticMean; v_j * dt_j for IntegralStep; (v_j + v_j+1) / 2 * dt_j for IntegralTrapezoid):

    |Sum_gpu - Sum_ref|            <= 2 (n + 2) u  sum_i |x_i|
    |Mean_gpu - Mean_ref|          <= that / count + 2 u |Mean_ref|
    |Integral_gpu - Integral_ref|  <= 4 (n + 2) u  sum_i |T_i|
    |WeightedAverage_gpu - ..ref|  <= that / float64(LastValue - FirstValue) + 2 u |ref|

(both sums are within (n - 1) u sum|T_i| of the exact sum - Higham, "Accuracy and Stability of Numerical Algorithms", (4.4); the
factors 2 and 4 cover the second-order terms, the rounding of each product and of the final division).  A transformation.Factor chain
scales the bound by the product of |factors| and adds one rounding per factor.  Everything else - Count, Min, Max, First, Last,
WindowStart, NumRows, validity, null counts, and EVERY reducer when long_windows == 0 - is bit-exact.

sum_i |T_i| per window comes from the oracle itself, run on the columns' absolute values (the reference's algorithm over |x| has
no cancellation: its own rounding error is at most n u relative, covered by the factors above)."""
import numpy as np

from oracle import pyoracle as orc

U = 2.0 ** -53
ORDER_SENSITIVE = {"Sum", "ArithmeticMean", "IntegralStep", "IntegralTrapezoid", "WeightedAverageStep", "WeightedAverageLinear"}
_MAG_KIND = {"Sum": "Sum", "ArithmeticMean": "Sum", "IntegralStep": "IntegralStep", "WeightedAverageStep": "IntegralStep",
             "IntegralTrapezoid": "IntegralTrapezoid", "WeightedAverageLinear": "IntegralTrapezoid"}


def abs_columns(ocols, ts_col):
    """the frame with every value column replaced by its absolute values; the interval column stays (it defines the windows) and
    its absolute values - reducers may read the interval column itself - come as one more column at index len(ocols)"""
    out = []
    for i, c in enumerate(list(ocols) + [ocols[ts_col]]):
        if i == ts_col:
            out.append(c)
            continue
        v = np.abs(c.values.astype(np.float64))          # (Int64 columns are read as float64(v): bowgetters.go:227-229)
        out.append(orc.Column(v, c.validity, orc.FLOAT64, c.offset, c.length))
    return out


def order_free_bounds(ocols, ts_col, interval, aggs, offset=0, inclusive=False, ref=None):
    """per aggregator: None (bit-exact kind) or an array of W absolute bounds.  ref: the oracle's outputs for `aggs` (list of
    orc.Column), needed for the |ref| terms of Mean / WeightedAverage / Factor chains."""
    kinds = [a[0] for a in aggs]
    if not any(k in ORDER_SENSITIVE for k in kinds):
        return [None] * len(aggs)
    if any(k in ("IntegralTrapezoid", "WeightedAverageLinear") for k in kinds):
        inclusive = True                                  # aggregation.go:183-185
    acols = abs_columns(ocols, ts_col)
    src = lambda col: len(ocols) if col == ts_col else col    # noqa: E731  (where a column's absolute values are)
    mag_aggs, where = [("WindowStart", ts_col)], {}
    for k, col in sorted({(_MAG_KIND[a[0]], a[1]) for a in aggs if a[0] in ORDER_SENSITIVE}):
        where[(k, col)] = len(mag_aggs)
        mag_aggs.append((k, src(col)))
    for col in sorted({a[1] for a in aggs if a[0] in ORDER_SENSITIVE}):
        where[("Count", col)] = len(mag_aggs)
        mag_aggs.append(("Count", src(col)))
        where[("NumRows", col)] = len(mag_aggs)
        mag_aggs.append(("NumRows", src(col)))
    mags, _ = orc.aggregate(acols, ts_col, interval, mag_aggs, offset=offset, inclusive=inclusive)
    out = []
    for i, a in enumerate(aggs):
        k, col = a[0], a[1]
        if k not in ORDER_SENSITIVE:
            out.append(None)
            continue
        W = mags[0].length
        m = mags[where[(_MAG_KIND[k], col)]]
        S = np.where(m.valid_mask(), m.values[:W], 0.0)
        n = mags[where[("NumRows", col)]].values[:W] + 1.0      # (+ the inclusive row)
        cnt = np.maximum(mags[where[("Count", col)]].values[:W].astype(np.float64), 1.0)
        r = np.zeros(W) if ref is None else np.where(ref[i].valid_mask(), np.abs(ref[i].values[:W].astype(np.float64)), 0.0)
        r = np.where(np.isfinite(r), r, 0.0)
        c = 2.0 if _MAG_KIND[k] == "Sum" else 4.0
        b = c * (n + 2.0) * U * S
        if k == "ArithmeticMean":
            b = b / cnt
        elif k in ("WeightedAverageStep", "WeightedAverageLinear"):
            b = b / float(interval)
        factors = list(a[2]) if len(a) > 2 and a[2] else []
        scale = float(np.prod(np.abs(factors))) if factors else 1.0
        if k in ("ArithmeticMean", "WeightedAverageStep", "WeightedAverageLinear") or factors:
            b = b * scale + (2.0 + len(factors)) * U * r
        else:
            b = b * scale
        out.append(b)
    return out


def assert_within(name, g, w, bound):
    """g, w: float64 arrays of the valid slots; bound: absolute bound per slot (same selection)"""
    g, w, bound = np.asarray(g, np.float64), np.asarray(w, np.float64), np.asarray(bound, np.float64)
    same = (g == w) | (np.isnan(g) & np.isnan(w))
    with np.errstate(invalid="ignore"):
        d = np.abs(g - w)
    # a window whose terms overflow to +-inf / hold NaN has no finite bound: both sides must then agree in kind
    finite = np.isfinite(bound) & np.isfinite(w) & np.isfinite(g)
    bad = ~same & finite & (d > bound)
    kind = ~same & ~finite & ~((np.isnan(g) & np.isnan(w)) | (np.isinf(g) & np.isinf(w) & (np.sign(g) == np.sign(w))) | ~np.isfinite(bound))
    idx = np.flatnonzero(bad | kind)
    assert idx.size == 0, (name, idx[:5], g[idx[:5]], w[idx[:5]], bound[idx[:5]])
