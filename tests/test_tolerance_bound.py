"""The stated tolerance of the order-free forms (tests/tolerance.py, include/bowgpu.h) is a bound in terms of sum|x_i|, not a
relative error.  CPU checks of the bound itself: a reassociated sum (numpy's pairwise tree, a reversed loop, a blocked tree like
the device's lanes -> trips -> chunks) of the SAME terms stays inside it against the oracle's left-to-right sum - on cancelling
data, where the relative error is unbounded - and a wrong result does not."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from tolerance import assert_within, order_free_bounds


def _frame(rng, n, kind):
    ts = np.arange(n, dtype=np.int64)
    if kind == "cancelling":      # 1e16-sized terms that cancel to O(100) per 20 000-row window: the relative error of a reordered sum is unbounded
        v = np.empty(n)
        for a in range(0, n, 20_000):
            big = rng.standard_normal(10_000) * 1e16
            v[a:a + 10_000] = big
            v[a + 10_000:a + 20_000] = -rng.permutation(big) + rng.standard_normal(10_000)
        return ts, v, np.ones(n, bool)
    elif kind == "mixed":
        v = rng.standard_normal(n) * 10.0 ** rng.integers(-8, 9, n)
    else:
        v = rng.random(n)
    valid = rng.random(n) > 0.2
    return ts, v, valid


def _blocked_tree_sum(x):
    """lanes, then trips, then chunks: the shape of the device's fixed summation tree"""
    x = np.concatenate([x, np.zeros((-len(x)) % 512)])
    lanes = x.reshape(-1, 8, 64)
    acc = lanes[:, 0, :].copy()
    for t in range(1, 8):
        acc = acc + lanes[:, t, :]
    w = 64
    while w > 1:
        w //= 2
        acc = acc[:, :w] + acc[:, w:2 * w]
    tot = 0.0
    for c in acc[:, 0]:
        tot = tot + c
    return tot


@pytest.mark.parametrize("kind", ["cancelling", "mixed", "uniform"])
def test_reassociated_sums_stay_inside_the_stated_bound(kind):
    rng = np.random.default_rng({"cancelling": 1, "mixed": 2, "uniform": 3}[kind])
    n, interval = 60_000, 20_000
    ts, v, valid = _frame(rng, n, kind)
    bm = np.packbits(valid, bitorder="little")
    ocols = [orc.Column(ts, None, orc.INT64), orc.Column(v, bm, orc.FLOAT64)]
    aggs = [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Sum", 1, [0.5, -3.0])]
    ref, _ = orc.aggregate(ocols, 0, interval, aggs)
    tol = order_free_bounds(ocols, 0, interval, aggs, ref=ref)
    assert tol[0] is None and all(t is not None for t in tol[1:])
    W = ref[0].length
    for form in ("pairwise", "reversed", "blocked"):
        sums = np.empty(W)
        for k in range(W):
            x = v[k * interval:(k + 1) * interval][valid[k * interval:(k + 1) * interval]]
            sums[k] = {"pairwise": lambda: float(np.sum(x)), "reversed": lambda: float(np.cumsum(x[::-1])[-1]) if len(x) else 0.0,
                       "blocked": lambda: float(_blocked_tree_sum(x))}[form]()
        cnt = np.array([valid[k * interval:(k + 1) * interval].sum() for k in range(W)], dtype=np.float64)
        assert_within((kind, form, "Sum"), sums, ref[1].values[:W], tol[1])
        assert_within((kind, form, "Mean"), sums / cnt, ref[2].values[:W], tol[2])
        assert_within((kind, form, "Sum x factors"), sums * 0.5 * -3.0, ref[3].values[:W], tol[3])
    if kind == "cancelling":
        # what the old "1e-11 relative" statement could not promise: the reordered sum is far outside it here, and inside the bound
        rel = np.abs(sums - ref[1].values[:W]) / np.abs(ref[1].values[:W])
        assert rel.max() > 1e-11
    # ... and the bound is not vacuous: an error of one part in 1e9 of sum|x| is caught
    S = np.array([np.abs(v[k * interval:(k + 1) * interval][valid[k * interval:(k + 1) * interval]]).sum() for k in range(W)])
    with pytest.raises(AssertionError):
        assert_within("must fail", ref[1].values[:W] + 1e-9 * S, ref[1].values[:W], tol[1])


def test_time_weighted_bounds_cover_a_reordered_integral():
    rng = np.random.default_rng(7)
    n, interval = 30_000, 10_000
    ts = np.cumsum(rng.integers(1, 4, n)).astype(np.int64)
    v = rng.standard_normal(n) * 1e6
    bm = np.packbits(np.ones(n, bool), bitorder="little")
    ocols = [orc.Column(ts, None, orc.INT64), orc.Column(v, bm, orc.FLOAT64)]
    aggs = [("WindowStart", 0), ("IntegralStep", 1), ("WeightedAverageStep", 1), ("IntegralTrapezoid", 1), ("WeightedAverageLinear", 1)]
    ref, _ = orc.aggregate(ocols, 0, interval, aggs)
    tol = order_free_bounds(ocols, 0, interval, aggs, ref=ref)
    W = ref[0].length
    s0 = int(ref[0].values[0])
    tsf = ts.astype(np.float64)
    step, trap = np.zeros(W), np.zeros(W)
    for k in range(W):
        a, b = np.searchsorted(ts, s0 + k * interval), np.searchsorted(ts, s0 + (k + 1) * interval)
        t, x = tsf[a:b], v[a:b]
        terms = np.concatenate([x[:-1] * (t[1:] - t[:-1]), [x[-1] * (float(s0 + (k + 1) * interval) - t[-1])]])
        step[k] = np.sum(terms)                      # numpy's pairwise tree instead of the reference's loop
        b2 = b + 1 if b < n and ts[b] == s0 + (k + 1) * interval else b   # the inclusive row (rolling.go:201-209)
        t, x = tsf[a:b2], v[a:b2]
        trap[k] = np.sum((x[:-1] + x[1:]) / 2 * (t[1:] - t[:-1]))
    assert_within("IntegralStep", step, ref[1].values[:W], tol[1])
    assert_within("WeightedAverageStep", step / float(interval), ref[2].values[:W], tol[2])
    assert_within("IntegralTrapezoid", trap, ref[3].values[:W], tol[3])
    assert_within("WeightedAverageLinear", trap / float(interval), ref[4].values[:W], tol[4])
