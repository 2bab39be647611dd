"""BASELINE.json configs[0]: the reference's benchmark input benchmarks/bow1-100000-rows.parquet, committed as raw
columns in tests/golden/bow1_100000_rows.npz (make_parquet_fixture.py).

CPU part (-m "not gpu"): the oracle on that data against (1) the facts SURVEY §8c states about it and (2) an
INDEPENDENT numpy restatement of the reducers (sort-free group-by on window ids) - a second opinion on the oracle
that shares no code with it.  GPU part (-m gpu): the HIP path, through the C ABI, against the oracle on the same
columns, for the calls the reference's own benchmarks make on this file (rolling Mean - configs[0];
FillLinear(0, 3), IsColSorted(0) / (1) - bowfill_test.go:550-585, bowassertion_test.go:93-111).
All expectations are restatement-derived, not reference-executed (no Go toolchain in this image)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def bow1():
    z = np.load(os.path.join(HERE, "golden", "bow1_100000_rows.npz"))
    return {k: z[k] for k in z.files}


def _valid(b, name):
    return np.unpackbits(b[name + "_valid"], bitorder="little")[:len(b[name])].astype(bool)


def numpy_reducers(ts, vals, valid, interval, offset):
    """Independent restatement (Appendix A.2/A.3/A.5/A.9) for non-negative, ascending ts: python ints for the plan,
    numpy group-by for the reducers.  Returns dict name -> (values, valid_mask)."""
    off = offset % interval if abs(offset) >= interval else offset  # Go's % keeps the sign of the dividend ...
    if off < 0:
        off += interval                                             # ... and rolling.go:124-126 adds the interval back
    t0 = int(ts[0])
    s0 = (t0 // interval) * interval + off                          # t0 >= 0 here: floor == Go's truncation
    if s0 > t0:
        s0 -= interval
    W = (int(ts[-1]) - s0) // interval + 1
    wid = (ts - s0) // interval
    x = vals.astype(np.float64)
    out = {}
    out["WindowStart"] = (s0 + interval * np.arange(W, dtype=np.int64), np.ones(W, bool))
    cnt = np.bincount(wid[valid], minlength=W).astype(np.int64)
    nrows = np.bincount(wid, minlength=W)
    out["Count"] = (cnt, np.ones(W, bool))
    # values are half-integers / small integers: every partial sum is exact in float64, so summation order is moot
    sm = np.bincount(wid[valid], weights=x[valid], minlength=W)
    out["Sum"] = (sm, np.ones(W, bool))
    with np.errstate(invalid="ignore", divide="ignore"):
        out["ArithmeticMean"] = (np.where(cnt > 0, sm / np.maximum(cnt, 1), 0.0), cnt > 0)
    mn = np.full(W, np.inf)
    mx = np.full(W, -np.inf)
    np.minimum.at(mn, wid[valid], x[valid])
    np.maximum.at(mx, wid[valid], x[valid])
    out["Min"] = (np.where(cnt > 0, mn, 0.0), cnt > 0)
    out["Max"] = (np.where(cnt > 0, mx, 0.0), cnt > 0)
    return s0, W, nrows, out


def test_facts_about_the_file(bow1):
    ts = bow1["Int64_ref"]
    assert len(ts) == 100_000 and ts[0] == 6 and ts[-1] == 999_995
    d = np.diff(ts)
    assert d.min() == 1 and d.max() == 19                      # strictly increasing
    assert int((~_valid(bow1, "Float64_bow1")).sum()) == 30_137
    tcol = orc.Column(ts, None, orc.INT64)
    assert orc.plan_windows(tcol, 10, 0) == (0, 100_000)         # s_0 = 0, W = 100 000 ...
    wins = orc.iterate_windows(tcol, 10, 0, False)
    assert all(w["slice_end"] - w["slice_begin"] == 1 for w in wins)  # ... exactly one row per window, none empty
    assert orc.plan_windows(tcol, 100, 7) == (-93, 10_001)       # interval 100 offset 7 => s_0 = -93, W = 10 001


@pytest.mark.parametrize("interval,offset", [(10, 0), (100, 0), (100, 7), (1000, 0), (1000, -250), (37, 5)])
@pytest.mark.parametrize("col", ["Float64_bow1", "Int64_bow1", "Int64_no_nils_bow1"])
def test_oracle_vs_independent_numpy(bow1, interval, offset, col):
    ts, vals, valid = bow1["Int64_ref"], bow1[col], _valid(bow1, col)
    s0, W, nrows, want = numpy_reducers(ts, vals, valid, interval, offset)
    typ = orc.FLOAT64 if vals.dtype == np.float64 else orc.INT64
    cols = [orc.Column(ts, None, orc.INT64), orc.Column(vals, bow1[col + "_valid"], typ)]
    assert orc.plan_windows(cols[0], interval, offset) == (s0, W)
    kinds = ["WindowStart", "Sum", "ArithmeticMean", "Min", "Max", "Count", "NumRows"]
    got, _ = orc.aggregate(cols, 0, interval, [(k, 0 if k == "WindowStart" else 1) for k in kinds], offset=offset)
    for k, g in zip(kinds, got):
        gv, gm = g.values[:g.length], g.valid_mask()
        if k == "NumRows":
            assert gm.all() and np.array_equal(gv, nrows.astype(np.float64))
            continue
        wv, wm = want[k]
        assert np.array_equal(gm, wm), k
        assert np.array_equal(gv[gm], wv[wm].astype(gv.dtype)), k


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_hip_rolling_on_the_file(bow1):
    from bow_amd import capi
    from test_gpu_aggregate import compare
    ts = bow1["Int64_ref"]
    kinds = ["WindowStart", "Sum", "ArithmeticMean", "Min", "Max", "Count", "First", "Last", "NumRows"]
    for col in ("Float64_bow1", "Int64_bow1", "Int64_no_nils_bow1"):
        vals, bm = bow1[col], bow1[col + "_valid"]
        typ = capi.FLOAT64 if vals.dtype == np.float64 else capi.INT64
        for interval, offset, inclusive in [(10, 0, False), (100, 7, True), (1000, 0, False), (100_000, 0, False)]:
            aggs = [(k, 0 if k == "WindowStart" else 1) for k in kinds]
            if inclusive:
                aggs += [("IntegralTrapezoid", 1), ("WeightedAverageStep", 1)]
            want, _ = orc.aggregate([orc.Column(ts, None, orc.INT64), orc.Column(vals, bm, typ)], 0, interval, aggs,
                                    offset=offset, inclusive=inclusive)
            got, info = capi.rolling_aggregate([capi.Column(ts), capi.Column(vals, bm, typ, 0, len(vals), -1)], 0, interval, aggs,
                                               offset=offset, inclusive=inclusive)
            from tolerance import order_free_bounds
            tol = order_free_bounds([orc.Column(ts, None, orc.INT64), orc.Column(vals, bm, typ)], 0, interval, aggs, offset=offset,
                                    inclusive=inclusive, ref=want) if info.long_windows else [None] * len(aggs)
            for i, ((k, _), g, w) in enumerate(zip(aggs, got, want)):
                # half-integer data: sums are exact whatever the association, so even the long-window path is bit-exact here
                exact = k not in ("IntegralTrapezoid", "WeightedAverageStep") or info.long_windows == 0
                compare("%s %s I=%d" % (col, k, interval), g, w, exact=exact, bound=None if exact else tol[i])


@pytest.mark.gpu
def test_hip_fill_and_sorted_on_the_file(bow1):
    from bow_amd import capi
    from test_gpu_callers import cmp_out
    ts = bow1["Int64_ref"]
    n = len(ts)
    vals, bm = bow1["Float64_bow1"], bow1["Float64_bow1_valid"]
    ccols = [capi.Column(ts), capi.Column(vals, bm, capi.FLOAT64, 0, n, -1)]
    ocols = [orc.Column(ts, None, orc.INT64), orc.Column(vals, bm, orc.FLOAT64)]
    got, unchanged = capi.fill_linear(ccols, 0, 1)            # data.FillLinear(0, 3)
    want, wu = orc.fill_linear(ocols, 0, 1)
    assert unchanged == wu is False
    cmp_out("FillLinear", got, want)
    for method in ("Previous", "Next", "Mean"):               # data.FillPrevious(3) / FillNext(3) / FillMean(3)
        got, _ = capi.fill(ccols[1], method)
        want, _ = orc.fill(ocols[1], method)
        cmp_out("Fill" + method, got, want)
    assert capi.is_col_sorted(ccols[0]) is True               # data.IsColSorted(0)
    unsorted = capi.Column(bow1["Int64_no_nils_bow1"])
    assert capi.is_col_sorted(unsorted) is False              # data.IsColSorted(1)
    # configs[2] in miniature: Linear interpolation at the window starts, then the rolling mean
    ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
    g = capi.rolling_interpolate(ccols, 0, 100, ip, offset=7)
    w = orc.interpolate(ocols, 0, 100, ip, offset=7)
    cmp_out("interp ts", g[0], w[0])
    cmp_out("interp val", g[1], w[1])
