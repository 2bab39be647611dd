"""GPU parity tests for Rolling.Aggregate: the HIP path (through the C ABI) against
(1) the reference's own golden vectors and (2) the CPU oracle on seeded inputs.
Bar: bit-exact for every output (values of valid slots, validity bitmaps, null slots == 0);
Sum/Mean/Integral of windows that take an order-free path: within the stated bound c n 2^-53 sum|x_i| (tests/tolerance.py)."""
import os

import numpy as np
import pytest

from bow_amd import capi
from oracle import pyoracle as orc
from tolerance import assert_within, order_free_bounds

pytestmark = pytest.mark.gpu

T = {"float64": capi.FLOAT64, "int64": capi.INT64}
ALL_AGGS = ["WindowStart", "Sum", "ArithmeticMean", "Min", "Max", "Count", "First", "Last", "NumRows"]
TIME_AGGS = ["IntegralStep", "IntegralTrapezoid", "WeightedAverageStep", "WeightedAverageLinear"]
ORDER_SENSITIVE = {"Sum", "ArithmeticMean", "IntegralStep", "IntegralTrapezoid", "WeightedAverageStep",
                   "WeightedAverageLinear"}


def same_list(a, b):
    assert len(a) == len(b), (a, b)
    for x, y in zip(a, b):
        assert (x is None) == (y is None) and (x is None or x == y), (a, b)


def compare(name, got, want, exact=True, bound=None):
    """got: capi.OutColumn, want: orc.Column.  exact=False: the values of an order-free reducer, checked against `bound`
    (absolute, one per output slot: tolerance.order_free_bounds) - everything else about the column stays bit-exact."""
    assert got.length == want.length, (name, got.length, want.length)
    assert got.type == want.type, (name, got.type, want.type)
    gv, gb = got.host_arrays()
    gm, wm = got.valid_mask(), want.valid_mask()
    assert np.array_equal(gm, wm), (name, np.flatnonzero(gm != wm)[:10])
    assert got.null_count == int((~wm).sum()), name
    wv = want.values[:want.length]
    gbits, wbits = gv.view(np.uint64), wv.view(np.uint64)
    # null slots hold 0 (bow.NewBuffer zero-init; never written)
    assert not gbits[~gm].any(), name
    if exact:
        diff = gbits[gm] != wbits[wm]
        if diff.any() and got.type == capi.FLOAT64:
            # A NaN that the arithmetic GENERATES (inf - inf, 0 * inf) carries the hardware's default payload: x86 (the oracle,
            # and Go on amd64) gives the negative "real indefinite" 0xFFF8..., gfx950 the positive 0x7FF8...; Go prints both as
            # NaN and no operation on this path tells them apart.  NaNs that come from the input keep their bits on both.
            diff &= ~(np.isnan(gv[gm].view(np.float64)) & np.isnan(wv[wm].view(np.float64)))
        bad = np.flatnonzero(diff)
        assert bad.size == 0, (name, bad[:10], gv[gm][bad[:5]], wv[wm][bad[:5]])
    else:
        assert bound is not None, "an order-free comparison needs its bound (tolerance.order_free_bounds)"
        assert_within(name, gv[gm].astype(np.float64), wv[wm].astype(np.float64), np.asarray(bound)[gm])
    # padding bits of the last validity byte stay clear
    n = got.length
    if n % 8:
        assert (gb[-1] >> (n % 8)) == 0, name


def _names(aggs):
    return [a[0] for a in aggs]


def run_both(ts, cols_np, interval, aggs, offset=0, inclusive=False, device=False):
    """cols_np: list of (values ndarray, valid bool ndarray or None).  Column 0 is ts.
    Runs the HIP path twice - lean kernel (where it applies) and general kernel
    (capi.ROUTE_FORCE_GENERAL) - checks both against the oracle, returns the first run."""
    ccols = [capi.Column(ts, None, capi.INT64)]
    ocols = [orc.Column(ts, None, orc.INT64)]
    for vals, valid in cols_np:
        bm = None if valid is None else np.packbits(valid, bitorder="little")
        typ = capi.INT64 if vals.dtype == np.int64 else capi.FLOAT64
        ccols.append(capi.Column(vals, bm, typ, 0, len(vals), -1))
        ocols.append(orc.Column(vals, bm, typ))
    if device:
        ccols = [c.to_device() for c in ccols]
    exp, nic = orc.aggregate(ocols, 0, interval, aggs, offset=offset, inclusive=inclusive)
    first, bounds = None, None
    # three HIP code paths over the same inputs: simple kernel where it applies (else lean), lean kernel, general kernel
    # (the lean / general runs also switch the long-only shortcut off, so windows of thousands of rows take both long paths)
    for label in capi.agg_routes():
        outs, info = capi.rolling_aggregate(ccols, 0, interval, aggs, offset=offset, inclusive=inclusive,
                                            out_residency=capi.DEVICE if device else capi.HOST)
        assert info.new_interval_col == nic
        if info.long_windows and bounds is None:
            bounds = order_free_bounds(ocols, 0, interval, aggs, offset=offset, inclusive=inclusive, ref=exp)
        for i, (k, g, w) in enumerate(zip(_names(aggs), outs, exp)):
            exact = info.long_windows == 0 or k not in ORDER_SENSITIVE
            compare("%s path=%s n=%d I=%d off=%d" % (k, label, len(ts), interval, offset), g, w, exact=exact,
                    bound=None if exact else bounds[i])
        if first is None:
            first = (outs, exp, info)
    if bounds is None:
        bounds = [None] * len(aggs)
    first[2].bounds = bounds          # (a plain Python attribute on the ctypes record: what re-comparisons of the first run need)
    return first


# ------------------------------------------------------------------ golden vectors through the HIP path
def test_golden_reducers(golden):
    n = 0
    for v in golden["reducers"]:
        b = golden["bows"][v["bow"]]
        if b["value_type"] != "float64":
            continue  # Boolean columns are outside the device path (north_star: int64/float64)
        cols = [capi.Column.from_list(b["time"], "int64"), capi.Column.from_list(b["value"], "float64")]
        outs, info = capi.rolling_aggregate(cols, 0, v["interval"], [("WindowStart", 0), (v["reducer"], 1, v["factors"])],
                                            offset=v["offset"])
        assert outs[0].type == capi.INT64
        same_list(outs[0].to_list(), v["expect_time"])
        assert outs[1].type == T[v["expect_type"]], (v["reducer"], v["name"])
        same_list(outs[1].to_list(), v["expect_value"])
        n += 1
    assert n >= 20


def test_golden_driver(golden):
    kind = {"FirstValue": ("WindowStart", None), "NumRows": ("NumRows", None), "NumRows2x": ("NumRows", [2.0])}
    names = {"time": 0, "value": 1}
    for v in golden["driver"]:
        cols = [capi.Column.from_list(v["time"], "int64"), capi.Column.from_list(v["value"], "float64")]
        if "error" in v:
            aggs = [(kind.get(k, ("WindowStart", None))[0], names.get(n, 7)) for n, k, _ in v["aggs"]]
            with pytest.raises(capi.BowGpuError) as e:
                capi.rolling_aggregate(cols, 0, v["interval"], aggs)
            assert e.value.code == (-5 if "must keep" in v["error"] else -6)
            continue
        aggs = [(kind[k][0], names[n], kind[k][1]) for n, k, _ in v["aggs"]]
        outs, info = capi.rolling_aggregate(cols, 0, v["interval"], aggs)
        for o, typ, exp in zip(outs, v["expect_types"], v["expect"]):
            assert o.type == T[typ]
            same_list(o.to_list(), exp)
        assert info.new_interval_col == max(i for i, a in enumerate(v["aggs"]) if a[0] == "time")


def test_golden_iterate_rows_via_numrows(golden):
    # window membership of rolling_test.go:111-297 observed through NumRows/First/Last/Sum
    for v in golden["iterate"]:
        if v["inclusive"]:
            continue
        cols = [capi.Column.from_list(v["time"], "int64"), capi.Column.from_list(v["value"], "float64")]
        outs, _ = capi.rolling_aggregate(cols, 0, v["interval"],
                                         [("WindowStart", 0), ("NumRows", 1), ("First", 1), ("Last", 1)], offset=v["offset"])
        same_list(outs[0].to_list(), [w["start"] for w in v["windows"]])
        same_list(outs[1].to_list(), [float(len(w["time_rows"])) for w in v["windows"]])
        same_list(outs[2].to_list(), [w["value_rows"][0] if w["value_rows"] else None for w in v["windows"]])
        same_list(outs[3].to_list(), [w["value_rows"][-1] if w["value_rows"] else None for w in v["windows"]])


def test_empty_bow():
    cols = [capi.Column.from_list([], "int64"), capi.Column.from_list([], "float64")]
    outs, info = capi.rolling_aggregate(cols, 0, 10, [("WindowStart", 0), ("ArithmeticMean", 1)])
    assert info.num_windows == 0 and outs[0].length == 0 and outs[1].length == 0
    assert outs[0].type == capi.INT64 and outs[1].type == capi.FLOAT64


# ------------------------------------------------------------------ randomized parity vs the oracle
def make_ts(rng, n, mode):
    if mode == "dense":
        return np.arange(n, dtype=np.int64) + int(rng.integers(-50, 50))
    if mode == "irregular":  # steps in [1,19] like bowgenerator.go:97-106
        return np.cumsum(rng.integers(1, 20, n)).astype(np.int64) + int(rng.integers(-1000, 1000))
    if mode == "dups":  # duplicates and runs
        return np.cumsum(rng.integers(0, 3, n)).astype(np.int64) - 77
    if mode == "gappy":  # occasional big jumps => runs of empty windows
        step = rng.integers(1, 5, n)
        step[rng.random(n) < 0.01] = rng.integers(100, 5000)
        return np.cumsum(step).astype(np.int64) - 12345
    if mode == "negative":
        return np.cumsum(rng.integers(1, 7, n)).astype(np.int64) - 3 * n
    raise ValueError(mode)


def make_vals(rng, n, kind, null_frac):
    if kind == "f64":
        v = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 6, n)
    elif kind == "i64":
        v = rng.integers(-(2 ** 40), 2 ** 40, n).astype(np.int64)
    else:
        raise ValueError(kind)
    valid = None if null_frac == 0 else rng.random(n) >= null_frac
    return v, valid


@pytest.mark.parametrize("mode", ["dense", "irregular", "dups", "gappy", "negative"])
@pytest.mark.parametrize("vkind,null_frac", [("f64", 0.0), ("f64", 0.3), ("i64", 0.3)])
def test_random_parity_exact(mode, vkind, null_frac):
    rng = np.random.default_rng(hash((mode, vkind, null_frac)) % (2 ** 32))
    for n, interval, offset in [(1, 10, 0), (7, 3, 1), (2047, 10, 0), (2048, 10, 3), (2049, 10, -4), (6000, 7, 8),
                                (50_000, 10, 0), (50_000, 13, 5), (200_003, 25, 11)]:
        ts = make_ts(rng, n, mode)
        vals, valid = make_vals(rng, n, vkind, null_frac)
        aggs = [(k, 0 if k == "WindowStart" else 1) for k in ALL_AGGS]
        outs, exp, info = run_both(ts, [(vals, valid)], interval, aggs, offset=offset)
        for (k, _), g, w in zip(aggs, outs, exp):
            exact = info.long_windows == 0 or k not in ORDER_SENSITIVE
            compare("%s/%s n=%d I=%d" % (mode, k, n, interval), g, w, exact=exact, bound=info.bounds[aggs.index((k, _))])


@pytest.mark.parametrize("inclusive", [False, True])
def test_time_weighted_reducers(inclusive):
    rng = np.random.default_rng(99)
    for mode in ["irregular", "dups", "gappy", "dense"]:
        for n, interval, offset in [(5, 10, 0), (3000, 10, 0), (40_000, 16, 5)]:
            ts = make_ts(rng, n, mode)
            vals, valid = make_vals(rng, n, "f64", 0.3)
            aggs = [("WindowStart", 0)] + [(k, 1) for k in TIME_AGGS] + [("Sum", 1), ("Count", 1)]
            outs, exp, info = run_both(ts, [(vals, valid)], interval, aggs, offset=offset, inclusive=inclusive)
            assert info.inclusive == 1  # trapezoid / linear force inclusive windows (aggregation.go:183-185)
            for (k, _), g, w in zip(aggs, outs, exp):
                exact = info.long_windows == 0 or k not in ORDER_SENSITIVE
                compare("%s/%s n=%d" % (mode, k, n), g, w, exact=exact, bound=info.bounds[aggs.index((k, _))])


def test_inclusive_option_does_not_change_plain_reducers():
    # A.6: reducers that don't need the inclusive point get UnsetInclusive windows
    rng = np.random.default_rng(5)
    ts = make_ts(rng, 30_000, "dups") * 2
    vals, valid = make_vals(rng, 30_000, "f64", 0.2)
    aggs = [("WindowStart", 0), ("Sum", 1), ("Count", 1), ("Last", 1), ("NumRows", 1)]
    outs, exp, _ = run_both(ts, [(vals, valid)], 4, aggs, inclusive=True)
    for (k, _), g, w in zip(aggs, outs, exp):
        compare(k, g, w)


def test_multi_column_and_factors():
    rng = np.random.default_rng(11)
    n = 120_000
    ts = make_ts(rng, n, "irregular")
    cols = [make_vals(rng, n, "f64", 0.0), make_vals(rng, n, "f64", 0.3), make_vals(rng, n, "i64", 0.1),
            make_vals(rng, n, "i64", 0.0)]
    aggs = [("ArithmeticMean", 1), ("WindowStart", 0, [0.5]), ("Sum", 2, [-1.0]), ("Max", 3), ("First", 3, [0.1]),
            ("Count", 4, [3.0]), ("Min", 4), ("Last", 2), ("WindowStart", 0), ("ArithmeticMean", 0), ("Sum", 1, [0.1, 10.0])]
    outs, exp, info = run_both(ts, cols, 100, aggs, offset=7)
    for (k, *_), g, w in zip(aggs, outs, exp):
        compare(k, g, w)
    assert info.new_interval_col == 9


@pytest.mark.parametrize("vkind", ["f64", "i64"])
def test_factors_on_the_simple_and_time_weighted_kernels(vkind):
    # transformation.Factor chains (factor.go:7-20) on every reducer, one column type per call so the simple / time-weighted
    # wave kernels take it; gappy ts => empty windows, where a negative factor turns Sum / NumRows (+0.0) into -0.0
    rng = np.random.default_rng(21)
    n = 90_000
    ts = make_ts(rng, n, "gappy")
    cols = [make_vals(rng, n, vkind, 0.25), make_vals(rng, n, vkind, 0.0)]
    aggs = [("WindowStart", 0, [0.5]), ("Sum", 1, [-1.0]), ("ArithmeticMean", 1, [0.1, 10.0]), ("Min", 2, [3.0]), ("Max", 1, [-2.5]),
            ("Count", 1, [3.0]), ("First", 2, [0.1]), ("Last", 1, [7.0]), ("NumRows", 1, [-2.0]), ("WindowStart", 0)]
    outs, exp, info = run_both(ts, cols, 10, aggs, offset=3)
    assert capi.last_kernel_name() == "rolling_agg_kernel"  # (run_both ends with the general kernel; "auto" was the simple one)
    tw = [("WindowStart", 0), ("IntegralStep", 1, [0.5]), ("IntegralTrapezoid", 1, [-1.0]), ("WeightedAverageStep", 2, [2.0]),
          ("WeightedAverageLinear", 1, [0.25, 4.0]), ("Sum", 1, [-1.0]), ("Count", 2, [2.0])]
    for inclusive in (False, True):
        run_both(ts, cols, 10, tw, offset=3, inclusive=inclusive)
    assert capi.get_route() == 0
    capi.rolling_aggregate([capi.Column(ts)] + [capi.Column(v, np.packbits(m, bitorder="little") if m is not None else None,
                                                             capi.INT64 if vkind == "i64" else capi.FLOAT64, 0, n, -1) for v, m in cols],
                           0, 10, tw, offset=3)
    assert capi.last_kernel_name() == "rolling_tw_kernel"
    capi.rolling_aggregate([capi.Column(ts)] + [capi.Column(v, np.packbits(m, bitorder="little") if m is not None else None,
                                                             capi.INT64 if vkind == "i64" else capi.FLOAT64, 0, n, -1) for v, m in cols],
                           0, 10, aggs, offset=3)
    assert capi.last_kernel_name() == "rolling_simple_kernel"


def test_sixteen_outputs_in_one_call():
    # the ABI's maximum (16 reducers) over three columns of both types, through every kernel
    rng = np.random.default_rng(2)
    n = 50_000
    ts = make_ts(rng, n, "irregular")
    cols = [make_vals(rng, n, "f64", 0.2), make_vals(rng, n, "i64", 0.0), make_vals(rng, n, "f64", 0.0)]
    aggs = [("WindowStart", 0)] + [(k, 1 + (i % 3)) for i, k in enumerate(["Sum", "ArithmeticMean", "Min", "Max", "Count", "First", "Last", "NumRows",
                                                                           "Sum", "ArithmeticMean", "Min", "Max", "Count", "First", "Last"])]
    assert len(aggs) == 16
    run_both(ts, cols, 25, aggs, offset=4)
    run_both(ts, cols, 60, aggs, offset=4)
    cc = [capi.Column(ts)] + [capi.Column(v, None if m is None else np.packbits(m, bitorder="little"),
                                          capi.INT64 if v.dtype == np.int64 else capi.FLOAT64, 0, n, -1) for v, m in cols]
    capi.rolling_aggregate(cc, 0, 25, aggs, offset=4)
    assert capi.last_kernel_name() == "rolling_simple_kernel"
    capi.rolling_aggregate(cc, 0, 12, aggs, offset=4)     # 1.2 rows per window: more heads than even the large list holds => redone by the wave kernel
    assert capi.last_kernel_name() == "rolling_wave_kernel"


def test_more_outputs_and_columns_than_one_launch_takes():
    """the reference loops over any number of aggregators (aggregation.go:190-238): calls beyond one launch's 16 outputs / 8
    column passes are cut into batches behind the ABI - 40 reducers over 12 columns, Mode and inclusive reducers among them"""
    rng = np.random.default_rng(8)
    n = 20_000
    ts = make_ts(rng, n, "irregular")
    cols = [make_vals(rng, n, "f64" if j % 3 else "i64", [0.0, 0.2, 0.6][j % 3]) for j in range(12)]
    kinds = ["Sum", "ArithmeticMean", "Min", "Max", "Count", "First", "Last", "NumRows", "IntegralStep", "WeightedAverageLinear", "Mode"]
    aggs = [("WindowStart", 0)]
    for i in range(39):
        k = kinds[int(rng.integers(0, len(kinds)))]
        a = (k, 1 + int(rng.integers(0, 12)))
        aggs.append(a + ([2.0],) if rng.random() < 0.2 and k != "Mode" else a)
    aggs.append(("Sum", 0))           # the interval column as a value column, last in the list (new interval column = this one)
    outs, exp, info = run_both(ts, cols, 25, aggs, offset=4)
    assert len(outs) == 41
    # all twelve columns with five nullable reducers each: more column passes than a launch has
    aggs = [("WindowStart", 0)] + [(k, 1 + j) for j in range(12) for k in ("ArithmeticMean", "Min", "Max", "First", "Last")]
    run_both(ts, cols, 7, aggs)


def test_nan_inf_signed_zero_semantics():
    # minmax.go: NaN result iff the FIRST valid value is NaN; -0.0/+0.0 ties keep the earlier one
    ts = np.array([0, 1, 2, 10, 11, 12, 20, 21, 30, 31, 40, 41, 42], dtype=np.int64)
    v = np.array([np.nan, 1.0, -1.0, 2.0, np.nan, 3.0, -0.0, 0.0, 0.0, -0.0, np.inf, -np.inf, 5.0])
    valid = np.ones(len(v), bool)
    aggs = [("WindowStart", 0), ("Min", 1), ("Max", 1), ("Sum", 1), ("ArithmeticMean", 1), ("First", 1), ("Last", 1)]
    outs, exp, _ = run_both(ts, [(v, valid)], 10, aggs)
    for (k, _), g, w in zip(aggs, outs, exp):
        compare(k, g, w)
    gmin = outs[1].host_arrays()[0]
    assert np.isnan(gmin[0]) and gmin[1] == 2.0
    assert np.signbit(gmin[2]) and not np.signbit(gmin[3])
    # all(-0.0) window sums to +0.0 because the accumulator starts at +0.0 (sum.go:15)
    outs, exp, _ = run_both(np.array([0, 1], dtype=np.int64), [(np.array([-0.0, -0.0]), None)], 10, [("WindowStart", 0), ("Sum", 1)])
    compare("Sum", outs[1], exp[1])
    assert not np.signbit(outs[1].host_arrays()[0][0])


@pytest.mark.parametrize("rows_per_window", [10, 40, 100])
def test_signalling_nan_inside_a_window_does_not_lose_the_running_extremum(rows_per_window):
    """minmax.go:22-27 compares: a NaN of either kind in mid-window is skipped.  v_min_f64 / v_max_f64 (IEEE mode) return a
    SIGNALLING operand quieted and the step after it drops that quiet NaN for the next row - [5, sNaN, 7] would give Min 7.  A
    tile that stages one walks its extrema by comparison (agg_device.h is_snan): the simple kernel (extrema alone, next to sums),
    the time-weighted kernel (extrema next to an integral), a nullable column in both of its forms (two walks in tiles of few
    windows, one predicated walk in tiles of many) and an Int64 neighbour column that cannot hold one."""
    rng = np.random.default_rng(9 + rows_per_window)
    n = 40_000
    ts = np.arange(n, dtype=np.int64)
    v = rng.normal(size=n)
    SNAN = np.array([0x7FF0000000000001, 0xFFF4000000000123], dtype=np.uint64).view(np.float64)
    # never a window's first or last valid row: rows 3 and 6 of every third window; the row before holds the window's extremum
    w0 = np.arange(0, n - rows_per_window, 3 * rows_per_window)
    v[w0 + 2] = 1e6
    v[w0 + 5] = -1e6
    v[w0 + 3] = SNAN[0]
    v[w0 + 6] = SNAN[1]
    valid = rng.random(n) > 0.3
    valid[w0[:, None] + np.arange(8)[None, :]] = True
    vi = rng.integers(-1000, 1000, n).astype(np.int64)
    for cols, aggs in [
        ([(v, None)], [("WindowStart", 0), ("Min", 1), ("Max", 1)]),
        ([(v, None)], [("WindowStart", 0), ("Min", 1), ("Max", 1), ("Sum", 1), ("First", 1), ("Last", 1)]),
        ([(v, valid)], [("WindowStart", 0), ("Min", 1), ("Max", 1)]),
        ([(v, valid)], [("WindowStart", 0), ("Min", 1), ("Max", 1), ("ArithmeticMean", 1)]),
        ([(v, None), (vi, None)], [("WindowStart", 0), ("Max", 1), ("Min", 2), ("Min", 1)]),
        ([(v, None)], [("WindowStart", 0), ("Min", 1), ("Max", 1), ("IntegralStep", 1)]),
        ([(v, valid)], [("WindowStart", 0), ("Min", 1), ("Max", 1), ("WeightedAverageLinear", 1), ("Sum", 1)]),
    ]:
        outs, exp, info = run_both(ts, cols, rows_per_window, aggs)
        assert info.long_windows == 0
        k = _names(aggs)
        gmin, gmax = outs[k.index("Min")], outs[k.index("Max")]
        if cols[0][1] is None and len(cols) == 1:   # (the planted extremes are what comes out: the sNaN behind them changed nothing)
            slot = w0 // rows_per_window
            assert (gmax.host_arrays()[0][slot] == 1e6).all() and (gmin.host_arrays()[0][slot] == -1e6).all()


@pytest.mark.parametrize("rows_per_window", [300, 1000, 6000])
def test_signalling_nan_inside_a_long_window(rows_per_window):
    """... and the same through the long-window forms (streaming, bisection, the tile kernels' queue): Min / Max are exact reducers
    there too - the extremum in front of an sNaN and the one behind it, with and without nulls, an sNaN as a window's very first value
    (minmax.go keeps a NaN seed: the result is that NaN, bit for bit)."""
    rng = np.random.default_rng(19 + rows_per_window)
    n = 60 * rows_per_window + 77
    ts = np.arange(n, dtype=np.int64)
    v = rng.normal(size=n)
    SNAN = np.array([0x7FF0000000000001, 0xFFF4000000000123], dtype=np.uint64).view(np.float64)
    w0 = np.arange(0, n - rows_per_window, 3 * rows_per_window)
    for d in (2, rows_per_window // 2, rows_per_window - 9):      # near the window's start, in its middle (another lane / trip), near its end
        v[w0 + d] = 1e6
        v[w0 + d + 3] = -1e6
        v[w0 + d + 1] = SNAN[0]
        v[w0 + d + 4] = SNAN[1]
    v[w0[::4] + rows_per_window] = SNAN[0]                      # a seed: the first row of the window behind every fourth planted one
    valid = rng.random(n) > 0.3
    valid[w0[:, None] + np.arange(8)[None, :]] = True
    for cols, aggs in [
        ([(v, None)], [("WindowStart", 0), ("Min", 1), ("Max", 1)]),
        ([(v, None)], [("WindowStart", 0), ("Min", 1), ("Max", 1), ("Sum", 1), ("First", 1), ("Last", 1)]),
        ([(v, valid)], [("WindowStart", 0), ("Min", 1), ("Max", 1)]),
        ([(v, valid)], [("WindowStart", 0), ("Min", 1), ("Max", 1), ("WeightedAverageStep", 1)]),
    ]:
        outs, exp, info = run_both(ts, cols, rows_per_window, aggs)
        k = _names(aggs)
        if cols[0][1] is None:
            slot = w0 // rows_per_window
            assert (outs[k.index("Max")].host_arrays()[0][slot] == 1e6).all() and (outs[k.index("Min")].host_arrays()[0][slot] == -1e6).all()


def test_device_resident_columns_and_arrow_offsets():
    rng = np.random.default_rng(21)
    n, off = 70_000, 13  # a sliced Arrow array: offset 13 into shared buffers (bow.go:279-283)
    ts_full = make_ts(rng, n + off, "irregular")
    v_full, valid_full = make_vals(rng, n + off, "f64", 0.3)
    bm = np.packbits(valid_full, bitorder="little")
    aggs = [("WindowStart", 0), ("ArithmeticMean", 1), ("Count", 1), ("Min", 1)]
    ocols = [orc.Column(ts_full, None, orc.INT64, offset=off, length=n), orc.Column(v_full, bm, orc.FLOAT64, offset=off, length=n)]
    exp, _ = orc.aggregate(ocols, 0, 50, aggs)
    for dev in (False, True):
        ccols = [capi.Column(ts_full, None, capi.INT64, off, n, 0), capi.Column(v_full, bm, capi.FLOAT64, off, n, -1)]
        if dev:
            ccols = [c.to_device() for c in ccols]
        outs, _ = capi.rolling_aggregate(ccols, 0, 50, aggs, out_residency=capi.DEVICE if dev else capi.HOST)
        for (k, _), g, w in zip(aggs, outs, exp):
            compare("%s dev=%s" % (k, dev), g, w)


def test_order_free_windows_on_cancelling_data_stay_within_the_stated_bound():
    """long windows (order-free forms) over terms of size 1e16 that cancel to O(100): the relative error of any reordered sum is
    unbounded there, the STATED bound c n 2^-53 sum|x_i| (tests/tolerance.py, bowgpu.h) is what the library promises - asserted
    through every route, long-only forms and the tile kernels' queue, together with the exact reducers bit for bit"""
    rng = np.random.default_rng(77)
    n = 200_000
    ts = np.arange(n, dtype=np.int64)
    v = np.empty(n)
    for a in range(0, n, 20_000):
        big = rng.standard_normal(10_000) * 1e16
        v[a:a + 10_000] = big
        v[a + 10_000:a + 20_000] = -rng.permutation(big) + rng.standard_normal(10_000)
    aggs = [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Sum", 1, [-0.25]), ("Min", 1), ("Count", 1),
            ("IntegralStep", 1), ("WeightedAverageLinear", 1)]
    for interval in (20_000, 40_000, 1_000):
        outs, exp, info = run_both(ts, [(v, None)], interval, aggs)
        assert info.long_windows > 0
        if interval == 20_000:   # the data is what the docstring says: the reference's own sums are tiny against sum|x|
            assert np.abs(exp[1].values[:exp[1].length]).max() < 1e-9 * np.abs(v).sum()
    # mostly short windows with a few long ones in between (the tile kernels' queue): ts jumps so that every 20 000 rows form one window
    ts2 = np.arange(n, dtype=np.int64) * 50
    ts2[100_000:120_000] = ts2[100_000] + np.arange(20_000) // 1000
    run_both(ts2, [(v, None)], 500, aggs)


def test_long_windows_cooperative_path():
    rng = np.random.default_rng(31)
    n = 300_000
    ts = make_ts(rng, n, "dense")
    vals, valid = make_vals(rng, n, "f64", 0.2)
    aggs = [(k, 0 if k == "WindowStart" else 1) for k in ALL_AGGS] + [(k, 1) for k in TIME_AGGS]
    for interval in (1000, 50_000, 10 ** 9):
        outs, exp, info = run_both(ts, [(vals, valid)], interval, aggs)
        assert info.long_windows > 0
        for (k, _), g, w in zip(aggs, outs, exp):
            compare("%s I=%d" % (k, interval), g, w, exact=k not in ORDER_SENSITIVE, bound=info.bounds[aggs.index((k, _))])


@pytest.mark.parametrize("inclusive", [False, True])
def test_long_windows_mixed_shapes(inclusive):
    """Windows of 1 .. 60k rows side by side (several 8192-row chunks, chunk boundaries inside null runs), runs of empty
    windows after long ones, an all-null long window, NaN seeds and -0.0/+0.0 ties for Min/Max, an Int64 column, two
    value columns - through the multi-workgroup long-window path of every tile kernel."""
    rng = np.random.default_rng(77 + inclusive)
    interval = 1000
    pieces, t = [], -5000
    for rows in (3, 700, 1, 9000, 60_000, 12, 2500, 8192, 8193, 40, 20_000, 641, 5):
        # `rows` rows spread inside one window, then skip 0..3 windows
        offs = np.sort(rng.integers(0, interval, rows))
        if inclusive and rows > 1:
            offs[0] = 0  # a row exactly on the window start => the previous window is inclusive
        pieces.append(t + offs)
        t += interval * int(rng.integers(1, 5))
    ts = np.concatenate(pieces).astype(np.int64)
    n = len(ts)
    f, fvalid = make_vals(rng, n, "f64", 0.25)
    f[rng.random(n) < 0.001] = np.nan
    f[rng.random(n) < 0.01] = 0.0
    f[rng.random(n) < 0.01] = -0.0
    i, ivalid = make_vals(rng, n, "i64", 0.1)
    # the 20_000-row window: all null in the float column; the 9000-row one: first valid value NaN
    starts = np.cumsum([0] + [len(p) for p in pieces])
    fvalid[starts[10]:starts[11]] = False
    fvalid[starts[3]] = True
    f[starts[3]] = np.nan
    # a null run across a chunk boundary of the 60k window
    fvalid[starts[4] + 8192 - 40:starts[4] + 8192 + 40] = False
    aggs = ([(k, 0 if k == "WindowStart" else 1) for k in ALL_AGGS] + [(k, 1) for k in TIME_AGGS] +
            [(k, 2) for k in ("Sum", "Min", "IntegralTrapezoid")])
    outs, exp, info = run_both(ts, [(f, fvalid), (i, ivalid)], interval, aggs, inclusive=inclusive)
    assert info.long_windows >= 5
    # the simple / lean kernels only take exclusive windows without time-weighted reducers: cover them too
    if not inclusive:
        aggs2 = [(k, 0 if k == "WindowStart" else 1) for k in ALL_AGGS] + [(k, 2) for k in ("Sum", "Min", "Max", "Last")]
        outs, exp, info = run_both(ts, [(f, fvalid), (i, ivalid)], interval, aggs2)
        assert info.long_windows >= 5


def test_streaming_form_chunk_edges():
    """The one-read streaming form (windows averaging >= 129 rows) on shapes built around its 512-row chunks: window boundaries exactly
    on chunk edges and one row off them, windows of exactly one chunk / several chunks / more chunks than a lane of the finish kernel
    looks ahead, a partial last chunk of 1 .. 511 rows, empty windows between long ones, an all-null window, Int64 values,
    time-weighted reducers and inclusive windows, rows below s0 in a long window 0 - against the oracle and against the
    bisection form (`classic-long` in run_both)."""
    rng = np.random.default_rng(512)
    base_aggs = [(k, 0 if k == "WindowStart" else 1) for k in ALL_AGGS]
    tw = [("WindowStart", 0)] + [(k, 1) for k in TIME_AGGS] + [("Sum", 2), ("Max", 2)]
    for n in (512 * 9, 512 * 9 + 1, 512 * 9 - 1, 512 * 40 + 255, 513, 511 * 3):
        for first in (0, 1, 511, -700):
            ts = np.arange(n, dtype=np.int64) + first
            f, fv = make_vals(rng, n, "f64", 0.2)
            i, iv = make_vals(rng, n, "i64", 0.1)
            fv[512:1024] = False                         # one chunk with no valid value at all
            for interval in (512, 256, 1024, 1536, 135, 700):
                if n // interval < 1 or n / max(1, (n // interval + 2)) < 129:      # (kLongStreamAnyAvgRows)
                    continue
                for offset in (0, 1, interval - 1):
                    run_both(ts, [(f, fv), (i, iv)], interval, base_aggs + [("Sum", 2), ("Min", 2)], offset=offset)
                    # ({sum, count} reducer sets take the streaming form on their own)
                    outs, exp, info = run_both(ts, [(f, fv), (i, iv)], interval, [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 2), ("Count", 1)], offset=offset)
                    assert info.long_windows == info.num_windows > 0, (n, first, interval, offset)
                run_both(ts, [(f, fv), (i, iv)], interval, tw, offset=3, inclusive=True)
    # windows over more chunks than a lane of stream_final_kernel walks (64) next to short ones, and runs of empty windows
    pieces, t = [], 0
    for rows in (40_000, 3, 600, 70_000, 1, 513, 33_000):
        pieces.append(t + np.sort(rng.integers(0, 1000, rows)))
        t += 1000 * int(rng.integers(1, 6))
    ts = np.concatenate(pieces).astype(np.int64)
    f, fv = make_vals(rng, len(ts), "f64", 0.3)
    outs, exp, info = run_both(ts, [(f, fv)], 1000, base_aggs + [(k, 1) for k in TIME_AGGS])
    assert info.long_windows == info.num_windows
    # reducers that read no value column at all (no column pass: only the windows' records)
    outs, exp, info = run_both(ts, [(f, fv)], 1000, [("WindowStart", 0), ("NumRows", 1), ("NumRows", 0)])
    assert info.long_windows == info.num_windows
    cols = [capi.Column(ts), capi.Column(f, np.packbits(fv, bitorder="little"), capi.FLOAT64, 0, len(ts), -1)]
    capi.rolling_aggregate(cols, 0, 1000, [("WindowStart", 0), ("ArithmeticMean", 1), ("Sum", 1), ("Count", 1)])
    assert capi.last_kernel_name() == "long_stream_kernel"
    capi.rolling_aggregate(cols, 0, 1000, base_aggs)       # extrema / first / last / time-weighted terms: the streaming form too
    assert capi.last_kernel_name() == "long_stream_kernel"
    with capi.route(capi.ROUTE_LONG_CLASSIC):               # (the bisection form stays reachable)
        capi.rolling_aggregate(cols, 0, 1000, base_aggs)
        assert capi.last_kernel_name() == "long_partial_kernel"


def test_nanosecond_timestamps_beyond_2_53():
    # nanosecond timestamps far above 2^53, where float64(ts) is lossy - and that lossy value is what the time-weighted
    # reducers read (integral.go:17,:49) - positive and negative; 1 s windows (tile kernels) and 1 day windows (long-window path).
    # (Rows within one interval of INT64_MAX / INT64_MIN make s_k + interval wrap in the reference itself: not a defined input.)
    rng = np.random.default_rng(3)
    aggs = [(k, 0 if k == "WindowStart" else 1) for k in ALL_AGGS]
    tw = [("WindowStart", 0)] + [(k, 1) for k in TIME_AGGS]
    for base, interval, offset in [(1_700_000_000_000_000_000, 1_000_000_000, 0), (1_700_000_000_000_000_000, 86_400_000_000_000, 5),
                                   (-1_700_000_000_000_000_000, 1_000_000_007, -3)]:
        n = 60_000
        ts = np.sort(rng.integers(0, 200_000_000_000_000, n)).astype(np.int64) + np.int64(base)
        vals, valid = make_vals(rng, n, "f64", 0.2)
        run_both(ts, [(vals, valid)], interval, aggs, offset=offset)
        for inclusive in (False, True):
            run_both(ts, [(vals, valid)], interval, tw, offset=offset, inclusive=inclusive)
    # dense nanosecond data (rows ~0.1 ms apart, 1.024 ms = 2^13 * 125 ns windows): the wave-tile kernels take it with window ids
    # relative to each tile's first window ((ts - tile base) >> 13 fits 32 bits) although the rows span 6e9 ns from the first window
    n = 60_000
    ts = np.cumsum(rng.integers(50_000, 150_000, n)).astype(np.int64) + np.int64(1_700_000_000_000_000_000)
    vals, valid = make_vals(rng, n, "f64", 0.2)
    run_both(ts, [(vals, valid)], 1_024_000, aggs, offset=77)
    run_both(ts, [(vals, valid)], 1_024_000, tw, offset=77, inclusive=True)
    cols = [capi.Column(ts), capi.Column(vals, np.packbits(valid, bitorder="little"), capi.FLOAT64, 0, n, -1)]
    capi.rolling_aggregate(cols, 0, 1_024_000, aggs, offset=77)
    assert capi.last_kernel_name() == "rolling_simple_kernel"
    capi.rolling_aggregate(cols, 0, 1_024_000, tw, offset=77)
    assert capi.last_kernel_name() == "rolling_tw_kernel"


def test_declines_unsorted_timestamps_and_what_null_timestamps_cannot_have():
    ts = np.array([1, 5, 3, 9, 12, 20], dtype=np.int64)
    v = np.arange(6, dtype=np.float64)
    with pytest.raises(capi.BowGpuError) as e:
        capi.rolling_aggregate([capi.Column(ts), capi.Column(v)], 0, 4, [("WindowStart", 0), ("Sum", 1)])
    assert e.value.code == -14
    # null timestamps: served on the device (test_null_timestamps_*: exclusive and inclusive windows, NumRows), declined -
    # BOWGPU_ERR_TS_NULLS - for Mode
    tsn = capi.Column.from_list([1, None, 3, 9], "int64")
    with pytest.raises(capi.BowGpuError) as e:
        capi.rolling_aggregate([tsn, capi.Column(v[:4])], 0, 4, [("WindowStart", 0), ("Mode", 1)])
    assert e.value.code == -13
    with pytest.raises(capi.BowGpuError) as e:  # first ts null is the reference's own ctor error (rolling.go:89-93)
        capi.rolling_aggregate([capi.Column.from_list([None, 2, 3, 9], "int64"), capi.Column(v[:4])], 0, 4,
                               [("WindowStart", 0), ("Sum", 1)])
    assert e.value.code == -3


def test_deterministic_and_generator_parity():
    n = 1_000_000
    ts_d, val_d = capi.gen_dense(0, n, seed=42)
    ts_o, val_o = orc.gen_dense(0, n, seed=42)
    assert np.array_equal(ts_d.values.to_numpy(np.int64, n), ts_o)
    assert np.array_equal(val_d.values.to_numpy(np.float64, n), val_o)
    aggs = [("WindowStart", 0), ("ArithmeticMean", 1), ("Sum", 1), ("Min", 1), ("Max", 1)]
    a, _ = capi.rolling_aggregate([ts_d, val_d], 0, 10, aggs)
    b, _ = capi.rolling_aggregate([ts_d, val_d], 0, 10, aggs)
    exp, _ = orc.aggregate([orc.Column(ts_o, None, orc.INT64), orc.Column(val_o, None, orc.FLOAT64)], 0, 10, aggs)
    for (k, _), x, y, w in zip(aggs, a, b, exp):
        assert np.array_equal(x.host_arrays()[0].view(np.uint64), y.host_arrays()[0].view(np.uint64))
        compare(k, x, w)
    # sparse generator
    ts_s, val_s = capi.gen_sparse(0, n, seed=3)
    ts_so, val_so, bm_so = orc.gen_sparse(0, n, seed=3)
    assert np.array_equal(ts_s.values.to_numpy(np.int64, n), ts_so)
    assert np.array_equal(val_s.values.to_numpy(np.float64, n), val_so)
    assert np.array_equal(val_s.validity.to_numpy(np.uint8, (n + 7) // 8), bm_so)
    outs, _ = capi.rolling_aggregate([ts_s, val_s], 0, 100, aggs, offset=7)
    exp, _ = orc.aggregate([orc.Column(ts_so, None, orc.INT64), orc.Column(val_so, bm_so, orc.FLOAT64)], 0, 100, aggs, offset=7)
    for (k, _), g, w in zip(aggs, outs, exp):
        compare(k, g, w)


def test_rows_below_first_window_start():
    # Negative ts: Go's truncating division can leave s0 ABOVE ts[0] (rolling.go:96-99); the rows
    # below s0 are skipped-but-spanned by the scan (rolling.go:194-196) and ride in window 0 -
    # unless window 0 has no row of its own, in which case it is an empty slice.
    for ts_list, interval, offset, inclusive in [
        ([-37, -36, -35, -30, -20, -5, 3], 10, -4, False),   # s0 = -34 > -37, window 0 has own rows
        ([-37, -36, -35, -20, -5, 3], 10, -4, False),        # window 0 = only rows below s0 => empty
        ([-37, -36, -24, -5, 3], 10, -4, True),              # only an inclusive row at s0+I
        ([-19, -19, -18, -3, 0, 4, 8], 10, 9, True),
    ]:
        ts = np.array(ts_list, dtype=np.int64)
        v = np.arange(len(ts), dtype=np.float64) + 0.5
        aggs = [("WindowStart", 0), ("Sum", 1), ("NumRows", 1), ("First", 1), ("ArithmeticMean", 1),
                ("IntegralTrapezoid", 1), ("IntegralStep", 1)]
        outs, exp, info = run_both(ts, [(v, None)], interval, aggs, offset=offset, inclusive=inclusive)
        assert info.s0 > ts[0]
        for (k, _), g, w in zip(aggs, outs, exp):
            compare("%s %s" % (k, ts_list), g, w)


@pytest.mark.parametrize("vkind", ["f64", "i64"])
def test_simple_kernel_shapes(vkind):
    # the shapes rolling_simple.hip takes: one null-free column, <= 4 factor-free outputs, 32-bit span
    rng = np.random.default_rng(41)
    for mode in ["dense", "irregular", "dups", "gappy"]:
        for n, interval, offset in [(1, 10, 0), (513, 10, 0), (640, 7, 3), (641, 10, 0), (100_000, 10, 0), (100_000, 3, 2), (100_001, 100, 7)]:
            ts = make_ts(rng, n, mode)
            vals, _ = make_vals(rng, n, vkind, 0.0)
            valid = rng.random(n) >= 0.5  # heavy nulls: some windows have no valid value at all
            run_both(ts, [(vals, valid)], interval, [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1),
                                                      ("Count", 1), ("First", 1), ("Last", 1)], offset=offset)
            for aggs in ([("WindowStart", 0), ("ArithmeticMean", 1)],
                         [("WindowStart", 0), ("Sum", 1), ("Min", 1), ("Max", 1)],
                         [("Count", 1), ("WindowStart", 0), ("First", 1), ("Last", 1)],
                         [("WindowStart", 0), ("NumRows", 1), ("ArithmeticMean", 1)],
                         [("WindowStart", 0), ("Sum", 0), ("Max", 0)]):
                run_both(ts, [(vals, None)], interval, aggs, offset=offset)


def test_simple_kernel_redo_when_tile_too_dense():
    # one window per row: more heads than the simple kernel's segment list holds => redone by the lean kernel
    ts = np.arange(5000, dtype=np.int64) * 3
    vals = np.arange(5000, dtype=np.float64)
    run_both(ts, [(vals, None)], 2, [("WindowStart", 0), ("Sum", 1), ("Count", 1)])


def test_simple_kernel_redo_when_ids_overflow():
    # > 65535 windows inside one 640-row tile: the simple kernel flags the call, the lean kernel redoes it
    ts = np.concatenate([np.arange(0, 300), np.arange(300, 600) * 1000]).astype(np.int64)
    vals = np.arange(len(ts), dtype=np.float64)
    run_both(ts, [(vals, None)], 2, [("WindowStart", 0), ("ArithmeticMean", 1), ("Count", 1)])


@pytest.mark.parametrize("rows_per_window", [14, 40, 100, 128, 150, 200, 250])
def test_compacting_kernel_for_nullable_columns_under_time_weighted_reducers(rows_per_window):
    """rolling_twc_kernel (round 5): a nullable column's valid points are compacted in LDS, the integrals' terms come from the dense
    neighbour logic, one lane adds a window's terms in row order - bit for bit the oracle, like the row-space form of rolling_tw.hip it
    replaces (run_both pushes every case through both: capi.ROUTE_TW_ROWS).  Window lengths either side of the two look-aheads (128 /
    256 rows), one and both kinds of integral, value reducers next to them, Int64 and multi-column calls, inclusive windows by option
    and by reducer, Factor chains, null densities from a few to nearly all, tiles with an all-null stretch."""
    rng = np.random.default_rng(500 + rows_per_window)
    n = 260_000
    ts = np.cumsum(rng.integers(1, 20, n)).astype(np.int64) + 777
    interval = 10 * rows_per_window
    for null_frac in (0.3, 0.02, 0.93):
        f, fm = make_vals(rng, n, "f64", null_frac)
        g, gm = make_vals(rng, n, "i64", 0.5)
        fm[100_000:103_000] = False       # a stretch without a valid point: windows with no point, tiles with none
        sets = [[("WindowStart", 0), ("WeightedAverageStep", 1)],
                [("WindowStart", 0)] + [(k, 1) for k in TIME_AGGS],
                [("WindowStart", 0), ("IntegralTrapezoid", 1), ("ArithmeticMean", 1), ("Min", 1), ("Last", 1), ("Count", 1), ("NumRows", 1)],
                [("WindowStart", 0), ("IntegralStep", 2, [0.1]), ("WeightedAverageLinear", 1, [2.0, -1.0]), ("First", 2), ("Max", 2), ("Sum", 1)],
                [("WindowStart", 0), ("Sum", 1), ("Min", 1), ("Max", 1)]]
        for aggs in sets:
            for inclusive in (False, True):
                outs, exp, info = run_both(ts, [(f, fm), (g, gm)], interval, aggs, offset=7, inclusive=inclusive)
                assert info.long_windows == 0 or rows_per_window >= 100   # (irregular rows: a few windows outgrow the look-ahead and take the cooperative path)
    # a frame that is mostly long windows with a stretch of one-row windows: more heads than the compacting kernel's list holds in
    # some tiles - the call is redone by the row-space form (and by the forms behind it); the answer is the same
    ts2 = ts.copy()
    ts2[50_000:50_400] = ts2[50_000] + np.arange(400) * interval
    ts2[50_400:] += 400 * interval
    run_both(ts2, [(f, fm)], interval, [("WindowStart", 0), ("WeightedAverageStep", 1), ("IntegralTrapezoid", 1)])


@pytest.mark.parametrize("W", [262_143, 262_144, 262_145, 300_001])
def test_null_counts_around_the_size_where_the_finish_launch_reports_to_the_host_itself(W):
    """up to 262 144 windows (common.h kFinishHostBits) ONE workgroup per output finishes the bitmap and stores the status words and
    the valid counts straight into the host's registered block (round 5: no copy command behind a small call); above it the counts are
    summed with atomics and copied.  Both sides of the boundary: outputs, null counts, and the unsorted-column status through that path."""
    rng = np.random.default_rng(W)
    n = W * 2 - 1                                    # two rows per window, the last window one
    ts = np.arange(n, dtype=np.int64)
    v, valid = make_vals(rng, n, "f64", 0.6)         # ~16 % of the windows hold no value: Mean / Min nil there
    aggs = [("WindowStart", 0), ("ArithmeticMean", 1), ("Min", 1), ("Count", 1), ("Last", 1)]
    for device in (False, True):
        outs, exp, info = run_both(ts, [(v, valid)], 2, aggs, device=device)
        assert outs[0].length == W
        for g, w in zip(outs, exp):
            assert g.null_count == w.length - int(w.valid_mask().sum())
    bad = ts.copy()
    bad[n // 2] = 0
    with pytest.raises(capi.BowGpuError) as e:
        capi.rolling_aggregate([capi.Column(bad), capi.Column(v, np.packbits(valid, bitorder="little"), capi.FLOAT64, 0, n, -1)], 0, 2, aggs)
    assert e.value.code == -14


def test_slow_route_counter_tells_a_caller_when_a_fallback_kernel_served_the_call():
    """bowgpu_last_call_slow_rows: 0 for the usual call; the call's rows when rolling_agg_kernel (here forced; in the product: the redo of
    a tile no wave-tile kernel can describe, intervals >= 2^32 with time-weighted reducers) or interp_tile_kernel served it"""
    n = 50_000
    ts = np.arange(n, dtype=np.int64) * 3
    v = np.arange(n, dtype=np.float64)
    cols = [capi.Column(ts), capi.Column(v, None, capi.FLOAT64)]
    aggs = [("WindowStart", 0), ("ArithmeticMean", 1)]
    capi.rolling_aggregate(cols, 0, 30, aggs)
    assert capi.last_call_slow_rows() == 0 and capi.last_kernel_name() == "rolling_simple_kernel"
    with capi.route(capi.ROUTE_FORCE_GENERAL | capi.ROUTE_NO_LONG_ONLY):
        capi.rolling_aggregate(cols, 0, 30, aggs)
        assert capi.last_call_slow_rows() == n and capi.last_kernel_name() == "rolling_agg_kernel"
    # the product's own way there: window ids 2^16 apart inside one tile (the simple kernel flags the call, the wave kernel redoes it - fast)
    ts2 = np.concatenate([np.arange(0, 300), np.arange(300, 600) * 1_000_000]).astype(np.int64)
    capi.rolling_aggregate([capi.Column(ts2), capi.Column(v[:600].copy(), None, capi.FLOAT64)], 0, 2, aggs)
    assert capi.last_call_slow_rows() == 0
    ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
    capi.rolling_interpolate(cols, 0, 30, ip)
    assert capi.last_call_slow_rows() == 0
    with capi.route(capi.ROUTE_INTERP_TILE):
        capi.rolling_interpolate(cols, 0, 30, ip)
        assert capi.last_call_slow_rows() == n
    capi.rolling_aggregate(cols, 0, 30, aggs)
    assert capi.last_call_slow_rows() == 0
    # ... and the fused Interpolate -> Aggregate call speaks of ITSELF (it used to leave the previous call's figure standing)
    with capi.route(capi.ROUTE_FORCE_GENERAL | capi.ROUTE_NO_LONG_ONLY):
        capi.rolling_aggregate(cols, 0, 30, aggs)
        assert capi.last_call_slow_rows() == n
    big = 200_000
    tsb = np.cumsum(np.random.default_rng(3).integers(1, 9, big)).astype(np.int64)
    colsb = [capi.Column(tsb), capi.Column(np.arange(big, dtype=np.float64), None, capi.FLOAT64)]
    capi.rolling_interpolate_aggregate(colsb, 0, 100, ip, aggs)
    assert capi.last_kernel_name() == "rolling_fused_kernel" and capi.last_call_slow_rows() == 0


# ------------------------------------------------------------------ aggregation.Mode (mode.go:8-32)
def _mode_values(rng, n, kind):
    if kind == "int":
        v = rng.integers(-3, 4, n).astype(np.int64)
        v[rng.random(n) < 0.02] = np.iinfo(np.int64).min
        return v
    v = rng.integers(-2, 3, n).astype(np.float64) / 2
    r = rng.random(n)
    v[r < 0.05] = np.nan          # every NaN is a map key of its own
    v[(r >= 0.05) & (r < 0.10)] = -0.0   # -0 == +0 as keys; the row's own value comes back
    v[(r >= 0.10) & (r < 0.12)] = np.inf
    return v


@pytest.mark.parametrize("kind", ["float", "int"])
@pytest.mark.parametrize("interval", [3, 10, 40, 400, 9000])
def test_mode_against_oracle(kind, interval):
    """ties (few distinct values), nulls, NaN / -0 keys; intervals that put windows in the lane-per-window class (<= 32 rows),
    the workgroup class (<= 7680 rows) and mixes of both"""
    rng = np.random.default_rng(interval * 7 + (kind == "int"))
    n = 30_000 if interval <= 400 else 60_000
    ts = np.cumsum(rng.integers(0, 3, n)).astype(np.int64) - 50
    vals = _mode_values(rng, n, kind)
    valid = rng.random(n) > 0.3
    valid[100:300] = False         # all-null windows
    aggs = [("WindowStart", 0), ("Mode", 1), ("Count", 1), ("Mode", 0), ("Mode", 2, [2.5]), ("ArithmeticMean", 2)]
    outs, exp, info = run_both(ts, [(vals, valid), (vals, None)], interval, aggs, offset=1)
    assert outs[1].type == (capi.INT64 if kind == "int" else capi.FLOAT64)
    assert outs[3].type == capi.INT64


def test_mode_long_windows_sort_path():
    """windows of more than 7680 rows: select valid rows -> key -> stable radix sort -> run lengths"""
    rng = np.random.default_rng(5)
    n = 50_000
    ts = np.arange(n, dtype=np.int64)
    for kind in ("float", "int"):
        vals = _mode_values(rng, n, kind)
        valid = rng.random(n) > 0.2
        valid[20_000:40_000] = rng.random(20_000) > 0.999     # a long window with a handful of valid rows
        outs, exp, info = run_both(ts, [(vals, valid), (vals, None)], 10_000, [("WindowStart", 0), ("Mode", 1), ("Mode", 2)])
        assert info.long_windows >= 5
    # all values distinct: every count is 1, the first valid row wins
    vals = rng.permutation(n).astype(np.float64)
    valid = np.ones(n, dtype=bool)
    valid[:7] = False
    outs, exp, info = run_both(ts, [(vals, valid)], 25_000, [("WindowStart", 0), ("Mode", 1)])
    assert outs[1].to_list()[0] == vals[7]
    # no valid row at all in a long window
    valid[:] = False
    run_both(ts, [(vals, valid)], 25_000, [("WindowStart", 0), ("Mode", 1)])


def test_mode_pre_rows_inclusive_and_alone():
    """rows below s0 ride in window 0; an inclusive call (IntegralTrapezoid) hands Mode the window without its extra row;
    Mode as the only kind of reducer of a call"""
    rng = np.random.default_rng(11)
    n = 5000
    ts = (np.cumsum(rng.integers(1, 4, n)) - 3000).astype(np.int64)
    vals = _mode_values(rng, n, "float")
    valid = rng.random(n) > 0.25
    run_both(ts, [(vals, valid)], 7, [("WindowStart", 0), ("Mode", 1), ("IntegralTrapezoid", 1)], offset=3)
    run_both(ts, [(vals, valid)], 7, [("Mode", 0), ("Mode", 1)], offset=3)
    run_both(ts, [(vals, valid)], 7, [("Mode", 0), ("Mode", 1)], offset=3, device=True)
    run_both(ts[:0], [(vals[:0], valid[:0])], 7, [("Mode", 0), ("Mode", 1)])


def test_mode_whole_frame():
    rng = np.random.default_rng(3)
    for n in (1, 20, 3000, 20_000):
        ts = np.arange(n, dtype=np.int64)
        vals = _mode_values(rng, n, "float")
        valid = rng.random(n) > 0.3
        bm = np.packbits(valid, bitorder="little")
        ccols = [capi.Column(ts, None, capi.INT64), capi.Column(vals, bm, capi.FLOAT64, 0, n, -1)]
        ocols = [orc.Column(ts, None, orc.INT64), orc.Column(vals, bm, orc.FLOAT64)]
        aggs = [("Mode", 1), ("Sum", 1), ("Mode", 0), ("Mode", 1, [3.0])]
        got = capi.aggregate_whole(ccols, 0, aggs)
        want = orc.aggregate_whole(ocols, 0, aggs)
        for k, g, w in zip(_names(aggs), got, want):
            # (whole-frame Sum: one window over all rows, reduced as a tree - tests/tolerance.py)
            compare("whole %s n=%d" % (k, n), g, w, exact=k != "Sum", bound=[2.0 * (n + 2) * 2.0 ** -53 * float(np.abs(vals[valid]).sum())])


def _null_ts_frame(rng, n, null_frac, mode="irregular", vnull=0.2):
    ts = make_ts(rng, n, mode)
    tvalid = rng.random(n) >= null_frac
    tvalid[0] = True                       # (the constructor needs a first timestamp: rolling.go:89-93)
    tvalid[-1] = True                      # (a null LAST timestamp ends the iteration before it starts: its own test)
    vals = np.round(rng.standard_normal(n) * 100, 3)
    vvalid = rng.random(n) >= vnull
    return ts, tvalid, vals, vvalid


NULL_TS_AGGS = [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1), ("Count", 1), ("First", 1), ("Last", 1),
                ("IntegralStep", 1), ("WeightedAverageStep", 1), ("Sum", 0), ("Count", 0), ("Last", 0), ("WeightedAverageStep", 0)]


# ... and what an INCLUSIVE iteration adds (Options.Inclusive, or one of the two reducers that ask for it - aggregation.go:183-185):
# the two themselves, NumRows, a Factor on one of them
NULL_TS_AGGS_INCL = NULL_TS_AGGS + [("NumRows", 1), ("IntegralTrapezoid", 1), ("WeightedAverageLinear", 1), ("IntegralTrapezoid", 0),
                                    ("WeightedAverageLinear", 1, [2.5])]


def _rows_on_a_start_with_a_null_behind(ts, tvalid, s0, interval):
    """the rows ts_nulls.hip singles out after an inclusive iteration: on a window start (not window 0's), the first with that
    timestamp, the next row's timestamp null (the next window begins at `rowIndex - 1`, rolling.go:214-218)"""
    n = len(ts)
    idx = np.arange(n)
    prev = np.maximum.accumulate(np.where(tvalid, idx, -1))
    hit = np.zeros(n, bool)
    hit[:-1] = tvalid[:-1] & ~tvalid[1:] & (ts[:-1] >= s0 + interval) & ((ts[:-1] - s0) % interval == 0)
    pp = np.concatenate(([-1], prev[:-1]))
    hit &= (pp < 0) | (ts[np.maximum(pp, 0)] < ts)
    return int(hit.sum())


def _run_null_ts(ts, tvalid, vals, vvalid, interval, offset=0, aggs=NULL_TS_AGGS, device=False, inclusive=False):
    n = len(ts)
    tbm = np.packbits(tvalid, bitorder="little")
    vbm = None if vvalid is None else np.packbits(vvalid, bitorder="little")
    ccols = [capi.Column(ts, tbm, capi.INT64, 0, n, -1), capi.Column(vals, vbm, capi.FLOAT64, 0, n, -1)]
    if device:
        ccols = [c.to_device() for c in ccols]
    ocols = [orc.Column(ts, tbm, orc.INT64), orc.Column(vals, vbm, orc.FLOAT64)]
    want, nic = orc.aggregate(ocols, 0, interval, aggs, offset=offset, inclusive=inclusive)
    bounds = None
    for label in capi.agg_routes():
        got, info = capi.rolling_aggregate(ccols, 0, interval, aggs, offset=offset, inclusive=inclusive,
                                           out_residency=capi.DEVICE if device else capi.HOST)
        assert info.new_interval_col == nic
        if info.long_windows and bounds is None:
            # (the bound's magnitudes: the oracle over |x| on the SAME frame, null timestamps included)
            bounds = order_free_bounds(ocols, 0, interval, aggs, offset=offset, ref=want, inclusive=inclusive)
        for i, (a, g, w) in enumerate(zip(aggs, got, want)):
            k, _c = a[0], a[1]
            exact = info.long_windows == 0 or k not in ORDER_SENSITIVE
            compare("null ts %s col %d path=%s n=%d I=%d incl=%d" % (k, _c, label, n, interval, inclusive), g, w, exact=exact,
                    bound=None if exact else bounds[i])
    return want


@pytest.mark.parametrize("null_frac", [0.02, 0.3, 0.9])
def test_null_timestamps_against_the_oracle(null_frac):
    """rolling.go:190-193 skips a row whose interval value is null; it stays inside its window's slice when taken rows surround it
    (its value columns are reduced) and belongs to no window behind the window's last taken row.  The device path (ts_nulls.hip)
    against the oracle's literal walk: every exclusive-window reducer incl. the step integrals and reducers over the interval
    column itself, every kernel route, windows shorter and longer than the null runs."""
    rng = np.random.default_rng(int(null_frac * 100))
    for n, interval, offset, mode in [(1, 10, 0, "dense"), (2, 3, 1, "dense"), (700, 5, 2, "dups"), (3000, 10, 0, "irregular"),
                                      (5000, 64, 7, "gappy"), (40_000, 25, -3, "irregular"), (40_000, 4000, 11, "dense"),
                                      (6000, 10, 3, "negative")]:
        ts, tvalid, vals, vvalid = _null_ts_frame(rng, n, null_frac, mode)
        _run_null_ts(ts, tvalid, vals, vvalid, interval, offset, device=(n % 2 == 0))


@pytest.mark.parametrize("null_frac", [0.02, 0.3, 0.9])
def test_null_timestamps_inclusive_windows_against_the_oracle(null_frac):
    """An inclusive iteration over an interval column with nulls (rolling.go:201-218): the window also takes the first row ON its
    end, the null rows in front of that row lie inside its slice, and the next window starts at `rowIndex - 1` - the inclusive row
    itself, or, when null rows follow it, the last of THEM: that window then begins without the row on its start.
    Every reducer incl. NumRows and the two that ask for inclusive windows, Options.Inclusive set or only implied, every route;
    frames where rows sit on window starts all the time (dense / dups timestamps, small intervals) so the A.5 case is common."""
    rng = np.random.default_rng(1000 + int(null_frac * 100))
    seen = 0
    for n, interval, offset, mode in [(1, 10, 0, "dense"), (2, 3, 1, "dense"), (700, 5, 2, "dups"), (3000, 10, 0, "irregular"),
                                      (5000, 64, 7, "gappy"), (40_000, 25, -3, "irregular"), (40_000, 4000, 11, "dense"),
                                      (6000, 10, 3, "negative"), (5000, 1, 0, "dups"), (20_000, 4, 1, "dense")]:
        ts, tvalid, vals, vvalid = _null_ts_frame(rng, n, null_frac, mode)
        s0, _W = orc.plan_windows(orc.Column(ts, np.packbits(tvalid, bitorder="little"), orc.INT64), interval, offset)
        seen += _rows_on_a_start_with_a_null_behind(ts, tvalid, s0, interval)
        _run_null_ts(ts, tvalid, vals, vvalid, interval, offset, aggs=NULL_TS_AGGS_INCL, device=(n % 2 == 0), inclusive=(n % 3 == 0))
        # Options.Inclusive alone, no reducer that asks for it: everything is read through UnsetInclusive, but the slices are the inclusive iteration's
        _run_null_ts(ts, tvalid, vals, vvalid, interval, offset, aggs=NULL_TS_AGGS + [("NumRows", 0)], device=(n % 2 == 1), inclusive=True)
    assert seen > 50


def test_null_timestamps_inclusive_large_frame_on_the_device():
    """3e6 rows, 4 % null timestamps, dense timestamps and an interval of 8: every eighth row sits on a window start and one in 25 of those
    has a null behind it - thousands of windows recomputed by ts_quirk_fix_kernel, the rest by the ordinary tile kernel"""
    rng = np.random.default_rng(4)
    n = 3_000_000
    ts, tvalid, vals, vvalid = _null_ts_frame(rng, n, 0.04, "dense", vnull=0.1)
    s0, _W = orc.plan_windows(orc.Column(ts, np.packbits(tvalid, bitorder="little"), orc.INT64), 8, 3)
    assert _rows_on_a_start_with_a_null_behind(ts, tvalid, s0, 8) > 5000
    _run_null_ts(ts, tvalid, vals, vvalid, 8, 3, aggs=NULL_TS_AGGS_INCL, device=True)


def test_null_timestamps_numrows_counts_the_rows_of_the_slice():
    ts = np.array([10, 11, 12, 13, 20, 21, 22, 30, 31, 45, 46], dtype=np.int64)
    vals = np.arange(1.0, 12.0)
    tvalid = np.ones(len(ts), bool)
    tvalid[[1, 2]] = False      # between taken rows of window [10, 20): inside the slice
    tvalid[[5, 6]] = False      # behind the last taken row of [20, 30): in no slice
    want = _run_null_ts(ts, tvalid, vals, None, 10, aggs=[("WindowStart", 0), ("NumRows", 1), ("Count", 1)])
    assert want[1].to_list() == [4.0, 1.0, 2.0, 2.0] and want[2].to_list() == [4, 1, 2, 2]
    # the A.5 case by hand: 20 sits on a start, a null behind it -> [20, 30) begins at the null row in front of 21, without row 20
    tvalid = np.ones(len(ts), bool)
    tvalid[5] = False
    ts2 = ts.copy(); ts2[5] = 20; ts2[6] = 21
    want = _run_null_ts(ts2, tvalid, vals, None, 10, aggs=[("WindowStart", 0), ("NumRows", 1), ("First", 1), ("IntegralTrapezoid", 1)])
    assert want[1].to_list()[1] == 2.0 and want[2].to_list()[1] == 6.0
    # a planned call (bowgpu_plan_windows_ex + bowgpu_rolling_aggregate_planned) over the same frame takes the same path
    tbm = np.packbits(tvalid, bitorder="little")
    ccols = [capi.Column(ts2, tbm, capi.INT64, 0, len(ts2), -1).to_device(), capi.Column(vals).to_device()]
    plan = capi.plan_windows_ex(ccols[0], 10, 0)
    aggs = [("WindowStart", 0), ("NumRows", 1), ("First", 1), ("IntegralTrapezoid", 1)]
    got, _info = capi.rolling_aggregate(ccols, 0, 10, aggs, plan=plan)
    for (k, _c), g, w in zip(aggs, got, want):
        compare("planned null ts %s" % k, g, w)


def test_null_timestamps_runs_at_window_edges_and_the_null_last_row():
    ts = np.array([10, 11, 12, 13, 20, 21, 22, 30, 31, 45, 46], dtype=np.int64)
    vals = np.arange(1.0, 12.0)
    allv = np.ones(len(ts), bool)
    for nulls in ([1], [3], [4], [3, 4], [2, 3, 4, 5], [1, 2, 3, 4, 5, 6, 7, 8, 9], [7, 8], [9]):
        tvalid = allv.copy()
        tvalid[nulls] = False
        want = _run_null_ts(ts, tvalid, vals, None, 10)
        _run_null_ts(ts, tvalid, vals, None, 10, offset=5)
    # window [10, 20) with its rows 12, 13 null: they trail the window's last taken row -> in no slice: Count 2, not 4
    tvalid = allv.copy(); tvalid[[2, 3]] = False
    want = _run_null_ts(ts, tvalid, vals, None, 10)
    assert want[5].to_list()[0] == 2 and want[1].to_list()[0] == 1.0 + 2.0
    # ... but null rows BETWEEN taken rows are reduced: rows 11, 12 null, 13 valid -> Count 4
    tvalid = allv.copy(); tvalid[[1, 2]] = False
    want = _run_null_ts(ts, tvalid, vals, None, 10)
    assert want[5].to_list()[0] == 4 and want[10].to_list()[0] == 10.0 + 13.0   # (Sum over the interval column itself skips them)
    # the physically last timestamp null: HasNext is false from the start (rolling.go:162-173) - numWindows slots, all nil
    tvalid = allv.copy(); tvalid[-1] = False
    want = _run_null_ts(ts, tvalid, vals, None, 10)
    assert want[0].length == 4 and all(x is None for w in want for x in w.to_list())   # (windows 10, 20, 30, 40: the last valid ts is 45)
    tvalid[-3:] = False
    want = _run_null_ts(ts, tvalid, vals, None, 10)
    assert want[0].length == 3 and all(x is None for w in want for x in w.to_list())


def test_null_timestamps_large_frame_on_the_device():
    """2e6 rows, 5 % null timestamps, device-resident: the rewritten call takes the ordinary tile kernel"""
    rng = np.random.default_rng(99)
    n = 2_000_000
    ts, tvalid, vals, vvalid = _null_ts_frame(rng, n, 0.05, "irregular", vnull=0.1)
    _run_null_ts(ts, tvalid, vals, vvalid, 20, 3, aggs=NULL_TS_AGGS[:10], device=True)
    with capi.route(0):
        pass
    assert capi.last_kernel_name() == "rolling_agg_kernel"   # (agg_routes ends with the general kernel)


def test_mode_only_call_declines_what_the_device_path_declines():
    """a call made of Mode reducers alone never reaches the tile kernels: the same contract errors must come from its own pass"""
    ts = np.array([1, 2, 5, 4, 9, 12], dtype=np.int64)
    v = np.array([1.0, 1.0, 2.0, 2.0, 2.0, 3.0])
    with pytest.raises(capi.BowGpuError) as e:
        capi.rolling_aggregate([capi.Column(ts), capi.Column(v)], 0, 4, [("Mode", 0), ("Mode", 1)])
    assert "TS_UNSORTED" in str(e.value)
    bm = np.packbits(np.array([1, 1, 0, 1, 1, 1], dtype=bool), bitorder="little")
    with pytest.raises(capi.BowGpuError) as e:
        capi.rolling_aggregate([capi.Column(np.sort(ts), bm, capi.INT64, 0, 6, -1), capi.Column(v)], 0, 4, [("Mode", 0), ("Mode", 1)])
    assert "TS_NULLS" in str(e.value)
    with pytest.raises(capi.BowGpuError) as e:   # aggregation.go:163-166: the interval column must be kept
        capi.rolling_aggregate([capi.Column(np.sort(ts)), capi.Column(v)], 0, 4, [("Mode", 1)])
    assert "KEEP_INTERVAL" in str(e.value)


def test_long_window_final_split():
    """long windows are finished by one lane when they have at most 8 chunk partials (4096 rows each) and at most 8 empty windows
    behind them, else by a workgroup: both sides of both limits, with nulls, for every reducer"""
    rng = np.random.default_rng(21)
    I = 1_000_000
    lens = [32768, 32769, 4096, 4097, 20_000, 70_000, 8 * 4096 - 1, 9 * 4096]
    gaps = [0, 8, 9, 1, 40, 0, 8, 9]
    ts_parts, t0 = [], 0
    for ln, gap in zip(lens, gaps):
        ts_parts.append(t0 + np.sort(rng.integers(0, I, ln)))
        t0 += (gap + 1) * I
    ts = np.concatenate(ts_parts).astype(np.int64)
    n = len(ts)
    vals = np.round(rng.standard_normal(n) * 100, 2)
    valid = rng.random(n) > 0.2
    aggs = [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1), ("Count", 1), ("First", 1), ("Last", 1),
            ("NumRows", 1), ("IntegralStep", 1), ("WeightedAverageLinear", 1)]
    outs, exp, info = run_both(ts, [(vals, valid)], I, aggs)
    assert info.long_windows >= len(lens)
    assert outs[0].length == sum(g + 1 for g in gaps[:-1]) + 1


@pytest.mark.parametrize("kind", ["int", "float"])
def test_mode_size_class_boundaries(kind):
    """windows of exactly 32 / 33 rows (lane | wavefront), 256 / 257 rows (wavefront | workgroup), 2048 / 2560 / 2561 rows (hash table at its fullest | the scan form),
    7680 / 7681 rows (scan | radix sort): mostly distinct values (a full table, long probe chains), a few planted repeats,
    Int64 -1 (the table's empty marker, counted apart) and NaNs"""
    rng = np.random.default_rng(31 + (kind == "int"))
    lens = [32, 33, 2048, 2049, 2560, 2561, 7680, 7681, 100, 2300, 64, 65, 255, 256, 257, 129]
    I = 100_000
    ts = np.concatenate([k * I + np.sort(rng.choice(I, ln, replace=False)) for k, ln in enumerate(lens)]).astype(np.int64)
    n = len(ts)
    if kind == "int":
        vals = rng.integers(-10 ** 9, 10 ** 9, n).astype(np.int64)
        vals[rng.random(n) < 0.02] = -1
    else:
        vals = rng.standard_normal(n) * 1e6
        vals[rng.random(n) < 0.02] = np.nan
        vals[rng.random(n) < 0.01] = -0.0
        vals[rng.random(n) < 0.01] = 0.0
    for _ in range(40):   # planted repeats, some of them late in their window
        i = int(rng.integers(0, n - 50))
        vals[i + int(rng.integers(1, 50))] = vals[i]
    valid = rng.random(n) > 0.1
    outs, exp, info = run_both(ts, [(vals, valid), (vals, None)], I, [("WindowStart", 0), ("Mode", 1), ("Mode", 2), ("Count", 1)])
    assert outs[0].length == len(lens)


def test_mode_windows_beyond_the_global_table():
    """windows of more than 2e6 rows take the radix-sort path; the oracle's Mode is quadratic, so the expectation is numpy's:
    among the values with the largest count, the one whose LAST occurrence comes first (= the row at which a count first reaches
    the maximum); a smaller window beside them goes through the global-memory table"""
    rng = np.random.default_rng(12)
    n1, n2 = 2_300_000, 900_000
    ts = np.concatenate([np.sort(rng.integers(0, 10 ** 9, n1)), 10 ** 9 + np.sort(rng.integers(0, 10 ** 9, n2))]).astype(np.int64)
    n = n1 + n2
    vals = rng.integers(0, 200_000, n).astype(np.float64) / 4      # ~11 occurrences per value: many ties for the maximum
    vals[rng.random(n) < 0.01] = np.nan
    valid = rng.random(n) > 0.2
    cols = [capi.Column(ts, None, capi.INT64), capi.Column(vals, np.packbits(valid, bitorder="little"), capi.FLOAT64, 0, n, -1)]
    outs, info = capi.rolling_aggregate(cols, 0, 10 ** 9, [("WindowStart", 0), ("Mode", 1)])
    assert outs[1].length == 2 and outs[1].null_count == 0
    got = outs[1].host_arrays()[0]
    for w, (lo, hi) in enumerate([(0, n1), (n1, n)]):
        v, ok = vals[lo:hi], valid[lo:hi] & ~np.isnan(vals[lo:hi])
        rows = np.flatnonzero(ok)
        uniq, inv, counts = np.unique(v[rows], return_inverse=True, return_counts=True)
        last = np.zeros(len(uniq), dtype=np.int64)
        np.maximum.at(last, inv, rows)
        cand = np.flatnonzero(counts == counts.max())
        want = uniq[cand[np.argmin(last[cand])]]
        assert got[w] == want, (w, got[w], want, counts.max())


def test_planned_call_equals_the_plain_call():
    """bowgpu_rolling_aggregate_planned: the plan the host keeps from the constructor (newIntervalRolling computes it once,
    rolling.go:69-112) gives the same outputs as the call that makes its own plan; a plan made for another column is rejected."""
    rng = np.random.default_rng(11)
    n = 50_000
    ts = np.cumsum(rng.integers(0, 6, n)).astype(np.int64) - 300
    vals = rng.standard_normal(n)
    valid = rng.random(n) > 0.2
    cols = [capi.Column(ts, None, capi.INT64).to_device(),
            capi.Column(vals, np.packbits(valid, bitorder="little"), capi.FLOAT64, 0, n, -1).to_device()]
    aggs = [("WindowStart", 0), ("ArithmeticMean", 1), ("Min", 1), ("Count", 1), ("WeightedAverageLinear", 1)]
    for interval, offset in ((7, 0), (100, -13), (5000, 4999)):
        plain, info0 = capi.rolling_aggregate(cols, 0, interval, aggs, offset=offset, out_residency=capi.DEVICE)
        plan = capi.plan_windows_ex(cols[0], interval, offset)
        assert (plan.s0, plan.num_windows) == (info0.s0, info0.num_windows) and plan.nrows == n
        planned, info1 = capi.rolling_aggregate(cols, 0, interval, aggs, offset=offset, out_residency=capi.DEVICE, plan=plan)
        assert (info1.s0, info1.num_windows, info1.inclusive) == (info0.s0, info0.num_windows, info0.inclusive)
        for a, b in zip(plain, planned):
            assert a.length == b.length and a.null_count == b.null_count and a.type == b.type
            assert np.array_equal(a.host_arrays()[0].view(np.uint64), b.host_arrays()[0].view(np.uint64))
            assert np.array_equal(a.host_arrays()[1], b.host_arrays()[1])
    short = [capi.Column(ts[:100], None, capi.INT64).to_device(), capi.Column(vals[:100], None, capi.FLOAT64).to_device()]
    with pytest.raises(capi.BowGpuError) as e:
        capi.rolling_aggregate(short, 0, 7, aggs, plan=plan, outs=[capi.OutColumn(plan.num_windows, capi.DEVICE) for _ in aggs])
    assert e.value.code == -10
    # a plan made for ANOTHER column of the same length (other first / last timestamp): caught by the pass itself, for every kernel
    # a planned call can take (tile kernels, long-only forms) - BOWGPU_ERR_ARG instead of wrong routes and silently dropped windows
    other = [capi.Column(ts + 1000, None, capi.INT64).to_device(), cols[1]]
    for interval in (7, 1000, 50_000):
        plan = capi.plan_windows_ex(cols[0], interval, 0)
        for label in capi.agg_routes():
            with pytest.raises(capi.BowGpuError) as e:
                capi.rolling_aggregate(other, 0, interval, aggs[:4], plan=plan, outs=[capi.OutColumn(plan.num_windows + 8, capi.DEVICE) for _ in aggs[:4]])
            assert e.value.code == -10 and "plan" in e.value.message, (interval, label)
            good, _ = capi.rolling_aggregate(cols, 0, interval, aggs[:4], plan=plan, out_residency=capi.DEVICE)   # (and the thread goes on working)
            assert good[0].length == plan.num_windows
    # ... and a plan that contradicts itself, on the host
    bad = capi.plan_windows_ex(cols[0], 7, 0)
    bad.num_windows += 1
    with pytest.raises(capi.BowGpuError) as e:
        capi.rolling_aggregate(cols, 0, 7, aggs, plan=bad, outs=[capi.OutColumn(bad.num_windows, capi.DEVICE) for _ in aggs])
    assert e.value.code == -10 and "consistent" in e.value.message


def test_device_output_bitmaps_of_any_alignment_and_length():
    """the caller's device validity buffer holds exactly ceil(W/8) bytes at any address: the fused finish kernel must write those
    bytes and not one more (a guard byte behind the buffer stays intact), clear the padding bits, and count the nulls"""
    rng = np.random.default_rng(5)
    for W_target in (1, 7, 8, 9, 31, 32, 33, 63, 64, 65, 1000, 4097):
        n = W_target * 3
        ts = np.arange(n, dtype=np.int64)
        vals = rng.standard_normal(n)
        valid = rng.random(n) > 0.6
        bm = np.packbits(valid, bitorder="little")
        cols = [capi.Column(ts, None, capi.INT64).to_device(), capi.Column(vals, bm, capi.FLOAT64, 0, n, -1).to_device()]
        aggs = [("WindowStart", 0), ("ArithmeticMean", 1), ("Sum", 1)]
        want, _ = orc.aggregate([orc.Column(ts, None, orc.INT64), orc.Column(vals, bm, orc.FLOAT64)], 0, 3, aggs)
        W = want[0].length
        assert W == W_target
        nb = (W + 7) // 8
        for shift in (0, 1, 2, 3):
            outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
            raw = []
            for o in outs:   # validity inside a bigger buffer: [shift guard bytes][ceil(W/8) bytes][guard bytes]
                big = capi.DeviceBuffer(nb + 16)
                capi.check(capi.lib().bowgpu_memset(capi.C.c_void_p(big.ptr), 0x5A, capi.C.c_int64(nb + 16)))
                raw.append(big)
            oarr = (capi.Out * len(aggs))()
            for i, o in enumerate(outs):
                oarr[i] = o.c()
                oarr[i].validity = raw[i].ptr + shift
            info = capi.AggInfo()
            opts = capi.Options(0, 0, 0)
            capi.check(capi.lib().bowgpu_rolling_aggregate(capi._cols(cols), 2, 0, capi.C.c_int64(3), capi.C.byref(opts), capi._aggs(aggs),
                                                           len(aggs), oarr, capi.C.byref(info)))
            capi.synchronize()
            for i, w in enumerate(want):
                host = raw[i].to_numpy(np.uint8, nb + 16)
                assert (host[:shift] == 0x5A).all() and (host[shift + nb:] == 0x5A).all(), (W, shift, i)
                got_bits = np.unpackbits(host[shift:shift + nb], bitorder="little")
                assert np.array_equal(got_bits[:W].astype(bool), w.valid_mask()), (W, shift, i)
                assert not got_bits[W:].any(), (W, shift, i)     # padding bits clear (bowbuffer.go:25)
                assert oarr[i].null_count == W - int(w.valid_mask().sum()), (W, shift, i)


def test_time_weighted_kernel_timestamp_forms_agree():
    """rolling_tw_kernel stages the interval column either as float64(ts) or - when every |ts| < 2^53, where it is exact - as
    32-bit offsets rebuilt to float64 in the walk (capi.ROUTE_TW_F64 keeps the float64 form): bit-identical outputs, both equal to
    the oracle; timestamps at and beyond 2^53 must take the float64 form by themselves."""
    rng = np.random.default_rng(53)
    aggs = [("WindowStart", 0), ("IntegralStep", 1), ("IntegralTrapezoid", 1), ("WeightedAverageStep", 1), ("WeightedAverageLinear", 1),
            ("ArithmeticMean", 1), ("Min", 1)]
    for shift in (0, -10**6, (1 << 53) - 5000, (1 << 60)):
        n = 30_000
        ts = (np.cumsum(rng.integers(0, 5, n)).astype(np.int64) + shift)
        vals = np.round(rng.standard_normal(n) * 10, 3)
        valid = rng.random(n) > 0.25
        bm = np.packbits(valid, bitorder="little")
        cols = [capi.Column(ts, None, capi.INT64).to_device(), capi.Column(vals, bm, capi.FLOAT64, 0, n, -1).to_device()]
        want, _ = orc.aggregate([orc.Column(ts, None, orc.INT64), orc.Column(vals, bm, orc.FLOAT64)], 0, 7, aggs, offset=2)
        res = []
        for mask in (0, capi.ROUTE_TW_F64):
            with capi.route(mask):
                got, info = capi.rolling_aggregate(cols, 0, 7, aggs, offset=2, out_residency=capi.DEVICE)
            assert capi.last_kernel_name() == "rolling_tw_kernel"
            for (k, _), g, w in zip(aggs, got, want):
                compare("tw forms shift=%d %s" % (shift, k), g, w)
            res.append([g.host_arrays() for g in got])
        for (va, ba), (vb, bb) in zip(*res):
            assert np.array_equal(va.view(np.uint64), vb.view(np.uint64)) and np.array_equal(ba, bb)


def test_pageable_columns_through_the_staging_halves_and_helper_threads():
    """Host (pageable) columns and outputs big enough for several 4 MB staging pieces, each split over the copy helpers, at odd
    sizes and Arrow offsets; the same call with device-resident columns gives the same bits."""
    rng = np.random.default_rng(4 << 20)
    n = 3_000_017
    ts_full = np.cumsum(rng.integers(1, 4, n + 9)).astype(np.int64)
    v_full = rng.standard_normal(n + 9)
    m_full = rng.random(n + 9) > 0.2
    bm_full = np.packbits(m_full, bitorder="little")
    aggs = [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Count", 1)]
    for off in (0, 5):
        host = [capi.Column(ts_full, None, capi.INT64, off, n, 0), capi.Column(v_full, bm_full, capi.FLOAT64, off, n, -1)]
        dev = [capi.Column(ts_full[off:off + n].copy(), None, capi.INT64).to_device(),
               capi.Column(v_full[off:off + n].copy(), np.packbits(m_full[off:off + n], bitorder="little"), capi.FLOAT64, 0, n, -1).to_device()]
        a, _ = capi.rolling_aggregate(host, 0, 3, aggs)                               # ~2e6 windows: 16 MB per output back through the halves
        b, _ = capi.rolling_aggregate(dev, 0, 3, aggs, out_residency=capi.DEVICE)
        want, _ = orc.aggregate([orc.Column(ts_full[off:off + n], None, orc.INT64),
                                 orc.Column(v_full[off:off + n], np.packbits(m_full[off:off + n], bitorder="little"), orc.FLOAT64)], 0, 3, aggs)
        for (k, _), x, y, w in zip(aggs, a, b, want):
            compare("pageable off=%d %s" % (off, k), x, w)
            assert np.array_equal(x.host_arrays()[0].view(np.uint64), y.host_arrays()[0].view(np.uint64))
            assert np.array_equal(x.host_arrays()[1], y.host_arrays()[1])


def test_time_weighted_kernel_window_ids_far_apart_inside_one_tile():
    """Gaps of millions of empty windows between neighbouring rows: the time-weighted kernel's 16-bit head entries carry no window
    id (it is recomputed from the staged 32-bit offset), so ids more than 65535 apart inside one tile stay on that kernel; the
    simple kernel's 32-bit entries hold 16 bits of id and hand such a tile to the wave kernel (status flag + redo)."""
    rng = np.random.default_rng(65536)
    n = 40_000
    step = rng.integers(1, 6, n)
    step[rng.random(n) < 0.0008] = rng.integers(2_000_000, 9_000_000)
    ts = np.cumsum(step).astype(np.int64) + 11
    vals, valid = make_vals(rng, n, "f64", 0.2)
    tw = [("WindowStart", 0)] + [(k, 1) for k in TIME_AGGS] + [("Count", 1)]
    for inclusive in (False, True):
        run_both(ts, [(vals, valid)], 25, tw, offset=2, inclusive=inclusive)
    cols = [capi.Column(ts), capi.Column(vals, np.packbits(valid, bitorder="little"), capi.FLOAT64, 0, n, -1)]
    capi.rolling_aggregate(cols, 0, 25, tw, offset=2)
    assert capi.last_kernel_name() == "rolling_tw_kernel"
    run_both(ts, [(vals, valid)], 25, [(k, 0 if k == "WindowStart" else 1) for k in ALL_AGGS], offset=2)


def test_pinned_zero_copy_residency_equals_the_other_residencies():
    """BOWGPU_HOST_PINNED: registered host buffers are read in place by the kernels (zero-copy) and outputs leave by DMA into
    registered buffers; the same call with pageable / device-resident columns, and with capi.ROUTE_PINNED_STAGE, gives the same bits.
    Arrow offsets and nulls included; an unregistered buffer passed as pinned is an error, not a fault."""
    rng = np.random.default_rng(8)
    n = 300_000
    ts_full = np.cumsum(rng.integers(0, 4, n + 100)).astype(np.int64)
    v_full = rng.standard_normal(n + 100)
    m_full = rng.random(n + 100) > 0.3
    bm_full = np.packbits(m_full, bitorder="little")
    aggs = [("WindowStart", 0), ("ArithmeticMean", 1), ("Min", 1), ("Count", 1), ("WeightedAverageStep", 1)]
    for off in (0, 37):
        def cols():
            return [capi.Column(ts_full, None, capi.INT64, off, n, 0), capi.Column(v_full, bm_full, capi.FLOAT64, off, n, -1)]
        want, _ = orc.aggregate([orc.Column(ts_full[off:off + n], None, orc.INT64),
                                 orc.Column(v_full[off:off + n], np.packbits(m_full[off:off + n], bitorder="little"), orc.FLOAT64)], 0, 10, aggs)
        host, _ = capi.rolling_aggregate(cols(), 0, 10, aggs)
        pinned_cols = [c.pin() for c in cols()]
        try:
            W = want[0].length
            outs = [capi.OutColumn(W, capi.HOST_PINNED) for _ in aggs]
            pinned, _ = capi.rolling_aggregate(pinned_cols, 0, 10, aggs, outs=outs)
            with capi.route(capi.ROUTE_PINNED_STAGE):
                staged, _ = capi.rolling_aggregate(pinned_cols, 0, 10, aggs)
            for (k, _), w, a, b, c_ in zip(aggs, want, host, pinned, staged):
                compare("pinned off=%d %s" % (off, k), b, w)
                for x in (a, c_):
                    assert np.array_equal(x.host_arrays()[0].view(np.uint64), b.host_arrays()[0].view(np.uint64))
                    assert np.array_equal(x.host_arrays()[1], b.host_arrays()[1])
            # Interpolate and a fill through the same residency
            ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
            gi = capi.rolling_interpolate(pinned_cols, 0, 10, ip)
            wi = orc.interpolate([orc.Column(ts_full[off:off + n], None, orc.INT64),
                                  orc.Column(v_full[off:off + n], np.packbits(m_full[off:off + n], bitorder="little"), orc.FLOAT64)], 0, 10, ip)
            from test_gpu_callers import cmp_out
            cmp_out("pinned interpolate ts", gi[0], wi[0])
            cmp_out("pinned interpolate val", gi[1], wi[1])
        finally:
            for c_ in pinned_cols:
                c_.unpin()
    bogus = [capi.Column(ts_full[:1000].copy(), None, capi.INT64), capi.Column(v_full[:1000].copy(), None, capi.FLOAT64)]
    for c_ in bogus:
        c_.residency = capi.HOST_PINNED     # never registered
    with pytest.raises(capi.BowGpuError) as e:
        capi.rolling_aggregate(bogus, 0, 10, aggs[:2])
    assert e.value.code == -10


@pytest.mark.parametrize("vkind,null_frac", [("f64", 0.0), ("f64", 0.3), ("f64", 0.95), ("i64", 0.0), ("i64", 0.4)])
def test_extrema_of_tiles_with_few_windows_by_all_lanes(vkind, null_frac):
    """Round 6: in a tile of few windows (rolling_simple.hip kCoopMaxHeads: windows of ~92 rows and more, ~128 next to sums) the extrema are found by all 64
    lanes - 16 per window over contiguous pieces - instead of by one lane per window.  minmax.go:16-28's two order-dependent rules must
    survive: a NaN FIRST value stays whatever follows; among values that compare equal (-0.0 / +0.0) the earlier row stays.  Windows of
    40 .. 600 rows (below / above the switch, longer than the look-ahead: queued), regular and irregular, columns full of NaN / +-0 /
    +-Inf and ties, nulls up to 95 %, Min / Max alone and next to sums and First / Last - every output bit for bit, every route."""
    rng = np.random.default_rng(int(null_frac * 100) + (7 if vkind == "f64" else 13))
    n = 120_000
    for mode, interval in [("dense", 40), ("dense", 53), ("dense", 64), ("dense", 100), ("dense", 128), ("dense", 160), ("dense", 250), ("dense", 600),
                           ("irregular", 600), ("irregular", 1500), ("gappy", 300), ("dups", 90)]:
        ts = make_ts(rng, n, mode)
        if vkind == "f64":
            v = rng.standard_normal(n) * 10.0 ** rng.integers(-2, 3, n)
            sp = rng.random(n)
            v[sp < 0.08] = np.nan
            v[(sp >= 0.08) & (sp < 0.16)] = 0.0
            v[(sp >= 0.16) & (sp < 0.24)] = -0.0
            v[(sp >= 0.24) & (sp < 0.27)] = np.inf
            v[(sp >= 0.27) & (sp < 0.30)] = -np.inf
            v[(sp >= 0.30) & (sp < 0.45)] = np.round(v[(sp >= 0.30) & (sp < 0.45)])     # ties
            if interval >= 100:      # whole windows of zeros of both signs, of NaN, of one infinity
                v[5_000:5_400] = np.where(rng.random(400) < 0.5, 0.0, -0.0)
                v[9_000:9_400] = np.nan
                v[12_000:12_400] = np.inf
        else:
            v = rng.integers(-5, 6, n).astype(np.int64)
        valid = None if null_frac == 0 else rng.random(n) >= null_frac
        for aggs in ([("WindowStart", 0), ("Min", 1), ("Max", 1)],
                     [("WindowStart", 0), ("Sum", 1), ("Min", 1), ("Max", 1), ("Count", 1)],
                     [("Max", 1), ("First", 1), ("WindowStart", 0), ("Last", 1), ("Min", 1, [-2.0]), ("ArithmeticMean", 1)]):
            outs, exp, info = run_both(ts, [(v, valid)], interval, aggs, offset=int(rng.integers(0, interval)))
            for (k, *_), g, w in zip(aggs, outs, exp):
                exact = info.long_windows == 0 or k not in ORDER_SENSITIVE
                if exact:
                    compare("few windows per tile %s I=%d %s nulls=%.2f %s" % (mode, interval, vkind, null_frac, k), g, w)


def test_strict_long_windows_find_an_unsorted_interval_column_themselves():
    """Round 6: under strict_order a call of long windows with a time-weighted reducer has no pass of its own over the interval column any
    more - the lane walks check the order of the rows they read (long_windows.hip walk_entry check_order) and that the windows' row
    ranges tile the frame.  One descent anywhere - inside a window, between two windows, at the frame's ends, in front of a run of empty
    windows - must still be BOWGPU_ERR_TS_UNSORTED; sets that read no timestamps keep ts_sorted_kernel."""
    rng = np.random.default_rng(5)
    n = 300_000
    base = np.cumsum(rng.integers(1, 4, n)).astype(np.int64)
    base[200_000:] += 50_000                       # a run of empty windows
    v = rng.standard_normal(n)
    tw = [("WindowStart", 0), ("WeightedAverageStep", 1), ("ArithmeticMean", 1)]
    plain = [("WindowStart", 0), ("ArithmeticMean", 1)]
    for aggs in (tw, plain):
        outs, info = capi.rolling_aggregate([capi.Column(base, None, capi.INT64), capi.Column(v, None, capi.FLOAT64)], 0, 1500, aggs, strict_order=True)
        assert capi.last_kernel_name() == "long_strict_kernel" and info.long_windows == 0
        exp, _ = orc.aggregate([orc.Column(base, None, orc.INT64), orc.Column(v, None, orc.FLOAT64)], 0, 1500, aggs)
        for a, g, w in zip(aggs, outs, exp):
            compare("strict sorted " + a[0], g, w)
        for spot in (1, 2, 7, 8, 9, 700, 751, 199_999, 200_000, 200_001, n - 2, n - 1, int(rng.integers(1, n))):
            for drop in (1, 10_000_000):
                if spot == n - 1 and drop > 1:
                    continue      # (a last timestamp below the first window start: countWindows gives no window at all, rolling.go:150-152 - nothing runs, nothing to find)
                ts = base.copy()
                ts[spot] = ts[spot - 1] - drop
                cols = [capi.Column(ts, None, capi.INT64), capi.Column(v, None, capi.FLOAT64)]
                with pytest.raises(capi.BowGpuError) as e:
                    capi.rolling_aggregate(cols, 0, 1500, aggs, strict_order=True)
                assert e.value.code in (-14, -9), (aggs[1][0], spot, drop, e.value)     # (-9: the bisection on unsorted rows made a window of > 2^20 rows - declined either way)
                if drop == 1:
                    assert e.value.code == -14, (aggs[1][0], spot, e.value)


def test_strict_order_and_the_pinned_form_thresholds():
    """bowgpu_options.strict_order: every window in the reference's row order (bit-exact, long_windows == 0) or the call is declined.
    And the window length at which a call changes form (common.h: kLongOnlyAvgRows = 128 rows on average for the calls with both kinds
    of integral, kLongStreamAnyAvgRows = 129 for the rest) - a change of it is a change of which calls are bit-exact."""
    rng = np.random.default_rng(77)
    n = 128 * 2400          # (whole windows of 128 and of 256 rows: the averages are the window lengths)
    ts = np.arange(n, dtype=np.int64)
    f = rng.standard_normal(n)
    cols = [capi.Column(ts, None, capi.INT64), capi.Column(f, None, capi.FLOAT64)]
    ocols = [orc.Column(ts, None, orc.INT64), orc.Column(f, None, orc.FLOAT64)]
    lite = [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1)]
    more = [("WindowStart", 0), ("Sum", 1), ("Min", 1), ("WeightedAverageStep", 1)]
    # which form runs where
    both = [("WindowStart", 0), ("IntegralStep", 1), ("WeightedAverageLinear", 1)]
    # (round 6: the windows a tile pass queues are walked behind it by long_queue_kernel - in row order, so long_windows stays 0 -, which
    # keeps columns WITHOUT nulls on the tile kernels up to 176 rows per window for sets with extrema, First / Last or one kind of integral,
    # up to 200 for extrema alone: api.cpp job_run tile_band_rows)
    mm_only = [("WindowStart", 0), ("Min", 1), ("Max", 1)]
    for aggs, interval, kernel in ((lite, 127, "rolling_simple_kernel"), (lite, 128, "rolling_simple_kernel"), (lite, 130, "long_stream_kernel"), (more, 127, None),
                                   (more, 128, "rolling_tw_kernel"), (more, 130, "rolling_tw_kernel"), (more, 176, "rolling_tw_kernel"), (more, 180, "long_stream_kernel"),
                                   (more, 256, "long_stream_kernel"), (mm_only, 200, "rolling_simple_kernel"), (mm_only, 208, "long_stream_kernel"),
                                   (more, 1000, "long_stream_kernel"), (both, 127, "rolling_tw_kernel"), (both, 128, "long_stream_kernel")):
        outs, info = capi.rolling_aggregate(cols, 0, interval, aggs)
        name = capi.last_kernel_name()
        if kernel is None:
            assert not name.startswith("long_stream"), (interval, name)
        else:
            assert name == kernel, (interval, name)
        assert (info.long_windows == info.num_windows) == name.startswith("long_"), (interval, name, info.long_windows)
    # ... with a nullable column the valid points are compacted first (rolling_twc.hip), from 12 rows per window on and up to 128 for both
    # kinds of integral too
    fv = rng.random(n) > 0.3
    ncols = [capi.Column(ts, None, capi.INT64), capi.Column(f, np.packbits(fv, bitorder="little"), capi.FLOAT64, 0, n, -1)]
    summ = [("WindowStart", 0), ("Sum", 1), ("Min", 1), ("Max", 1)]
    for aggs, interval, kernel in ((more, 10, "rolling_tw_kernel"), (more, 12, "rolling_twc_kernel"), (more, 128, "rolling_twc_kernel"), (more, 130, "rolling_twc_kernel"),
                                   (more, 170, "rolling_twc_kernel"), (more, 180, "long_stream_kernel"), ([("WindowStart", 0), ("WeightedAverageStep", 1)], 250, "rolling_twc_kernel"),
                                   ([("WindowStart", 0), ("WeightedAverageStep", 1)], 260, "long_stream_kernel"), (both, 128, "rolling_twc_kernel"), (both, 170, "rolling_twc_kernel"),
                                   (both, 180, "long_stream_kernel"), (lite, 64, "rolling_simple_kernel"), (summ, 40, "rolling_simple_kernel"), (summ, 50, "rolling_twc_kernel"),
                                   (summ, 128, "rolling_twc_kernel"), (summ, 170, "rolling_twc_kernel"), (summ, 180, "long_stream_kernel"),
                                   ([("WindowStart", 0), ("First", 1), ("Last", 1)], 250, "rolling_twc_kernel"), (lite, 130, "long_stream_kernel"),
                                   # round 6: extrema / First + Last ALONE on a nullable column stay on rolling_simple.hip (+ the queue launch) through the band
                                   (mm_only, 150, "rolling_simple_kernel"), (mm_only, 200, "rolling_simple_kernel"), (mm_only, 208, "long_stream_kernel"),
                                   ([("WindowStart", 0), ("First", 1), ("Last", 1)], 150, "rolling_simple_kernel"),
                                   ([("WindowStart", 0), ("First", 1), ("Last", 1)], 180, "rolling_twc_kernel"),
                                   ([("WindowStart", 0), ("Min", 1), ("Last", 1)], 170, "rolling_simple_kernel")):
        capi.rolling_aggregate(ncols, 0, interval, aggs)
        assert capi.last_kernel_name() == kernel, (interval, capi.last_kernel_name())
    with capi.route(capi.ROUTE_TW_ROWS):
        capi.rolling_aggregate(ncols, 0, 64, more)
        assert capi.last_kernel_name() == "rolling_tw_kernel"
    # strict order: short windows are exact either way ...
    for interval in (10, 100):
        exp, _ = orc.aggregate(ocols, 0, interval, more)
        outs, info = capi.rolling_aggregate(cols, 0, interval, more, strict_order=True)
        assert info.long_windows == 0
        for k, g, w in zip(_names(more), outs, exp):
            compare("strict %s I=%d" % (k, interval), g, w, exact=True)
    # ... and windows no tile can hold are walked in row order by one lane each (round 4: long_strict_kernel; round 3 declined them):
    # bit for bit, whatever the reducer set, planned or not, per call or per thread, inclusive or not
    every = [(k, 0 if k == "WindowStart" else 1) for k in ALL_AGGS]
    for aggs in (lite, more, every):
        for interval in (130, 1000, 100_000):
            exp, _ = orc.aggregate(ocols, 0, interval, aggs)
            outs, info = capi.rolling_aggregate(cols, 0, interval, aggs, strict_order=True)
            assert info.long_windows == 0 and capi.last_kernel_name() == "long_strict_kernel", (interval, capi.last_kernel_name())
            for k, g, w in zip(_names(aggs), outs, exp):
                compare("strict %s I=%d" % (k, interval), g, w, exact=True)
        # (128-row windows: a tile still holds them - in row order)
        exp, _ = orc.aggregate(ocols, 0, 128, aggs)
        outs, info = capi.rolling_aggregate(cols, 0, 128, aggs, strict_order=True)
        assert info.long_windows == 0 and not capi.last_kernel_name().startswith("long_")
        for k, g, w in zip(_names(aggs), outs, exp):
            compare("strict %s I=128" % k, g, w, exact=True)
    plan = capi.plan_windows_ex(cols[0], 1000, 0)
    exp, _ = orc.aggregate(ocols, 0, 1000, lite)
    outs, info = capi.rolling_aggregate(cols, 0, 1000, lite, plan=plan, strict_order=True)
    with capi.route(capi.ROUTE_STRICT_ORDER):
        outs2, info2 = capi.rolling_aggregate(cols, 0, 1000, lite)
    for k, g, g2, w in zip(_names(lite), outs, outs2, exp):
        compare("strict planned %s" % k, g, w, exact=True)
        compare("strict by route %s" % k, g2, w, exact=True)
    assert info.long_windows == 0 and info2.long_windows == 0
    outs, info = capi.rolling_aggregate(cols, 0, 1000, lite)       # (the flag does not stick to the thread)
    assert info.long_windows == info.num_windows


def test_strict_order_mixed_lengths_nulls_and_the_stated_limit():
    """strict_order on irregular data: mostly short windows with a few that outgrow a tile (the tile kernel queues them, one lane each
    walks them in row order), nullable and Int64 columns, inclusive windows, rows below s0; and the stated limit - a window of more
    than 2^20 rows declines the call (BOWGPU_ERR_UNSUPPORTED), it is never reduced as a tree behind the caller's back."""
    rng = np.random.default_rng(2024)
    n = 400_000
    step = rng.integers(1, 5, n)
    step[rng.random(n) < 0.002] = 4000            # a gap now and then: empty windows behind a window
    ts = np.cumsum(step).astype(np.int64) - 3000  # (negative timestamps: rows below s0 ride in window 0)
    burst = rng.random(n) < 0.0005
    for i in np.flatnonzero(burst)[:40]:          # bursts of rows with one timestamp: windows of hundreds / thousands of rows
        ts[i:i + int(rng.integers(200, 3000))] = ts[i]
    ts = np.sort(ts)
    f, fm = make_vals(rng, n, "f64", 0.3)
    g, gm = make_vals(rng, n, "i64", 0.1)
    aggs = [(k, 0 if k == "WindowStart" else 1) for k in ALL_AGGS + TIME_AGGS] + [("Sum", 2), ("ArithmeticMean", 2), ("First", 2), ("WeightedAverageLinear", 2)]
    for interval, offset in ((50, 0), (300, 7)):
        for inclusive in (False, True):
            cols = [capi.Column(ts), capi.Column(f, np.packbits(fm, bitorder="little"), capi.FLOAT64, 0, n, -1),
                    capi.Column(g, np.packbits(gm, bitorder="little"), capi.INT64, 0, n, -1)]
            ocols = [orc.Column(ts, None, orc.INT64), orc.Column(f, np.packbits(fm, bitorder="little"), orc.FLOAT64),
                     orc.Column(g, np.packbits(gm, bitorder="little"), orc.INT64)]
            exp, _ = orc.aggregate(ocols, 0, interval, aggs, offset=offset, inclusive=inclusive)
            outs, info = capi.rolling_aggregate(cols, 0, interval, aggs, offset=offset, inclusive=inclusive, strict_order=True)
            assert info.long_windows == 0
            for k, got, want in zip(_names(aggs), outs, exp):
                compare("strict mixed %s I=%d incl=%s" % (k, interval, inclusive), got, want, exact=True)
    # the limit
    m = (1 << 20) + 5000
    bigv = rng.standard_normal(m)
    big = [capi.Column(np.arange(m, dtype=np.int64)), capi.Column(bigv, None, capi.FLOAT64)]
    with pytest.raises(capi.BowGpuError) as e:
        capi.rolling_aggregate(big, 0, 1 << 21, [("WindowStart", 0), ("Sum", 1)], strict_order=True)
    assert e.value.code == -9 and "2^20" in e.value.message
    outs, info = capi.rolling_aggregate(big, 0, 1 << 19, [("WindowStart", 0), ("Sum", 1)], strict_order=True)   # (half a million rows per window: served)
    exp, _ = orc.aggregate([orc.Column(np.arange(m, dtype=np.int64), None, orc.INT64), orc.Column(bigv, None, orc.FLOAT64)], 0, 1 << 19,
                           [("WindowStart", 0), ("Sum", 1)])
    compare("strict 2^19-row windows Sum", outs[1], exp[1], exact=True)


@pytest.mark.parametrize("data", ["regular", "irregular", "ns"])
def test_time_weighted_kernel_forms_by_shape(data):
    """rolling_tw_kernel's instantiations, each reached on purpose (round 4; DESIGN.md "The walks"): the lean form (one column without
    nulls, integrals only), the one-walk form of short windows, the term phases (nulls: previous points gathered; values next to
    integrals; both kinds of integral with the second kind waiting in registers), several columns incl. an Int64 one - at window lengths
    either side of every threshold (14 / 20 / 44 / 64 rows per window, 48 heads per tile), exclusive and inclusive, 32-bit and 64-bit
    times.  Every window fits a tile here, so every reducer is bit-exact."""
    rng = np.random.default_rng({"regular": 1, "irregular": 2, "ns": 3}[data])
    n = 60_000
    if data == "regular":
        ts = np.arange(n, dtype=np.int64) * 3 + 17
        unit = 3
    elif data == "irregular":
        ts = np.cumsum(rng.integers(1, 6, n)).astype(np.int64) - 500
        unit = 3
    else:   # nanosecond epochs at ~10 Hz: beyond 2^53 and far wider than 2^32 (64-bit times, per-tile window ids)
        ts = (1_700_000_000_000_000_000 + np.cumsum(rng.integers(90_000_000, 110_000_000, n))).astype(np.int64)
        unit = 100_000_000
    f, fm = make_vals(rng, n, "f64", 0.3)
    g, gm = make_vals(rng, n, "i64", 0.2)
    dense = (f, None)
    one_kind = [("WindowStart", 0), ("WeightedAverageStep", 1)]
    one_trap = [("WindowStart", 0), ("WeightedAverageLinear", 1), ("Count", 1)]
    both = [("WindowStart", 0), ("IntegralStep", 1), ("IntegralTrapezoid", 1), ("WeightedAverageStep", 1), ("WeightedAverageLinear", 1)]
    mixed = [("WindowStart", 0), ("IntegralStep", 1), ("ArithmeticMean", 1), ("Min", 1), ("Last", 1)]
    mixed_both = [("WindowStart", 0), ("IntegralTrapezoid", 1), ("WeightedAverageStep", 1), ("Sum", 1), ("Max", 1), ("First", 1)]
    multi = [("WindowStart", 0), ("WeightedAverageStep", 1), ("IntegralTrapezoid", 2), ("Sum", 2), ("WeightedAverageLinear", 3), ("First", 3)]
    for rows_per_window in (5, 13, 16, 19, 22, 40, 48, 60, 70, 110):
        interval = rows_per_window * unit
        for inclusive in (False, True):
            for cols, sets in (([dense], (one_kind, one_trap, both, mixed, mixed_both)),          # lean / one-walk / phases without nulls
                               ([(f, fm)], (one_kind, both, mixed, mixed_both)),                      # gathers
                               ([dense, (f, fm), (g, gm)], (multi,))):                                # several columns, an Int64 one
                for aggs in sets:
                    outs, exp, info = run_both(ts, cols, interval, aggs, offset=int(rng.integers(0, interval)), inclusive=inclusive)
                    assert info.long_windows == 0
