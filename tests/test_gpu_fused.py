"""GPU parity for r.Interpolate(interps...).Aggregate(aggs...) made as ONE call (bowgpu_rolling_interpolate_aggregate; reference
rolling/interpolation.go:30-69 + rolling/aggregation.go:123-145): the fused kernel (rolling_fused.hip: the interpolated frame is never
written), the same entry point pushed through its two-call form (capi.ROUTE_NO_FUSED), and the two public calls made one after the
other - all three bit for bit equal to oracle interpolate -> oracle aggregate on the same inputs."""
import numpy as np
import pytest

from bow_amd import capi
from oracle import pyoracle as orc
from test_gpu_aggregate import compare, make_ts

pytestmark = pytest.mark.gpu

SIMPLE_AGGS = ["WindowStart", "Sum", "ArithmeticMean", "Min", "Max", "Count", "First", "Last", "NumRows"]


def _cols(ts, cols_np):
    ccols, ocols = [capi.Column(ts, None, capi.INT64)], [orc.Column(ts, None, orc.INT64)]
    for vals, valid in cols_np:
        bm = None if valid is None else np.packbits(valid, bitorder="little")
        typ = capi.INT64 if vals.dtype == np.int64 else capi.FLOAT64
        ccols.append(capi.Column(vals, bm, typ, 0, len(vals), -1))
        ocols.append(orc.Column(vals, bm, typ))
    return ccols, ocols


def run_fused(ts, cols_np, interval, interps, aggs, offset=0, inclusive=False, device=False, expect=None, strict_order=False):
    """expect: "fused" / "two-call" - which form the plain call must have taken (None: either).  strict_order: for frames with a window
    longer than a tile holds - the two-call form then walks it in row order too (bowgpu_options.strict_order), so every comparison stays
    bit for bit (without it such a window's float sums carry the stated order-free tolerance: bowgpu_agg_info.long_windows)"""
    ccols, ocols = _cols(ts, cols_np)
    if device:
        ccols = [c.to_device() for c in ccols]
    mid = orc.interpolate(ocols, 0, interval, interps, offset=offset, inclusive=inclusive)
    want, nic = orc.aggregate(mid, 0, interval, aggs, offset=offset, inclusive=inclusive)
    res = capi.DEVICE if device else capi.HOST
    label = "n=%d I=%d off=%d %s" % (len(ts), interval, offset, [a[0] for a in aggs])
    # 1. the one call
    capi.rolling_aggregate(ccols[:1], 0, interval, [("WindowStart", 0)], offset=offset)   # (something else as the thread's last kernel)
    got, info = capi.rolling_interpolate_aggregate(ccols, 0, interval, interps, aggs, offset=offset, inclusive=inclusive, out_residency=res,
                                                   strict_order=strict_order)
    took = "fused" if capi.last_kernel_name() == "rolling_fused_kernel" else "two-call"
    if expect is not None:
        assert took == expect, (label, took)
    assert info.new_interval_col == nic and info.num_windows == want[0].length and info.long_windows == 0
    for (k, _c, *_f), g, w in zip(aggs, got, want):
        compare("one call (%s) %s %s" % (took, k, label), g, w)
    # 2. the same entry point, two calls through device temporaries
    with capi.route(capi.ROUTE_NO_FUSED):
        got2, _ = capi.rolling_interpolate_aggregate(ccols, 0, interval, interps, aggs, offset=offset, inclusive=inclusive, out_residency=res,
                                                     strict_order=strict_order)
        assert capi.last_kernel_name() != "rolling_fused_kernel"
    for (k, _c, *_f), g, w in zip(aggs, got2, want):
        compare("two calls behind the entry point %s %s" % (k, label), g, w)
    # 3. the two public calls
    filled = capi.rolling_interpolate(ccols, 0, interval, interps, offset=offset, inclusive=inclusive, out_residency=capi.DEVICE)
    fcols = [capi.Column(f.values, f.validity, f.type, 0, f.length, -1) for f in filled]   # (device buffers of the first call's outputs)
    if fcols[0].length > 0:
        got3, _ = capi.rolling_aggregate(fcols, 0, interval, aggs, offset=offset, inclusive=inclusive, out_residency=res, strict_order=strict_order)
        for (k, _c, *_f), g, w in zip(aggs, got3, want):
            compare("Interpolate then Aggregate %s %s" % (k, label), g, w)
    return got, want, took


def _vals(rng, n, kind, null_frac):
    if kind == "f64":
        v = np.round(rng.standard_normal(n) * 100, 2)
    else:
        v = rng.integers(-1000, 1000, n).astype(np.int64)
    valid = None if null_frac == 0 else rng.random(n) >= null_frac
    return v, valid


@pytest.mark.parametrize("kind", ["Linear", "StepPrevious", "None"])
@pytest.mark.parametrize("vkind,null_frac", [("f64", 0.0), ("f64", 0.3), ("i64", 0.3), ("f64", 0.9)])
def test_fused_equals_interpolate_then_aggregate(kind, vkind, null_frac):
    rng = np.random.default_rng(hash((kind, vkind, null_frac)) % (1 << 31))
    for n, interval, offset in [(1, 5, 0), (40, 30, 1), (700, 50, 3), (5000, 100, 0), (60_000, 100, 7), (60_000, 64, 0), (200_000, 1000, 13)]:
        ts = np.cumsum(rng.integers(0, 20, n)).astype(np.int64) + 500
        v, valid = _vals(rng, n, vkind, null_frac)
        ip = [{"kind": "WindowStart", "col": 0}, {"kind": kind, "col": 1}]
        for aggs in ([("WindowStart", 0), ("ArithmeticMean", 1)],
                     [("WindowStart", 0)] + [(k, 1) for k in SIMPLE_AGGS[1:]],
                     [("Min", 1), ("Max", 1), ("WindowStart", 0), ("NumRows", 0)],
                     [("First", 1), ("Last", 1), ("Count", 1), ("WindowStart", 0)]):
            run_fused(ts, [(v, valid)], interval, ip, aggs, offset=offset, expect="fused" if n >= 5000 else None)


def test_fused_prev_row_multi_columns_factors_and_the_interval_column_as_a_value():
    rng = np.random.default_rng(77)
    n = 80_000
    ts = np.cumsum(rng.integers(1, 12, n)).astype(np.int64)
    a, va = _vals(rng, n, "f64", 0.4)
    b, vb = _vals(rng, n, "i64", 0.2)
    c, _ = _vals(rng, n, "f64", 0.0)
    va[:300] = False           # no previous point for the first windows: Options.PrevRow serves (linear.go:14-25, stepprevious.go:13-15)
    vb[:450] = False
    prev = (float(ts[0] - 3), True, 42.5, True, 42)
    ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1, "prev": prev}, {"kind": "StepPrevious", "col": 2, "prev": prev},
          {"kind": "Linear", "col": 3}]
    aggs = [("WindowStart", 0), ("ArithmeticMean", 1), ("First", 2, [0.5]), ("Sum", 3, [2.0, -1.0]), ("Max", 2), ("Min", 1), ("Count", 0),
            ("ArithmeticMean", 0), ("Last", 2), ("NumRows", 3), ("Count", 3)]
    for interval, offset in ((60, 0), (100, 7), (333, -5)):
        run_fused(ts, [(a, va), (b, vb), (c, None)], interval, ip, aggs, offset=offset, expect="fused")
        run_fused(ts, [(a, va), (b, vb), (c, None)], interval, ip, aggs, offset=offset, device=True, expect="fused")


def test_fused_empty_window_runs_exact_heads_and_duplicates():
    """runs of empty windows (each holds ONE row in the interpolated frame: its synthetic row), rows sitting exactly on a window
    start (no synthetic row), duplicated timestamps on a start (only the first is the head)"""
    rng = np.random.default_rng(5)
    n = 120_000
    step = rng.integers(0, 6, n)
    step[rng.random(n) < 0.002] = 3000          # gaps of ~30 empty windows
    step[n // 2] = 2_000_000                    # one run of 20 000 empty windows
    ts = np.cumsum(step).astype(np.int64) + 1000
    on = rng.random(n) < 0.2
    ts[on] -= ts[on] % 100                      # many rows exactly on a window start, with duplicates
    ts = np.sort(ts)
    for vkind, nf in (("f64", 0.3), ("i64", 0.0), ("f64", 0.0)):
        v, valid = _vals(rng, n, vkind, nf)
        for kind in ("Linear", "StepPrevious", "None"):
            ip = [{"kind": "WindowStart", "col": 0}, {"kind": kind, "col": 1}]
            aggs = [("WindowStart", 0)] + [(k, 1) for k in SIMPLE_AGGS[1:]]
            got, want, took = run_fused(ts, [(v, valid)], 100, ip, aggs)
            assert want[0].length > 20_000


def test_fused_declines_what_it_cannot_describe_and_the_answer_stays_the_same():
    rng = np.random.default_rng(12)
    n = 100_000
    ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
    aggs = [("WindowStart", 0), ("ArithmeticMean", 1), ("Min", 1), ("Count", 1)]
    # a run of 5000 nulls: a neighbour point further away than the bounded search looks
    ts = np.arange(n, dtype=np.int64) * 3 + 1     # (no row sits on a window start: every window gets a synthetic row)
    v, valid = _vals(rng, n, "f64", 0.2)
    valid[40_000:45_000] = False
    run_fused(ts, [(v, valid)], 60, ip, aggs, expect="two-call")
    valid[40_000:45_000] = True
    run_fused(ts, [(v, valid)], 60, ip, aggs, expect="fused")
    # one window of 700 rows among windows of 20 (longer than any tile holds; a window of up to 128 rows always fits one)
    ts2 = ts.copy()
    ts2[50_000:50_700] = ts2[50_000]
    run_fused(np.sort(ts2), [(v, valid)], 60, ip, aggs, expect="two-call", strict_order=True)
    # a stretch of windows of one row each: more heads than a tile's list holds
    ts3 = ts.copy()
    ts3[60_000:] += np.arange(n - 60_000) * 1000
    run_fused(ts3[:61_000], [(v[:61_000], valid[:61_000])], 60, ip, aggs)
    # shapes the host declines up front: time-weighted reducers, inclusive windows, negative timestamps, a constant interpolator
    run_fused(ts, [(v, valid)], 60, ip, [("WindowStart", 0), ("WeightedAverageStep", 1)], expect="two-call")
    run_fused(ts, [(v, valid)], 60, ip, [("WindowStart", 0), ("IntegralTrapezoid", 1), ("ArithmeticMean", 1)], expect="two-call")
    run_fused(ts, [(v, valid)], 60, ip, aggs, inclusive=True, expect="two-call")
    # negative timestamps: fine as such (the first window start lies below the first row) ...
    run_fused(ts - 777, [(v, valid)], 60, ip, aggs, expect="fused")
    # ... but not with a window that starts at -1, the reference's "no first value" sentinel (interpolation.go:99-105: that window never
    # gets a synthetic row)
    run_fused(ts - 62, [(v, valid)], 60, ip, aggs, offset=59, expect="two-call")     # (windows start at -61, -1, 59, ...)
    # rows below the first window start (Go's truncating division: rolling.go:96-99): the synthetic row of window 0 lands IN FRONT of
    # them (interpolation.go:160), so the interpolated interval column is not ascending - the Aggregate step declines it, in one call as
    # in two (the caller keeps the reference's own path for such a frame)
    ccols, _o = _cols(ts - 51, [(v, valid)])
    for mask in (0, capi.ROUTE_NO_FUSED):
        with capi.route(mask), pytest.raises(capi.BowGpuError) as e:
            capi.rolling_interpolate_aggregate(ccols, 0, 60, ip, aggs, offset=50)
        assert e.value.code == -14
    filled = capi.rolling_interpolate(ccols, 0, 60, ip, offset=50)
    assert filled[0].to_list()[:2] == [-10, -50]
    with pytest.raises(capi.BowGpuError) as e:
        capi.rolling_aggregate([capi.Column(f.host_arrays()[0], f.host_arrays()[1], f.type, 0, f.length, -1) for f in filled], 0, 60, aggs, offset=50)
    assert e.value.code == -14
    run_fused(ts, [(v, valid)], 60, [{"kind": "WindowStart", "col": 0}, {"kind": "Const", "col": 1, "const": 9.9}], aggs, expect="two-call")
    # nanosecond epochs: wider than 2^32 from the first window
    run_fused(ts * 1_000_000 + 1_700_000_000_000_000_000, [(v, valid)], 60_000_000, ip, aggs, expect="two-call")


def test_fused_nan_zero_and_signalling_nan_semantics_with_a_synthetic_seed():
    """the synthetic row is the window's FIRST row: it seeds Min / Max (a NaN seed stays, minmax.go:16-28), 0.0 + (-0.0) = +0.0 is the
    Sum of a window whose only row is a synthetic -0.0, a signalling NaN behind the synthetic seed does not lose the running extremum"""
    n = 30_000
    ts = np.arange(n, dtype=np.int64) * 7 + 3       # never on a window start of interval 50 with offset 0 ... mostly
    rng = np.random.default_rng(3)
    v = rng.normal(size=n)
    v[rng.random(n) < 0.05] = np.nan
    v[rng.random(n) < 0.05] = -0.0
    v[rng.random(n) < 0.05] = 0.0
    v[rng.random(n) < 0.01] = np.inf
    snan = np.array([0x7FF0000000000001], dtype=np.uint64).view(np.float64)[0]
    v[np.arange(11, n, 97)] = snan
    valid = rng.random(n) > 0.25
    aggs = [("WindowStart", 0)] + [(k, 1) for k in SIMPLE_AGGS[1:]]
    for kind in ("StepPrevious", "Linear"):
        ip = [{"kind": "WindowStart", "col": 0}, {"kind": kind, "col": 1}]
        for cols in ([(v, valid)], [(v, None)]):
            run_fused(ts, cols, 50, ip, aggs, expect="fused")
            run_fused(ts, cols, 350, ip, [("WindowStart", 0), ("Min", 1), ("Max", 1)], expect="fused")


def test_fused_errors_come_in_the_references_order():
    ts = capi.Column.from_list([10, 13, 17, 30], "int64")
    val = capi.Column.from_list([1.0, None, 3.0, 4.0], "float64")
    ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
    with pytest.raises(capi.BowGpuError) as e:      # newIntervalRolling first (rolling.go:115-117)
        capi.rolling_interpolate_aggregate([ts, val], 0, 0, ip, [("WindowStart", 0)])
    assert e.value.code == -1
    with pytest.raises(capi.BowGpuError) as e:      # then Interpolate's validation (windowstart.go:9: Int64 only)
        capi.rolling_interpolate_aggregate([ts, val], 0, 5, [{"kind": "WindowStart", "col": 0}, {"kind": "WindowStart", "col": 1}], [("Sum", 1)])
    assert e.value.code == -7 and e.value.message == "accepts types [int64], got type float64"
    with pytest.raises(capi.BowGpuError) as e:      # then Aggregate's (aggregation.go:163-166)
        capi.rolling_interpolate_aggregate([ts, val], 0, 5, ip, [("Sum", 1)])
    assert e.value.code == -5
    unsorted = capi.Column(np.array([10, 13, 12, 30] * 2000, dtype=np.int64))
    v = capi.Column(np.arange(8000, dtype=np.float64), None, capi.FLOAT64)
    with pytest.raises(capi.BowGpuError) as e:
        capi.rolling_interpolate_aggregate([unsorted, v], 0, 5, ip, [("WindowStart", 0), ("Sum", 1)])
    assert e.value.code == -14
    # the golden Interpolate vectors of the reference (linear_test.go:26-126), aggregated
    for tsl, vl, interval, offset in (([10, 15, 17], [10.0, 15.0, 17.0], 2, 0), ([10, 15, 17], [30.0, 25.0, 24.0], 2, 3)):
        tsn, vn = np.array(tsl, dtype=np.int64), np.array(vl)
        run_fused(tsn, [(vn, None)], interval, ip, [("WindowStart", 0), ("Sum", 1), ("Count", 1), ("First", 1)], offset=offset)
