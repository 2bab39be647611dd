"""Differential fuzzing of the HIP path against the oracle: seeded random shapes of everything the C ABI accepts - row
counts around the tile sizes, timestamp patterns (dense, irregular, duplicates, gaps, negative / rows below s0), intervals
and raw offsets, Arrow offsets into longer buffers, null densities from 0 to 100 %, NaN / +-0 / Inf values, mixed column
types, random reducer lists with Factor chains, inclusive windows, host / device residency - each case through every tile
kernel that covers it (BOW_FUZZ_SEEDS=N runs N seeds instead of 64).  Bit-exact except Sum / Mean / integrals of windows on an
order-free path, which must lie within the stated bound (tests/tolerance.py)."""
import os

import numpy as np
import pytest

from bow_amd import capi
from oracle import pyoracle as orc
from test_gpu_aggregate import ALL_AGGS, TIME_AGGS, ORDER_SENSITIVE, compare
from tolerance import assert_within, order_free_bounds
from test_gpu_callers import both_interp_kernels
from test_gpu_callers import cmp_out

pytestmark = pytest.mark.gpu


def rand_ts(rng, n):
    mode = rng.integers(0, 7)
    if n == 0:
        return np.zeros(0, np.int64)
    if mode == 0:
        ts = np.arange(n, dtype=np.int64)
    elif mode == 1:
        ts = np.cumsum(rng.integers(1, 20, n))
    elif mode == 2:
        ts = np.cumsum(rng.integers(0, 3, n))
    elif mode == 3:
        step = rng.integers(1, 5, n)
        step[rng.random(n) < 0.02] = rng.integers(50, 20_000)
        ts = np.cumsum(step)
    elif mode == 4:
        ts = np.cumsum(rng.integers(1, 7, n)) - 3 * n
    elif mode == 5:
        ts = np.cumsum(rng.integers(0, 2, n)) * int(rng.integers(1, 1000))
    else:
        ts = np.sort(rng.integers(-5000, 5000, n))
    return ts.astype(np.int64) + int(rng.integers(-2000, 2000))


def rand_col(rng, n, pad):
    """(values, validity bytes or None, type, arrow offset): a column of n rows living at `pad` rows into longer buffers"""
    tot = n + pad + int(rng.integers(0, 9))
    if rng.random() < 0.5:
        v = rng.standard_normal(tot) * 10.0 ** rng.integers(-3, 6, tot)
        sp = rng.random(tot)
        v[sp < 0.01] = np.nan
        v[(sp >= 0.01) & (sp < 0.02)] = 0.0
        v[(sp >= 0.02) & (sp < 0.03)] = -0.0
        v[(sp >= 0.03) & (sp < 0.035)] = np.inf
        v[(sp >= 0.035) & (sp < 0.04)] = -np.inf
        typ = capi.FLOAT64
    else:
        v = rng.integers(-(2 ** 50), 2 ** 50, tot).astype(np.int64)
        typ = capi.INT64
    if rng.random() < 0.3:   # few distinct values: ties for Mode, equal neighbours for Min / Max / First / Last
        small = rng.integers(-2, 3, tot)
        v = np.where(rng.random(tot) < 0.85, small.astype(v.dtype), v)
    frac = [0.0, 0.0, 0.05, 0.3, 0.9, 1.0][int(rng.integers(0, 6))]
    if frac == 0.0 and rng.random() < 0.5:
        return v, None, typ, pad
    valid = rng.random(tot) >= frac
    return v, np.packbits(valid, bitorder="little"), typ, pad


def run_paths(ccols, ocols, interval, aggs, offset, inclusive, label):
    exp, nic = orc.aggregate(ocols, 0, interval, aggs, offset=offset, inclusive=inclusive)
    bounds = None
    for path in capi.agg_routes():
        outs, info = capi.rolling_aggregate(ccols, 0, interval, aggs, offset=offset, inclusive=inclusive)
        assert info.new_interval_col == nic, label
        if info.long_windows and bounds is None:
            bounds = order_free_bounds(ocols, 0, interval, aggs, offset=offset, inclusive=inclusive, ref=exp)
        for i, (a, g, w) in enumerate(zip(aggs, outs, exp)):
            exact = info.long_windows == 0 or a[0] not in ORDER_SENSITIVE
            compare("%s %s path=%s" % (label, a[0], path), g, w, exact=exact, bound=None if exact else bounds[i])
    # bowgpu_options.strict_order: every window in row order - every reducer bit for bit, whatever the window lengths (round 4:
    # long windows by one lane each); declined only when a window holds more than 2^20 rows
    try:
        outs, info = capi.rolling_aggregate(ccols, 0, interval, aggs, offset=offset, inclusive=inclusive, strict_order=True)
    except capi.BowGpuError as e:
        assert e.code == -9 and "2^20" in e.message, (label, e.message)
        return
    assert info.long_windows == 0, label
    for a, g, w in zip(aggs, outs, exp):
        compare("%s %s strict" % (label, a[0]), g, w, exact=True)


def aggregate_cases(seed):
    """the seeded cases of test_fuzz_aggregate: (ccols, ocols, n, interval, aggs, offset, inclusive_call, label) - also what
    tests/test_gpu_multi.py pushes through the multi-device fan-out"""
    rng = np.random.default_rng(1000 + seed)
    sizes = [0, 1, 2, 63, 64, 65, 511, 512, 513, 639, 640, 641, 1023, 1025, 2047, 2049, 5000, 40_000]
    if os.environ.get("BOW_FUZZ_BIG") == "1":   # soak runs: frames of millions of rows (thousands of tiles per call)
        sizes += [300_000, 1_000_000, 3_000_000]
    for case in range(45):
        n = int(sizes[int(rng.integers(0, len(sizes)))]) if rng.random() < 0.8 else int(rng.integers(0, 3000))
        ts = rand_ts(rng, n)
        interval = int([1, 2, 3, 7, 10, 64, 100, 1000, 12345, 10 ** 6][int(rng.integers(0, 10))])
        while n and (int(ts[-1]) - int(ts[0])) // interval > 1_500_000:
            interval *= 10  # keep the number of windows (output slots) reasonable
        if n and rng.random() < 0.3:
            # the same shape at nanosecond scale: rows span far more than 2^32 from the first window (per-tile window ids)
            scale = int(10 ** rng.integers(5, 10)) + int(rng.integers(0, 3))
            ts = ts * scale + int(rng.integers(-2, 3)) * 1_500_000_000_000_000_000 // 2
            interval *= scale
            if interval >= 2 ** 32 and rng.random() < 0.7:
                interval = int(rng.integers(1, 2 ** 32 - 1))
                while (int(ts[-1]) - int(ts[0])) // interval > 1_500_000:
                    interval = min(interval * 10, 2 ** 62)
        offset = int(rng.integers(-3 * interval, 3 * interval + 1))
        ncols = int(rng.integers(1, 4))
        pad = int(rng.integers(0, 70)) if rng.random() < 0.5 else 0
        raw = [rand_col(rng, n, pad) for _ in range(ncols)]
        ts_buf = np.concatenate([np.zeros(pad, np.int64), ts, np.zeros(3, np.int64)])
        ccols = [capi.Column(ts_buf, None, capi.INT64, pad, n, 0)]
        ocols = [orc.Column(ts_buf, None, orc.INT64, offset=pad, length=n)]
        for v, bm, typ, off in raw:
            ccols.append(capi.Column(v, bm, typ, off, n, -1 if bm is not None else 0))
            ocols.append(orc.Column(v, bm, typ, offset=off, length=n))
        if rng.random() < 0.3:
            ccols = [c.to_device() for c in ccols]
        inclusive = bool(rng.random() < 0.35)
        kinds = list(ALL_AGGS) + (TIME_AGGS if (inclusive or rng.random() < 0.3) else [])
        na = int(rng.integers(1, 9))
        aggs = [("WindowStart", 0)]
        for _ in range(na):
            k = kinds[int(rng.integers(0, len(kinds)))]
            col = 0 if k == "WindowStart" else int(rng.integers(0 if rng.random() < 0.1 else 1, ncols + 1))
            if rng.random() < 0.25:
                aggs.append((k, col, [float(rng.choice([0.5, -1.0, 2.0, 0.1, 1e3])) for _ in range(int(rng.integers(1, 3)))]))
            else:
                aggs.append((k, col))
        if n <= 5000 and rng.random() < 0.4:   # (the oracle's Mode is quadratic in the window's rows)
            aggs.append(("Mode", int(rng.integers(0, ncols + 1))) if rng.random() < 0.7 else
                        ("Mode", int(rng.integers(1, ncols + 1)), [float(rng.choice([0.5, -1.0, 1e3]))]))
        if inclusive and not any(a[0] in ("IntegralTrapezoid", "WeightedAverageLinear") for a in aggs):
            aggs.append(("IntegralTrapezoid", 1))
        inclusive_call = any(a[0] in ("IntegralTrapezoid", "WeightedAverageLinear") for a in aggs)
        label = "seed=%d case=%d n=%d I=%d off=%d pad=%d" % (seed, case, n, interval, offset, pad)
        yield ccols, ocols, n, interval, aggs, offset, inclusive_call, label


@pytest.mark.parametrize("seed", range(int(os.environ.get("BOW_FUZZ_SEEDS", "64"))))
def test_fuzz_aggregate(seed):
    for ccols, ocols, n, interval, aggs, offset, inclusive_call, label in aggregate_cases(seed):
        if n == 0:
            outs, info = capi.rolling_aggregate(ccols, 0, interval, aggs, offset=offset)
            assert all(o.length == 0 for o in outs), label
            continue
        run_paths(ccols, ocols, interval, aggs, offset, inclusive_call, label)


@pytest.mark.parametrize("seed", range(int(os.environ.get("BOW_FUZZ_SEEDS", "64")) // 2))
def test_fuzz_interpolate_and_fills(seed):
    rng = np.random.default_rng(2000 + seed)
    for case in range(40):
        # (sizes are bounded by the ORACLE: like the reference's GetPrevFloat64s walks it is cubic on all-null columns)
        n = int(rng.integers(1, 500)) if rng.random() < 0.7 else int([511, 512, 513, 700][int(rng.integers(0, 4))])
        ts = rand_ts(rng, n)
        interval = int([1, 2, 5, 10, 64, 100, 1000][int(rng.integers(0, 7))])
        offset = int(rng.integers(-2 * interval, 2 * interval + 1))
        pad = int(rng.integers(0, 40)) if rng.random() < 0.5 else 0
        v, bm, typ, off = rand_col(rng, n, pad)
        ts_buf = np.concatenate([np.zeros(pad, np.int64), ts, np.zeros(3, np.int64)])
        ccols = [capi.Column(ts_buf, None, capi.INT64, pad, n, 0), capi.Column(v, bm, typ, off, n, -1 if bm is not None else 0)]
        ocols = [orc.Column(ts_buf, None, orc.INT64, offset=pad, length=n), orc.Column(v, bm, typ, offset=off, length=n)]
        label = "seed=%d case=%d n=%d I=%d off=%d pad=%d" % (seed, case, n, interval, offset, pad)
        kind = ["Linear", "StepPrevious", "None"][int(rng.integers(0, 3))]
        ip = [{"kind": "WindowStart", "col": 0}, {"kind": kind, "col": 1}]
        if rng.random() < 0.3:
            ip[1]["prev"] = (float(ts[0] - 3), True, 42.5, True, 42)
        got = both_interp_kernels(lambda: capi.rolling_interpolate(ccols, 0, interval, ip, offset=offset))
        want = orc.interpolate(ocols, 0, interval, ip, offset=offset)
        cmp_out(label + " interp ts", got[0], want[0])
        cmp_out(label + " interp " + kind, got[1], want[1])
        for method in ("Previous", "Next", "Mean"):
            g, gu = capi.fill(ccols[1], method)
            w, wu = orc.fill(ocols[1], method)
            assert gu == wu, label
            cmp_out(label + " fill " + method, g, w)
        # FillLinear against a sorted reference column (the interval column)
        if n > 0 and (np.diff(ts) >= 0).all():
            g, gu = capi.fill_linear(ccols, 0, 1)
            w, wu = orc.fill_linear(ocols, 0, 1)
            assert gu == wu, label
            cmp_out(label + " fill_linear", g, w)
        # window bounds
        for inclusive in (False, True):
            s0, W, fi, sb, se, inc = capi.window_bounds(ccols[0], interval, offset, inclusive)
            wins = orc.iterate_windows(ocols[0], interval, offset, inclusive)
            assert W == len(wins), label
            assert np.array_equal(fi, [w_["first_index"] for w_ in wins]), label
            assert np.array_equal(sb, [w_["slice_begin"] for w_ in wins]), label
            assert np.array_equal(se, [w_["slice_end"] for w_ in wins]), label


@pytest.mark.parametrize("seed", range(int(os.environ.get("BOW_FUZZ_SEEDS", "64")) // 2))
def test_fuzz_interpolate_then_aggregate_as_one_call(seed):
    """bowgpu_rolling_interpolate_aggregate against oracle interpolate -> oracle aggregate: frames inside the fused kernel's domain
    (windows of a few to ~100 rows, ascending timestamps, any null density, one or two value columns of either type, PrevRow, Factor
    chains) and frames anywhere in the C ABI's (which take the two-call form); every case also through the two-call form of the
    same entry point (capi.ROUTE_NO_FUSED).  A frame whose interpolated interval column comes out unsorted (rows below the first
    window start) must be declined by both forms alike."""
    rng = np.random.default_rng(7000 + seed)
    fused = 0
    for case in range(24):
        in_domain = rng.random() < 0.7
        if in_domain:
            n = int(rng.integers(600, 4000))
            ts = np.cumsum(rng.integers(0, int(rng.integers(2, 30)), n)).astype(np.int64) + int(rng.integers(-3000, 3000))
            if rng.random() < 0.3:
                ts[n // 2:] += int(rng.integers(1000, 200_000))     # a run of empty windows
            span = max(int(ts[-1] - ts[0]), 1)
            interval = max(1, int(span / (n / float(rng.integers(5, 90)))))
        else:
            n = int(rng.integers(1, 500))
            ts = rand_ts(rng, n)
            interval = int([1, 2, 5, 10, 64, 100, 1000][int(rng.integers(0, 7))])
        offset = int(rng.integers(-2 * interval, 2 * interval + 1))
        ncol = 1 + int(rng.random() < 0.4)
        ccols, ocols = [capi.Column(ts, None, capi.INT64)], [orc.Column(ts, None, orc.INT64)]
        ip = [{"kind": "WindowStart", "col": 0}]
        for j in range(ncol):
            v, bm, typ, off = rand_col(rng, n, 0)
            if bm is not None and n > 600 and not np.unpackbits(bm, bitorder="little")[:n].any():
                bm = None      # (an all-null column of thousands of rows: the oracle's neighbour walks are quadratic)
            ccols.append(capi.Column(v, bm, typ, 0, n, -1 if bm is not None else 0))
            ocols.append(orc.Column(v, bm, typ, offset=0, length=n))
            ip.append({"kind": ["Linear", "StepPrevious", "None"][int(rng.integers(0, 3))], "col": 1 + j})
            if rng.random() < 0.3:
                ip[-1]["prev"] = (float(ts[0] - 3), True, 42.5, True, 42)
        kinds = list(rng.choice(ALL_AGGS[1:], size=int(rng.integers(1, 7))))
        aggs = [("WindowStart", 0)] + [(str(k), int(rng.integers(0 if k in ("Count", "NumRows") else 1, ncol + 1))) for k in kinds]
        if rng.random() < 0.3:
            i = int(rng.integers(1, len(aggs)))
            aggs[i] = aggs[i] + ([float(rng.choice([2.0, -1.0, 0.5, 1e3]))],)
        label = "seed=%d case=%d n=%d I=%d off=%d %s %s" % (seed, case, n, interval, offset, [a[0] for a in aggs], [i_["kind"] for i_ in ip])
        res = {}
        for form, mask in (("one call", 0), ("two-call form", capi.ROUTE_NO_FUSED)):
            with capi.route(mask):
                try:
                    res[form] = capi.rolling_interpolate_aggregate(ccols, 0, interval, ip, aggs, offset=offset, strict_order=True)
                    if mask == 0:
                        fused += capi.last_kernel_name() == "rolling_fused_kernel"
                except capi.BowGpuError as e:
                    res[form] = e.code
        if isinstance(res["one call"], int) or isinstance(res["two-call form"], int):
            assert res["one call"] == res["two-call form"] == -14, (label, res)   # the interpolated interval column is not ascending
            continue
        mid = orc.interpolate(ocols, 0, interval, ip, offset=offset)
        want, nic = orc.aggregate(mid, 0, interval, aggs, offset=offset)
        for form, (outs, info) in res.items():
            assert info.new_interval_col == nic and info.long_windows == 0, label
            for a, g, w in zip(aggs, outs, want):
                compare("%s %s %s" % (label, form, a[0]), g, w)
    assert fused >= 4, fused       # (the fused kernel is what these seeds exercise: ~17 of 24 frames are of the first kind, those
                                   #  with time-weighted reducers, rows below s0 or the -1 sentinel fall back - 7 .. 15 over 300 seeds)


@pytest.mark.parametrize("seed", range(int(os.environ.get("BOW_FUZZ_SEEDS", "64")) // 2))
def test_fuzz_sharded(seed):
    """random row-range splits (empty shards, one-row shards, shards smaller than a window) through the real protocol of
    bow_amd/sharded.py on simulated ranks; the stitched result must equal the oracle on the whole frame"""
    from test_gpu_sharded import run_sharded, AGGS as PLAIN_AGGS, TW_AGGS, ORDER as PLAIN_ORDER
    rng = np.random.default_rng(3000 + seed)
    for case in range(12):
        # plain reducers on exclusive windows, or the time-weighted ones (which make every window inclusive)
        tw = bool(rng.random() < 0.5)
        AGGS = TW_AGGS if tw else PLAIN_AGGS
        ORDER = {"IntegralStep", "WeightedAverageStep", "IntegralTrapezoid", "WeightedAverageLinear", "ArithmeticMean"} if tw else PLAIN_ORDER
        n = int(rng.integers(2, 6000))
        ts = rand_ts(rng, n)
        if ts[0] < 0 and rng.random() < 0.5:
            ts = ts - ts[0]
        interval = int([1, 3, 7, 10, 64, 100, 1000, 5000][int(rng.integers(0, 8))])
        while (int(ts[-1]) - int(ts[0])) // interval > 1_000_000:
            interval *= 10
        offset = int(rng.integers(-2 * interval, 2 * interval + 1))
        vals = rng.standard_normal(n) * 100
        valid = rng.random(n) >= [0.0, 0.2, 0.9][int(rng.integers(0, 3))]
        world = int(rng.integers(2, 7))
        cuts = np.sort(rng.integers(0, n + 1, world - 1))
        bounds = [0] + [int(c) for c in cuts] + [n]
        label = "seed=%d case=%d n=%d I=%d off=%d bounds=%s" % (seed, case, n, interval, offset, bounds)
        bm = np.packbits(valid, bitorder="little")
        try:
            exp, _ = orc.aggregate([orc.Column(ts, None, orc.INT64), orc.Column(vals, bm, orc.FLOAT64)], 0, interval, AGGS, offset=offset,
                                   inclusive=tw)
        except orc.OracleError:
            continue
        # (frames with rows below s0 - Go's truncating division on a negative first timestamp - are included: the shard
        # protocol settles them with its second exchange)
        res, plan = run_sharded(ts, vals, valid, bounds, interval, offset=offset, aggs=AGGS)
        tol = order_free_bounds([orc.Column(ts, None, orc.INT64), orc.Column(vals, bm, orc.FLOAT64)], 0, interval, AGGS, offset=offset,
                                inclusive=tw, ref=exp)
        for i, ((k, _), (gv, gm, typ), w) in enumerate(zip(AGGS, res, exp)):
            assert len(gv) == w.length, (label, k, len(gv), w.length)
            wm = w.valid_mask()
            assert np.array_equal(gm, wm), (label, k)
            wv = w.values[:w.length].view(np.uint64)
            if k in ORDER:
                assert_within((label, k), gv.view(np.float64)[gm], wv.view(np.float64)[wm], tol[i][wm])
            else:
                assert np.array_equal(gv[gm], wv[wm]), (label, k)


@pytest.mark.parametrize("seed", range(int(os.environ.get("BOW_FUZZ_SEEDS", "64")) // 2))
def test_fuzz_sharded_interpolate(seed):
    """Rolling.Interpolate over random row-range splits (empty shards, one-row shards): concatenated shard outputs == the oracle on
    the whole frame"""
    from bow_amd import sharded
    rng = np.random.default_rng(4000 + seed)
    for case in range(15):
        n = int(rng.integers(2, 500))
        ts = rand_ts(rng, n)
        ts = ts - min(int(ts[0]), 0) + int(rng.integers(0, 50))      # non-negative frame: the sharded path's domain
        interval = int([1, 2, 5, 10, 64, 100, 1000][int(rng.integers(0, 7))])
        offset = int(rng.integers(0, interval))
        s0 = sharded.first_window_start(int(ts[0]), interval, offset)
        if s0 < 0 or s0 > ts[0]:
            continue
        v, bm, typ, _ = rand_col(rng, n, 0)
        kind = ["Linear", "StepPrevious", "None"][int(rng.integers(0, 3))]
        ip = [{"kind": "WindowStart", "col": 0}, {"kind": kind, "col": 1}]
        world = int(rng.integers(2, 6))
        bounds = [0] + [int(c) for c in np.sort(rng.integers(0, n + 1, world - 1))] + [n]
        # inclusive windows (rolling.go:201-209): on frames whose windows hold two rows or more on average (the device path's domain)
        inclusive = bool(rng.random() < 0.4) and (int(ts[-1]) - s0) // interval + 1 <= n // 2
        label = "seed=%d case=%d n=%d I=%d off=%d %s bounds=%s incl=%d" % (seed, case, n, interval, offset, kind, bounds, inclusive)
        valid = np.ones(n, bool) if bm is None else np.unpackbits(bm, bitorder="little")[:n].astype(bool)
        want = orc.interpolate([orc.Column(ts, None, orc.INT64), orc.Column(v[:n], None if bm is None else np.packbits(valid, bitorder="little"), typ)],
                               0, interval, ip, offset=offset, inclusive=inclusive)
        shards = []
        for r in range(world):
            a, b = bounds[r], bounds[r + 1]
            shards.append([capi.Column(ts[a:b].copy(), None, capi.INT64).to_device(),
                           capi.Column(v[a:b].copy(), np.packbits(valid[a:b], bitorder="little"), typ, 0, b - a, -1).to_device()])
        points = [capi.shard_interp_points(cols, 0) for cols in shards]
        run = lambda: [capi.shard_interpolate(cols, 0, interval, ip, s0, r, points, offset=offset, inclusive=inclusive) for r, cols in enumerate(shards)]
        try:
            outs = both_interp_kernels(run)
        except capi.BowGpuError as e:
            # (a SHARD may hold windows shorter than two rows on average although the frame does not: declined, not wrong)
            assert inclusive and e.code == -9, (label, str(e))
            continue
        for c in range(2):
            gv = np.concatenate([o[c].host_arrays()[0].view(np.uint64) for o in outs])
            gm = np.concatenate([o[c].valid_mask() for o in outs])
            wm = want[c].valid_mask()
            assert len(gv) == want[c].length, (label, c, len(gv), want[c].length)
            assert np.array_equal(gm, wm), (label, c, np.flatnonzero(gm != wm)[:10])
            wv = want[c].values[:want[c].length].view(np.uint64)
            diff = gv[gm] != wv[wm]
            if diff.any() and typ == capi.FLOAT64 and c == 1:
                diff &= ~(np.isnan(gv[gm].view(np.float64)) & np.isnan(wv[wm].view(np.float64)))
            assert not diff.any(), (label, c, np.flatnonzero(diff)[:10])
