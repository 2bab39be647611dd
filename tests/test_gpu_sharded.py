"""Row-range sharding on ONE GPU: K simulated ranks each hold a row range in HBM and
run the real protocol behind the C ABI (bowgpu_shard_begin -> the gathered records -> bowgpu_shard_finish; bow_amd/sharded.py
is the transport) with the HIP provider; the stitched windows must equal the oracle on the whole frame."""
import numpy as np
import pytest

from bow_amd import capi, sharded
from test_gpu_callers import both_interp_kernels
from oracle import pyoracle as orc
from tolerance import assert_within, order_free_bounds

pytestmark = pytest.mark.gpu

AGGS = [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1), ("Count", 1), ("First", 1),
        ("Last", 1), ("NumRows", 1)]
ORDER = {"Sum", "ArithmeticMean"}


TW_AGGS = [("WindowStart", 0), ("IntegralStep", 1), ("WeightedAverageStep", 1), ("IntegralTrapezoid", 1), ("WeightedAverageLinear", 1),
           ("ArithmeticMean", 1), ("Count", 1), ("Last", 1), ("NumRows", 1)]


def run_sharded(ts, vals, valid, bounds, interval, offset=0, aggs=None, strict_order=False, host=False):
    """bounds: row boundaries [0, b1, ..., n] of the simulated ranks; host: the shards' columns and outputs live in host memory"""
    AGGS = aggs if aggs is not None else globals()["AGGS"]
    world = len(bounds) - 1
    provs = []
    for r in range(world):
        a, b = bounds[r], bounds[r + 1]
        bm = None if valid is None else np.packbits(valid[a:b], bitorder="little")
        cols = [capi.Column(ts[a:b].copy(), None, capi.INT64), capi.Column(vals[a:b].copy(), bm, capi.FLOAT64 if vals.dtype == np.float64 else capi.INT64, 0, b - a, -1)]
        if not host:
            cols = [c_.to_device() for c_ in cols]
        provs.append(sharded.GpuProvider(cols, 0, interval, AGGS, offset=offset, strict_order=strict_order, out_residency=capi.HOST if host else None))
    decisions = sharded.run_local(provs)
    owned = [(d.first_slot_window_id, d.windows_owned) for d in decisions]
    # assemble the global result from what each rank owns
    W = max((fs + n for fs, n in owned if fs >= 0), default=0)   # (0: every row lies below the first window start - the reference builds no window)
    res = []
    for i, (k, _) in enumerate(AGGS):
        vals_g = np.zeros(W, dtype=np.uint64)
        valid_g = np.zeros(W, dtype=bool)
        seen = np.zeros(W, dtype=int)
        for r, (fs, n) in enumerate(owned):
            if fs < 0 or n <= 0:
                continue
            v, _ = provs[r].outs[i].host_arrays()
            m = provs[r].outs[i].valid_mask()
            vals_g[fs:fs + n] = v.view(np.uint64)[:n]
            valid_g[fs:fs + n] = m[:n]
            seen[fs:fs + n] += 1
        assert (seen == 1).all(), (k, np.flatnonzero(seen != 1)[:10])
        res.append((vals_g, valid_g, provs[0].outs[i].type))
    return res, sharded.ShardPlan(decisions=decisions)


@pytest.mark.parametrize("mode", ["dense", "irregular", "gappy"])
def test_sharded_equals_whole(mode):
    rng = np.random.default_rng(77)
    n = 40_000
    if mode == "dense":
        ts = np.arange(n, dtype=np.int64)
    elif mode == "irregular":
        ts = np.cumsum(rng.integers(1, 20, n)).astype(np.int64)
    else:
        step = rng.integers(1, 5, n)
        step[rng.random(n) < 0.002] = rng.integers(100, 3000)
        ts = np.cumsum(step).astype(np.int64)
    vals = rng.standard_normal(n) * 100
    valid = rng.random(n) >= 0.2
    for interval, bounds in [(7, [0, 10_000, 20_000, 30_000, n]),          # windows straddle every boundary
                             (10, [0, 9_999, 20_001, 20_002, n]),          # a 1-row shard
                             (64, [0, 13, 40, 41, 20_000, n]),             # shards smaller than a window: 3+ ranks per window
                             (1000, [0, 5_000, 5_000, 25_000, n])]:        # an EMPTY shard in the middle
        res, plan = run_sharded(ts, vals, valid, bounds, interval)
        bm = np.packbits(valid, bitorder="little")
        exp, _ = orc.aggregate([orc.Column(ts, None, orc.INT64), orc.Column(vals, bm, orc.FLOAT64)], 0, interval, AGGS)
        multi = any(len(plan.seed_ranks(r)) > 1 for r in range(plan.world))
        tol = order_free_bounds([orc.Column(ts, None, orc.INT64), orc.Column(vals, bm, orc.FLOAT64)], 0, interval, AGGS, ref=exp)
        for i, ((k, _), (gv, gm, typ), w) in enumerate(zip(AGGS, res, exp)):
            assert len(gv) == w.length, (k, len(gv), w.length)
            wm = w.valid_mask()
            assert np.array_equal(gm, wm), (mode, interval, k)
            wv = w.values[:w.length].view(np.uint64)
            if k in ORDER and (multi or interval >= 129):
                assert_within((mode, interval, k), gv.view(np.float64)[gm], wv.view(np.float64)[wm], tol[i][wm])
            else:
                bad = np.flatnonzero(gv[gm] != wv[wm])
                assert bad.size == 0, (mode, interval, k, bad[:5])


def test_sharded_strict_order_is_row_order_across_a_shard_boundary():
    """bowgpu_options.strict_order on a sharded call (round 5; declined before): a rank's own windows by the unsharded forms (long ones
    walked by one lane each), a window shared by two ranks by the right rank's re-walk seeded with the left rank's running state -
    every reducer bit for bit, windows of 10 .. 3000 rows, nulls, offsets that put a window across every boundary; a window spread
    over three ranks is declined (its middle rank would contribute a partial sum)."""
    rng = np.random.default_rng(5)
    n = 60_000
    ts = np.cumsum(rng.integers(1, 4, n)).astype(np.int64)
    vals = rng.standard_normal(n) * 1e6
    valid = rng.random(n) >= 0.25
    bm = np.packbits(valid, bitorder="little")
    aggs = AGGS + [("IntegralStep", 1), ("WeightedAverageLinear", 1)]
    for interval, offset, bounds in [(20, 3, [0, 15_000, 30_001, 44_444, n]), (400, 7, [0, 20_000, 40_000, n]), (6_000, 1, [0, 19_999, 41_000, n])]:
        res, plan = run_sharded(ts, vals, valid, bounds, interval, offset=offset, aggs=aggs, strict_order=True)
        exp, _ = orc.aggregate([orc.Column(ts, None, orc.INT64), orc.Column(vals, bm, orc.FLOAT64)], 0, interval, aggs, offset=offset)
        for (k, _), (gv, gm, typ), w in zip(aggs, res, exp):
            wm = w.valid_mask()
            assert len(gv) == w.length and np.array_equal(gm, wm), (interval, k)
            wv = w.values[:w.length].view(np.uint64)
            diff = gv[gm] != wv[wm]
            if diff.any():   # (generated NaNs carry the hardware's default payload: test_gpu_aggregate.compare)
                diff &= ~(np.isnan(gv.view(np.float64)[gm]) & np.isnan(wv.view(np.float64)[wm]))
            assert not diff.any(), (interval, k, np.flatnonzero(diff)[:5])
    with pytest.raises(capi.BowGpuError) as e:      # ranks of 100 rows, windows of ~1000: every window is spread over many ranks
        run_sharded(ts[:2000], vals[:2000], valid[:2000], list(range(0, 2001, 100)), 2000, aggs=AGGS, strict_order=True)
    assert e.value.code == -9 and "three or more shards" in e.value.message


def test_sharded_host_resident_columns_and_outputs():
    """the record protocol on shards whose columns AND outputs live in host memory (round 5; declined before): staged through HBM per
    call like the unsharded entry points, the pass put in flight by bowgpu_shard_pass_begin keeps its staged copies until
    bowgpu_shard_finish collects it.  Same bits as the device-resident run, inclusive windows and the exchange of first rows included."""
    rng = np.random.default_rng(8)
    n = 30_000
    ts = np.cumsum(rng.integers(1, 9, n)).astype(np.int64)
    vals = rng.standard_normal(n)
    valid = rng.random(n) >= 0.3
    for aggs in (AGGS, TW_AGGS):
        for interval, bounds in ((13, [0, 7_000, 7_001, 20_000, n]), (500, [0, 10_000, 10_000, n])):
            dev, _ = run_sharded(ts, vals, valid, bounds, interval, offset=5, aggs=aggs)
            hst, _ = run_sharded(ts, vals, valid, bounds, interval, offset=5, aggs=aggs, host=True)
            for (k, _c), (gv, gm, t1), (hv, hm, t2) in zip(aggs, dev, hst):
                assert t1 == t2 and np.array_equal(gm, hm) and np.array_equal(gv, hv), (k, interval)


def test_sharded_gaps_between_shards():
    # the first row of a shard is many empty windows after the last row of the previous one
    ts = np.concatenate([np.arange(0, 1000), np.arange(50_000, 51_000), np.arange(200_000, 200_500)]).astype(np.int64)
    vals = np.arange(len(ts), dtype=np.float64)
    res, plan = run_sharded(ts, vals, None, [0, 1000, 2000, len(ts)], 10)
    exp, _ = orc.aggregate([orc.Column(ts, None, orc.INT64), orc.Column(vals, None, orc.FLOAT64)], 0, 10, AGGS)
    assert plan.lead_empty(1) > 0 and plan.lead_empty(2) > 0
    for (k, _), (gv, gm, typ), w in zip(AGGS, res, exp):
        wm = w.valid_mask()
        assert len(gv) == w.length and np.array_equal(gm, wm), k
        assert np.array_equal(gv[gm], w.values[:w.length].view(np.uint64)[wm]), k
        assert not gv[~gm].any(), k


@pytest.mark.parametrize("mode", ["dense", "irregular", "gappy"])
def test_sharded_time_weighted_and_inclusive_windows(mode):
    """IntegralStep / WeightedAverageStep carry their last point across the boundary; IntegralTrapezoid / WeightedAverageLinear make
    every window inclusive, so a window that ends exactly where a shard ends takes the NEXT shard's first row (shipped with the plan
    exchange), and a straddling one takes it from wherever its successor starts"""
    rng = np.random.default_rng(91)
    n = 30_000
    if mode == "dense":
        ts = np.arange(n, dtype=np.int64)
    elif mode == "irregular":
        ts = np.cumsum(rng.integers(1, 20, n)).astype(np.int64)
    else:
        step = rng.integers(1, 5, n)
        step[rng.random(n) < 0.003] = rng.integers(100, 3000)
        ts = np.cumsum(step).astype(np.int64)
    vals = np.round(rng.standard_normal(n) * 100, 3)
    valid = rng.random(n) >= 0.2
    cases = [(10, [0, 10_000, 20_000, n]),             # dense: shard ends ARE window ends => next shard's first row is the inclusive row
             (7, [0, 9_999, 20_001, 20_002, n]),       # straddling windows, a one-row shard
             (64, [0, 13, 40, 41, 15_000, n]),         # three and more ranks per window
             (1000, [0, 5_000, 5_000, 25_000, n])]     # an empty shard in the middle
    for interval, bounds in cases:
        res, plan = run_sharded(ts, vals, valid, bounds, interval, aggs=TW_AGGS)
        bm = np.packbits(valid, bitorder="little")
        exp, _ = orc.aggregate([orc.Column(ts, None, orc.INT64), orc.Column(vals, bm, orc.FLOAT64)], 0, interval, TW_AGGS, inclusive=True)
        multi = any(len(plan.seed_ranks(r)) > 1 for r in range(plan.world))
        tol = order_free_bounds([orc.Column(ts, None, orc.INT64), orc.Column(vals, bm, orc.FLOAT64)], 0, interval, TW_AGGS, inclusive=True, ref=exp)
        for i, ((k, _), (gv, gm, typ), w) in enumerate(zip(TW_AGGS, res, exp)):
            assert len(gv) == w.length, (mode, interval, k, len(gv), w.length)
            wm = w.valid_mask()
            assert np.array_equal(gm, wm), (mode, interval, k, np.flatnonzero(gm != wm)[:8])
            wv = w.values[:w.length].view(np.uint64)
            if k in ("WindowStart", "Count", "Last", "NumRows"):
                assert np.array_equal(gv[gm], wv[wm]), (mode, interval, k)
            elif multi or interval >= 129:
                assert_within((mode, interval, k), gv.view(np.float64)[gm], wv.view(np.float64)[wm], tol[i][wm])
            else:
                bad = np.flatnonzero(gv[gm] != wv[wm])
                assert bad.size == 0, (mode, interval, k, bad[:5], gv.view(np.float64)[gm][bad[:3]], wv.view(np.float64)[wm][bad[:3]])


@pytest.mark.parametrize("kind", ["Linear", "StepPrevious", "None"])
def test_sharded_interpolate_equals_whole(kind):
    """Rolling.Interpolate over row-range shards: every shard emits the synthetic rows in front of ITS rows (including the
    empty windows since the last row to its left) and finds the nearest valid neighbours of a window start on other shards
    through the exchanged first / last valid points; the outputs concatenated in rank order are the unsharded result."""
    rng = np.random.default_rng(17)
    ip = [{"kind": "WindowStart", "col": 0}, {"kind": kind, "col": 1}, {"kind": "Linear" if kind != "Linear" else "StepPrevious", "col": 2}]
    for mode, n, interval, offset, bounds in [
            ("irregular", 20_000, 100, 7, [0, 5_000, 5_001, 12_000, 20_000]),
            ("dense", 9_000, 10, 0, [0, 3_000, 6_000, 9_000]),               # shard ends are window ends: exact heads on the boundary
            ("gappy", 15_000, 50, 3, [0, 4_000, 4_000, 9_000, 15_000]),      # an empty shard; long runs of empty windows
            ("irregular", 6_000, 1000, 0, [0, 10, 20, 30, 6_000])]:          # shards smaller than a window
        if mode == "dense":
            ts = np.arange(n, dtype=np.int64) * 2
        elif mode == "irregular":
            ts = np.cumsum(rng.integers(1, 20, n)).astype(np.int64)
        else:
            step = rng.integers(1, 5, n)
            step[rng.random(n) < 0.004] = rng.integers(100, 3000)
            ts = np.cumsum(step).astype(np.int64)
        v1 = np.round(rng.standard_normal(n) * 100, 2)
        v2 = rng.integers(-1000, 1000, n).astype(np.int64)
        m1, m2 = rng.random(n) >= 0.4, rng.random(n) >= 0.4
        m1[bounds[1] - 40:bounds[1] + 60] = False       # the nearest valid points of the boundary windows lie deep inside the neighbours
        m2[:bounds[1]] = rng.random(bounds[1]) >= 0.97
        b1, b2 = np.packbits(m1, bitorder="little"), np.packbits(m2, bitorder="little")
        for inclusive in (False, True):      # (inclusive windows, rolling.go:201-209: every window puts one row in front of its first - a synthetic row or the copy of a row on its start)
            if inclusive and interval < 50:
                continue                        # (windows of two rows and more on average: the inclusive kernel's domain)
            want = orc.interpolate([orc.Column(ts, None, orc.INT64), orc.Column(v1, b1, orc.FLOAT64), orc.Column(v2, b2, orc.INT64)],
                                   0, interval, ip, offset=offset, inclusive=inclusive)
            s0 = sharded.first_window_start(int(ts[0]), interval, offset)
            shards = []
            for r in range(len(bounds) - 1):
                a, b = bounds[r], bounds[r + 1]
                shards.append([capi.Column(ts[a:b].copy(), None, capi.INT64).to_device(),
                               capi.Column(v1[a:b].copy(), np.packbits(m1[a:b], bitorder="little"), capi.FLOAT64, 0, b - a, -1).to_device(),
                               capi.Column(v2[a:b].copy(), np.packbits(m2[a:b], bitorder="little"), capi.INT64, 0, b - a, -1).to_device()])
            points = [capi.shard_interp_points(cols, 0) for cols in shards]
            outs = both_interp_kernels(lambda: [capi.shard_interpolate(cols, 0, interval, ip, s0, r, points, offset=offset, inclusive=inclusive) for r, cols in enumerate(shards)])
            for c in range(3):
                gv = np.concatenate([o[c].host_arrays()[0].view(np.uint64) for o in outs])
                gm = np.concatenate([o[c].valid_mask() for o in outs])
                wm = want[c].valid_mask()
                assert len(gv) == want[c].length, (mode, inclusive, c, len(gv), want[c].length)
                assert np.array_equal(gm, wm), (mode, inclusive, c, np.flatnonzero(gm != wm)[:10])
                wv = want[c].values[:want[c].length].view(np.uint64)
                assert np.array_equal(gv[gm], wv[wm]), (mode, inclusive, c, np.flatnonzero(gv[gm] != wv[wm])[:10])


def test_sharded_interpolate_negative_timestamps():
    """the sharded Interpolate on frames that start below 0 - negative window starts, the window that
    starts at -1 (the reference's "no first value" sentinel, interpolation.go:99-119: it never gets a synthetic row) with and without a row
    of its own and on either side of a shard boundary, and rows below the first window start (Go's truncating division, rolling.go:96-99)
    on the frame's first shard.  Shards concatenated = the oracle's unsharded frame, bit for bit, through both kernels."""
    rng = np.random.default_rng(23)
    ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}, {"kind": "StepPrevious", "col": 2}]
    cases = []
    for interval, offset, t0, n, sentinel in [(100, 0, -5_017, 4_000, None), (100, 99, -5_017, 4_000, "empty"), (100, 99, -5_017, 4_000, "row"),
                                              (10, 9, -2_003, 3_000, "empty"), (7, 0, -777, 2_500, None), (50, 49, -1_051, 2_000, "row")]:
        ts = t0 + np.cumsum(rng.integers(1, 9, n)).astype(np.int64)
        if sentinel == "empty":
            ts = ts[(ts < -1) | (ts >= -1 + interval)]        # the window [-1, -1 + interval) has no row
        elif sentinel == "row":
            assert ((ts >= -1) & (ts < -1 + interval)).any()
        cases.append((ts, interval, offset))
    # rows below s0 that ride in window 0 (a row at or above s0 inside its first interval) and rows that belong to no window (none there)
    cases.append((np.concatenate([np.array([-15, -12, -11, -9], dtype=np.int64), np.cumsum(rng.integers(1, 9, 1500)).astype(np.int64)]), 10, 9))
    cases.append((np.concatenate([np.array([-15, -12], dtype=np.int64), 100 + np.cumsum(rng.integers(1, 9, 1500)).astype(np.int64)]), 10, 9))
    served = 0
    for ts, interval, offset in cases:
        n = len(ts)
        v1 = np.round(rng.standard_normal(n) * 100, 2)
        v2 = rng.integers(-1000, 1000, n).astype(np.int64)
        m1, m2 = rng.random(n) >= 0.3, rng.random(n) >= 0.3
        b1, b2 = np.packbits(m1, bitorder="little"), np.packbits(m2, bitorder="little")
        want = orc.interpolate([orc.Column(ts, None, orc.INT64), orc.Column(v1, b1, orc.FLOAT64), orc.Column(v2, b2, orc.INT64)], 0, interval, ip, offset=offset)
        s0 = sharded.first_window_start(int(ts[0]), interval, offset)
        zero = int(np.searchsorted(ts, -1))                    # the first row at or above -1: a boundary right there, and one inside the window
        for bounds in ([0, n // 3, 2 * n // 3, n], [0, max(zero, 1), min(zero + 2, n - 1), n], [0, 5, n], [0, max(zero - 3, 1), n]):
            bounds = sorted(set(bounds))
            shards = []
            for r in range(len(bounds) - 1):
                a, b = bounds[r], bounds[r + 1]
                shards.append([capi.Column(ts[a:b].copy(), None, capi.INT64).to_device(),
                               capi.Column(v1[a:b].copy(), np.packbits(m1[a:b], bitorder="little"), capi.FLOAT64, 0, b - a, -1).to_device(),
                               capi.Column(v2[a:b].copy(), np.packbits(m2[a:b], bitorder="little"), capi.INT64, 0, b - a, -1).to_device()])
            points = [capi.shard_interp_points(cols, 0) for cols in shards]
            try:
                outs = both_interp_kernels(lambda: [capi.shard_interpolate(cols, 0, interval, ip, s0, r, points, offset=offset) for r, cols in enumerate(shards)])
            except capi.BowGpuError as e:
                # the one corner that stays declined: a first shard of nothing but rows below the first window start
                assert e.code == -9 and ts[0] < s0 and ts[bounds[1] - 1] < s0, (list(ts[:6]), interval, offset, bounds, str(e))
                continue
            served += 1
            for c in range(3):
                gv = np.concatenate([o[c].host_arrays()[0].view(np.uint64) for o in outs])
                gm = np.concatenate([o[c].valid_mask() for o in outs])
                wm = want[c].valid_mask()
                label = (list(ts[:4]), interval, offset, bounds, c)
                assert len(gv) == want[c].length, label + (len(gv), want[c].length)
                assert np.array_equal(gm, wm), label + (np.flatnonzero(gm != wm)[:10],)
                wv = want[c].values[:want[c].length].view(np.uint64)
                assert np.array_equal(gv[gm], wv[wm]), label + (np.flatnonzero(gv[gm] != wv[wm])[:10],)
    assert served >= 4 * len(cases) - 4, served


@pytest.mark.parametrize("tw", [False, True])
def test_sharded_window_0_of_rows_below_s0_only(tw):
    """negative timestamps: Go's truncating division puts s0 above the first rows (rolling.go:96-99), window 0 spans them and is an
    EMPTY slice unless one of its rows reaches s0 or it takes an inclusive row (rolling.go:194-228).  Frames whose window 0 holds
    ONLY rows below s0, split across ranks in every way, with plain and time-weighted (inclusive) reducers: the window stitched
    from the shards' running states must come out empty, as in the unsharded call and the oracle."""
    aggs = TW_AGGS if tw else AGGS
    frames = [
        # ([-15, -12] with interval 10: s0 = -10 + 0 ... offset below picks s0 = -11) then a far row: window 0 = rows below s0 only
        (np.array([-15, -12, 100, 101, 130], dtype=np.int64), 10, 9),
        (np.array([-15, -14, -13, -12, 100], dtype=np.int64), 10, 9),
        (np.array([-15, -12, -11, 100], dtype=np.int64), 10, 9),          # ... and one where a row DOES reach s0 = -11
        (np.array([-15, -12, -1, 100], dtype=np.int64), 10, 9),           # a row exactly on window 0's end (-1): its inclusive row
        (np.array([-25, -23, -22, 17, 40], dtype=np.int64), 20, 19),
    ]
    rng = np.random.default_rng(5)
    for ts, interval, offset in frames:
        n = len(ts)
        vals = np.round(rng.standard_normal(n) * 10, 2)
        valid = np.ones(n, bool)
        bm = np.packbits(valid, bitorder="little")
        ocols = [orc.Column(ts, None, orc.INT64), orc.Column(vals, bm, orc.FLOAT64)]
        s0, W = orc.plan_windows(ocols[0], interval, offset)
        assert s0 > ts[0]                                   # the corner this test is about
        exp, _ = orc.aggregate(ocols, 0, interval, aggs, offset=offset, inclusive=tw)
        whole, _ = capi.rolling_aggregate([capi.Column(ts), capi.Column(vals, bm, capi.FLOAT64, 0, n, -1)], 0, interval, aggs, offset=offset)
        for cut in [[0, k, n] for k in range(1, n)] + [[0, 1, 2, n], [0, 2, 2, 3, n], [0, 1, 2, 3, n]]:
            res, plan = run_sharded(ts, vals, valid, cut, interval, offset=offset, aggs=aggs)
            for (k, _), (gv, gm, typ), w, wh in zip(aggs, res, exp, whole):
                label = (list(ts), cut, k)
                assert len(gv) == w.length == wh.length, label
                wm = w.valid_mask()
                assert np.array_equal(gm, wm), label
                wv = w.values[:w.length].view(np.uint64)
                assert np.array_equal(gv[gm], wv[wm]), label


def test_carry_only_equals_the_carry_of_the_pass():
    """bowgpu_shard_carry_only (what the bench's ranks exchange while their main pass runs) against the carry that
    bowgpu_shard_aggregate returns, over random splits - empty shards, one-row shards, last windows of thousands of rows,
    rows below s0 on the first shard"""
    rng = np.random.default_rng(77)
    for case in range(40):
        n = int(rng.integers(1, 30_000))
        ts = (np.cumsum(rng.integers(0, 4, n)) - int(rng.choice([0, 5000]))).astype(np.int64)
        vals = np.round(rng.standard_normal(n) * 100, 1)
        valid = rng.random(n) > rng.choice([0.0, 0.3])
        interval = int(rng.choice([3, 10, 1000, 20_000]))
        offset = int(rng.integers(-interval, interval + 1))
        K = int(rng.integers(2, 6))
        cuts = np.sort(rng.integers(0, n + 1, K - 1))
        bounds = list(zip([0] + list(cuts), list(cuts) + [n]))
        first_nonempty = next(a for a, b in bounds if b > a)
        s0 = sharded.first_window_start(int(ts[first_nonempty]), interval, offset)
        aggs = TW_AGGS[:3] + AGGS[1:5] if case % 2 else AGGS
        aggs = [a for a in aggs if a[0] not in ("IntegralTrapezoid", "WeightedAverageLinear")]
        for r, (a, b) in enumerate(bounds):
            if b == a:
                continue
            cols = [capi.Column(ts[a:b].copy(), None, capi.INT64).to_device(),
                    capi.Column(vals[a:b].copy(), np.packbits(valid[a:b], bitorder="little"), capi.FLOAT64, 0, b - a, -1).to_device()]
            prov = sharded.GpuProvider(cols, 0, interval, aggs, offset=offset)
            prov.first_last_nrows()
            early = prov.shard_carry_only(s0, a == 0)
            full = prov.shard_aggregate(s0, a == 0, 0)
            assert early == full, (case, r, n, interval, offset)
            # ... and the states bowgpu_shard_begin puts in the rank's record (computed on the offset-aligned grid, before any
            # rank knows s0) are those same states whenever the record covers the rank's whole last window
            rec = capi.ShardRecord.from_buffer_copy(prov.begin())
            car = capi.ShardCarry.from_buffer_copy(full)
            if s0 <= int(ts[a]):
                assert bytes(rec.last) == bytes(car.last), (case, r, n, interval, offset)
