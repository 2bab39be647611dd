"""ONE Rolling.Aggregate call over N devices behind the C ABI (bowgpu_set_devices, bow_amd/csrc/multi.cpp; reference
rolling/aggregation.go:123-145 - the user makes one call).  On a one-GPU box the same device is listed N times: N library threads, N
contexts and streams, N row ranges, the records exchanged in host memory, every rank putting its own windows into the caller's buffers.
Checked against the oracle AND against the one-device call of the same inputs (bit for bit wherever no window took an order-free form)."""
import os
import threading

import numpy as np
import pytest

from bow_amd import capi
from oracle import pyoracle as orc
from test_gpu_aggregate import ALL_AGGS, TIME_AGGS, ORDER_SENSITIVE, compare
from test_gpu_fuzz import aggregate_cases
from tolerance import order_free_bounds

pytestmark = pytest.mark.gpu


def same_bits(label, a, b):
    """two OutColumns of the same call: identical in every byte the ABI defines"""
    assert a.length == b.length and a.type == b.type and a.null_count == b.null_count, (label, a.length, b.length, a.null_count, b.null_count)
    av, ab = a.host_arrays()
    bv, bb = b.host_arrays()
    assert np.array_equal(ab, bb), (label, "validity")
    diff = av.view(np.uint64) != bv.view(np.uint64)
    if diff.any() and a.type == capi.FLOAT64:
        # what "bit-exact" excludes everywhere (include/bowgpu.h): sign and payload of a NaN the arithmetic GENERATES or passes on through an
        # addition - two NaN operands keep the FIRST one's bits, and a window stitched across two ranks adds its halves in another operand
        # order than the one-lane walk does (found by the 200-seed soak: IntegralStep over a column of NaN / Inf, 0x7FF8... against 0xFFF8...)
        diff &= ~(np.isnan(av.view(np.float64)) & np.isnan(bv.view(np.float64)))
    assert not diff.any(), (label, np.flatnonzero(diff)[:10])


def check_case(ccols, ocols, interval, aggs, offset, inclusive, label, ids, min_rows, strict=False, out_residency=capi.HOST, expect_ranks=None):
    exp, nic = orc.aggregate(ocols, 0, interval, aggs, offset=offset, inclusive=inclusive)
    one, info1 = capi.rolling_aggregate(ccols, 0, interval, aggs, offset=offset, inclusive=inclusive, strict_order=strict, out_residency=out_residency)
    assert capi.last_call_ranks() == 1
    with capi.devices(ids, min_rows=min_rows):
        outs, info = capi.rolling_aggregate(ccols, 0, interval, aggs, offset=offset, inclusive=inclusive, strict_order=strict, out_residency=out_residency)
        ranks = capi.last_call_ranks()
    if expect_ranks is not None:
        assert ranks == expect_ranks, (label, ranks)
    assert (info.s0, info.num_windows, info.new_interval_col, info.inclusive) == (info1.s0, info1.num_windows, info1.new_interval_col, info1.inclusive), label
    assert info.new_interval_col == nic, label
    bounds = None
    if info.long_windows:
        bounds = order_free_bounds(ocols, 0, interval, aggs, offset=offset, inclusive=inclusive, ref=exp)
    for i, (a, g, w) in enumerate(zip(aggs, outs, exp)):
        exact = info.long_windows == 0 or a[0] not in ORDER_SENSITIVE
        compare("%s %s ranks=%d" % (label, a[0], ranks), g, w, exact=exact, bound=None if exact else bounds[i])
        if info.long_windows == 0 and info1.long_windows == 0:
            same_bits("%s %s ranks=%d vs one device" % (label, a[0], ranks), g, one[i])
    return ranks


@pytest.mark.parametrize("seed", range(int(os.environ.get("BOW_FUZZ_SEEDS", "64")) // 4))
def test_fuzz_cases_through_the_fan_out(seed):
    """test_fuzz_aggregate's seeded cases (row counts around the tile sizes, every timestamp pattern incl. rows below the first window
    start, Arrow offsets, nulls 0 - 100 %, NaN / Inf, Factor chains, inclusive windows, host and device residency) as ONE call over 2 - 8
    ranks of a few hundred rows each - windows straddle every boundary, some span several ranks."""
    rng = np.random.default_rng(9000 + seed)
    served = 0
    for ccols, ocols, n, interval, aggs, offset, inclusive, label in aggregate_cases(seed):
        if n < 2:
            continue
        k = int([2, 3, 4, 8][int(rng.integers(0, 4))])
        min_rows = max(1, n // (k + int(rng.integers(0, 3))))
        has_mode = any(a[0] == "Mode" for a in aggs)
        ranks = check_case(ccols, ocols, interval, aggs, offset, inclusive, label, [0] * k, min_rows)
        assert not (has_mode and ranks > 1), label            # Mode has no constant-size partial state: the one-device path
        served += ranks > 1
        if rng.random() < 0.3 and not has_mode:               # the same under strict_order: row order across the boundaries, or declined -> one device
            try:
                check_case(ccols, ocols, interval, aggs, offset, inclusive, label + " strict", [0] * k, min_rows, strict=True)
            except capi.BowGpuError as e:
                assert e.code == -9 and "2^20" in e.message, (label, e.message)
    assert served >= 10, served


def frame(n, mode, seed, nulls=0.3, int_values=False):
    rng = np.random.default_rng(seed)
    if mode == "dense":
        ts = np.arange(n, dtype=np.int64)
    elif mode == "irregular":
        ts = np.cumsum(rng.integers(1, 20, n)).astype(np.int64)
    elif mode == "negative":
        ts = (np.cumsum(rng.integers(1, 6, n)) - 3 * n).astype(np.int64)
    else:
        step = rng.integers(1, 5, n)
        step[rng.random(n) < 0.002] = rng.integers(100, 3000)
        ts = np.cumsum(step).astype(np.int64)
    vals = (rng.integers(-10 ** 9, 10 ** 9, n).astype(np.int64) if int_values else rng.standard_normal(n) * 100)
    valid = rng.random(n) >= nulls if nulls > 0 else None
    return ts, vals, valid


def columns(ts, vals, valid, residency):
    bm = None if valid is None else np.packbits(valid, bitorder="little")
    typ = capi.INT64 if vals.dtype == np.int64 else capi.FLOAT64
    if residency == capi.HOST_PINNED:
        t = capi.page_aligned(len(ts), np.int64); t[:] = ts
        v = capi.page_aligned(len(vals), vals.dtype); v[:] = vals
        b = None
        if bm is not None:
            b = capi.page_aligned(len(bm), np.uint8); b[:] = bm
        ccols = [capi.Column(t, None, capi.INT64).pin(), capi.Column(v, b, typ, 0, len(vals), -1).pin()]
    else:
        ccols = [capi.Column(ts, None, capi.INT64), capi.Column(vals, bm, typ, 0, len(vals), -1)]
        if residency == capi.DEVICE:
            ccols = [c.to_device() for c in ccols]
    ocols = [orc.Column(ts, None, orc.INT64), orc.Column(vals, bm, typ)]
    return ccols, ocols


PLAIN = [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1), ("Count", 1), ("First", 1), ("Last", 1), ("NumRows", 1)]
TW = [("WindowStart", 0), ("IntegralStep", 1), ("WeightedAverageStep", 1), ("IntegralTrapezoid", 1), ("WeightedAverageLinear", 1),
      ("ArithmeticMean", 1, [0.5, -2.0]), ("Count", 1), ("Last", 1)]


@pytest.mark.parametrize("residency", [capi.HOST, capi.HOST_PINNED, capi.DEVICE], ids=["host", "pinned", "device"])
@pytest.mark.parametrize("mode", ["dense", "irregular", "negative", "gappy"])
def test_one_call_over_four_and_eight_ranks(mode, residency):
    """every residency x timestamp pattern (incl. rows below the first window start: the protocol's second round) x plain / time-weighted
    (inclusive) reducer sets x window lengths from 7 rows to longer than a rank"""
    n = 150_000
    ts, vals, valid = frame(n, mode, 11)
    ccols, ocols = columns(ts, vals, valid, residency)
    try:
        for aggs, inclusive in [(PLAIN, False), (TW, True), (PLAIN, True)]:
            for interval, offset, k in [(7, 0, 4), (100, 13, 8), (3_000, -5, 4), (90_000, 1, 8)]:
                label = "%s I=%d k=%d" % (mode, interval, k)
                check_case(ccols, ocols, interval, aggs, offset, inclusive, label, [0] * k, 1000, out_residency=residency, expect_ranks=k)
    finally:
        for c in ccols:
            c.unpin()


def test_int64_values_no_nulls_and_strict_order():
    n = 120_000
    ts, vals, _ = frame(n, "irregular", 3, nulls=0, int_values=True)
    ccols, ocols = columns(ts, vals, None, capi.HOST)
    for interval, k in [(10, 3), (64, 8), (5_000, 4)]:
        check_case(ccols, ocols, interval, PLAIN, 3, False, "int64 I=%d" % interval, [0] * k, 500, expect_ranks=k)
        check_case(ccols, ocols, interval, PLAIN + [("IntegralStep", 1)], 3, False, "int64 strict I=%d" % interval, [0] * k, 500, strict=True, expect_ranks=k)


def test_the_planned_entry_point_fans_out_too():
    n = 80_000
    ts, vals, valid = frame(n, "irregular", 5)
    ccols, ocols = columns(ts, vals, valid, capi.HOST)
    plan = capi.plan_windows_ex(ccols[0], 50, 7)
    exp, _ = orc.aggregate(ocols, 0, 50, PLAIN, offset=7)
    with capi.devices([0, 0, 0], min_rows=100):
        outs, info = capi.rolling_aggregate(ccols, 0, 50, PLAIN, offset=7, plan=plan)
        assert capi.last_call_ranks() == 3
        for a, g, w in zip(PLAIN, outs, exp):
            compare("planned " + a[0], g, w)
        # a plan made for another column: the same refusal as on one device
        other = capi.plan_windows_ex(capi.Column(ts + 1000, None, capi.INT64), 50, 7)
        with pytest.raises(capi.BowGpuError) as e:
            capi.rolling_aggregate(ccols, 0, 50, PLAIN, offset=7, plan=other)
        assert e.value.code == -10


def test_what_the_fan_out_leaves_to_the_one_device_path():
    n = 50_000
    ts, vals, valid = frame(n, "irregular", 8)
    ccols, ocols = columns(ts, vals, valid, capi.HOST)
    with capi.devices([0, 0], min_rows=100):
        # Mode: not mergeable
        aggs = [("WindowStart", 0), ("Mode", 1)]
        outs, _ = capi.rolling_aggregate(ccols, 0, 100, aggs)
        assert capi.last_call_ranks() == 1
        exp, _ = orc.aggregate(ocols, 0, 100, aggs)
        for a, g, w in zip(aggs, outs, exp):
            compare("mode " + a[0], g, w)
        # an interval column with nulls
        tv = np.ones(n, bool); tv[5::97] = False
        nts = [capi.Column(ts, np.packbits(tv, bitorder="little"), capi.INT64, 0, n, -1), ccols[1]]
        onts = [orc.Column(ts, np.packbits(tv, bitorder="little"), orc.INT64), ocols[1]]
        outs, _ = capi.rolling_aggregate(nts, 0, 100, PLAIN)
        assert capi.last_call_ranks() == 1
        exp, _ = orc.aggregate(onts, 0, 100, PLAIN)
        for a, g, w in zip(PLAIN, outs, exp):
            compare("null ts " + a[0], g, w)
        # too few rows for two ranks
        capi.set_devices([0, 0], min_rows=n)
        capi.rolling_aggregate(ccols, 0, 100, PLAIN)
        assert capi.last_call_ranks() == 1
    assert capi.get_devices() == []


def test_errors_are_the_one_device_call_s():
    n = 40_000
    ts, vals, valid = frame(n, "irregular", 9)
    bad = ts.copy()
    bad[n // 2 + 1] = bad[n // 2] - 50          # not ascending, exactly where two ranks meet ...
    bad2 = ts.copy()
    bad2[1234] = bad2[1233] - 50                # ... and inside a rank
    with capi.devices([0, 0], min_rows=100):
        for t in (bad, bad2):
            ccols, _ = columns(t, vals, valid, capi.HOST)
            with pytest.raises(capi.BowGpuError) as e:
                capi.rolling_aggregate(ccols, 0, 10, PLAIN)
            assert e.value.code == -14, e.value
        ccols, _ = columns(ts, vals, valid, capi.HOST)
        W = capi.plan_windows(ccols[0], 10, 0)[1]
        with pytest.raises(capi.BowGpuError) as e:
            capi.rolling_aggregate(ccols, 0, 10, PLAIN, outs=[capi.OutColumn(W - 1) for _ in PLAIN])
        assert e.value.code == -10
        with pytest.raises(capi.BowGpuError) as e:   # validateAggregation's error first (aggregation.go:163-166)
            capi.rolling_aggregate(ccols, 0, 10, [("Sum", 1)])
        assert e.value.code == -5
    with pytest.raises(capi.BowGpuError) as e:
        capi.set_devices([0, capi.device_count()])
    assert e.value.code == -11
    assert capi.get_devices() == []


def test_fanned_out_calls_from_several_threads_are_serialised_and_right():
    n = 200_000
    ts, vals, valid = frame(n, "irregular", 21)
    ccols, ocols = columns(ts, vals, valid, capi.HOST)
    exp, _ = orc.aggregate(ocols, 0, 25, PLAIN)
    errors = []

    def work(tid):
        try:
            for _ in range(5):
                outs, _ = capi.rolling_aggregate(ccols, 0, 25, PLAIN)
                assert capi.last_call_ranks() == 4
                for a, g, w in zip(PLAIN, outs, exp):
                    compare("thread %d %s" % (tid, a[0]), g, w)
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    with capi.devices([0] * 4, min_rows=1000):
        threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    assert not errors, errors[0]


def test_ten_million_rows_registered_host_memory_over_eight_ranks():
    """the shape the path is for: a Bow in registered host memory, every rank reading ITS row range in place over the host link;
    every window against the one-device call (bit for bit: 10-row windows are walked in row order on both) and the oracle"""
    n = 10_000_000
    rng = np.random.default_rng(1)
    ts = capi.page_aligned(n, np.int64); ts[:] = np.arange(n) * 3 + rng.integers(0, 3, n)
    vals = capi.page_aligned(n, np.float64); vals[:] = rng.random(n)
    valid = rng.random(n) >= 0.3
    bm = capi.page_aligned((n + 7) // 8, np.uint8); bm[:] = np.packbits(valid, bitorder="little")
    ccols = [capi.Column(ts, None, capi.INT64).pin(), capi.Column(vals, bm, capi.FLOAT64, 0, n, int((~valid).sum())).pin()]
    ocols = [orc.Column(ts, None, orc.INT64), orc.Column(vals, bm, orc.FLOAT64)]
    aggs = [("WindowStart", 0), ("ArithmeticMean", 1), ("Min", 1), ("Count", 1)]
    try:
        check_case(ccols, ocols, 30, aggs, 0, False, "1e7 pinned", [0] * 8, 1 << 20, expect_ranks=8)
    finally:
        for c in ccols:
            c.unpin()


# ------------------------------------------------------------------ r.Interpolate(...).Aggregate(...) as ONE call over N ranks
def check_pipeline(ccols, ocols, interval, interps, aggs, offset, ids, min_rows, label, strict=False, expect_ranks=None, out_residency=capi.HOST):
    """bowgpu_rolling_interpolate_aggregate with the fan-out on, against oracle interpolate -> oracle aggregate and against the same
    call on one device (bit for bit: the ranks' interpolated rows concatenated ARE the unsharded Interpolate; windows of these tests are
    reduced in row order on both)"""
    mid = orc.interpolate(ocols, 0, interval, interps, offset=offset)
    want, nic = orc.aggregate(mid, 0, interval, aggs, offset=offset)
    one, info1 = capi.rolling_interpolate_aggregate(ccols, 0, interval, interps, aggs, offset=offset, strict_order=strict, out_residency=out_residency)
    assert capi.last_call_ranks() == 1
    with capi.devices(ids, min_rows=min_rows):
        got, info = capi.rolling_interpolate_aggregate(ccols, 0, interval, interps, aggs, offset=offset, strict_order=strict, out_residency=out_residency)
        ranks = capi.last_call_ranks()
    if expect_ranks is not None:
        assert ranks == expect_ranks, (label, ranks)
    assert (info.s0, info.num_windows, info.new_interval_col) == (info1.s0, info1.num_windows, nic), label
    for a, g, w, o in zip(aggs, got, want, one):
        if info.long_windows == 0:
            compare("%s %s ranks=%d" % (label, a[0], ranks), g, w)
            if info1.long_windows == 0:
                same_bits("%s %s vs one device" % (label, a[0]), g, o)
    return ranks


@pytest.mark.parametrize("residency", [capi.HOST, capi.HOST_PINNED, capi.DEVICE], ids=["host", "pinned", "device"])
@pytest.mark.parametrize("kind", ["Linear", "StepPrevious", "None"])
def test_interpolate_then_aggregate_as_one_call_over_ranks(kind, residency):
    """configs[2]'s pipeline through the fan-out: every rank interpolates ITS rows (its neighbours' nearest valid points reach it
    through the host-memory exchange - long runs of nulls put them several ranks away), then the ranks aggregate the interpolated
    rows; Float64 and Int64 columns, PrevRow, a Factor chain"""
    n = 120_000
    rng = np.random.default_rng(31)
    ts = (np.cumsum(rng.integers(1, 20, n)) + 500).astype(np.int64)
    a = np.round(rng.standard_normal(n) * 100, 2)
    va = rng.random(n) >= 0.3
    va[31_000:36_000] = False                  # runs of nulls across the rank boundaries (32 768 for 4 and 8 ranks, 40 960 for 3): a rank's nearest
    va[39_000:43_000] = False                  # valid point lies on its neighbour (the oracle's neighbour walks bound the run lengths: quadratic)
    b = rng.integers(-1000, 1000, n).astype(np.int64)
    vb = rng.random(n) >= 0.5
    vb[:6_000] = False                         # nothing valid to the left of the first windows: Options.PrevRow serves
    prev = (float(ts[0] - 3), True, 42.5, True, 42)
    ip = [{"kind": "WindowStart", "col": 0}, {"kind": kind, "col": 1}, {"kind": kind, "col": 2, "prev": prev}]
    aggs = [("WindowStart", 0), ("ArithmeticMean", 1), ("Sum", 2, [0.5]), ("Min", 1), ("Max", 2), ("Count", 1), ("First", 2), ("Last", 1), ("NumRows", 0)]
    bma, bmb = np.packbits(va, bitorder="little"), np.packbits(vb, bitorder="little")
    ocols = [orc.Column(ts, None, orc.INT64), orc.Column(a, bma, orc.FLOAT64), orc.Column(b, bmb, orc.INT64)]
    if residency == capi.HOST_PINNED:
        bufs = []
        for arr in (ts, a, b, bma, bmb):
            p = capi.page_aligned(len(arr), arr.dtype); p[:] = arr; bufs.append(p)
        ccols = [capi.Column(bufs[0], None, capi.INT64).pin(), capi.Column(bufs[1], bufs[3], capi.FLOAT64, 0, n, -1).pin(),
                 capi.Column(bufs[2], bufs[4], capi.INT64, 0, n, -1).pin()]
    else:
        ccols = [capi.Column(ts, None, capi.INT64), capi.Column(a, bma, capi.FLOAT64, 0, n, -1), capi.Column(b, bmb, capi.INT64, 0, n, -1)]
        if residency == capi.DEVICE:
            ccols = [c.to_device() for c in ccols]
    try:
        for interval, offset, k in [(100, 0, 4), (64, 7, 8), (1000, 3, 3)]:
            check_pipeline(ccols, ocols, interval, ip, aggs, offset, [0] * k, 2000, "%s I=%d k=%d" % (kind, interval, k), expect_ranks=k,
                           out_residency=residency)
    finally:
        for c in ccols:
            c.unpin()


def test_pipeline_shapes_the_fan_out_leaves_to_one_device():
    """rows below the first window start (the interpolated frame then starts with its synthetic row at s0 and goes on below it), more than 8
    columns: the same call, served by the calling thread's device.  Negative window starts - the window that starts at -1 among them - are
    served by the ranks since round 6."""
    n = 40_000
    rng = np.random.default_rng(4)
    ts = (np.cumsum(rng.integers(1, 9, n)) - 1000).astype(np.int64)       # starts below zero
    v = rng.standard_normal(n)
    valid = rng.random(n) >= 0.3
    bm = np.packbits(valid, bitorder="little")
    ccols = [capi.Column(ts, None, capi.INT64), capi.Column(v, bm, capi.FLOAT64, 0, n, -1)]
    ocols = [orc.Column(ts, None, orc.INT64), orc.Column(v, bm, orc.FLOAT64)]
    ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
    aggs = [("WindowStart", 0), ("ArithmeticMean", 1), ("Count", 1)]
    assert check_pipeline(ccols, ocols, 50, ip, aggs, 0, [0, 0, 0], 500, "negative starts") == 3
    tsm = (ts - ts[0] - 951).astype(np.int64)                                # first row ON the first window start -951 = -1 - 19 * 50 (offset 49)
    c1 = [capi.Column(tsm, None, capi.INT64), ccols[1]]
    o1 = [orc.Column(tsm, None, orc.INT64), ocols[1]]
    assert orc.plan_windows(o1[0], 50, 49)[0] == tsm[0] == -951
    assert check_pipeline(c1, o1, 50, ip, aggs, 49, [0, 0, 0, 0], 500, "a window that starts at -1") == 4
    keep = (tsm < -1) | (tsm >= 49)                                          # ... and the same with that window empty: no synthetic row for it
    c2 = [capi.Column(tsm[keep].copy(), None, capi.INT64), capi.Column(v[keep].copy(), np.packbits(valid[keep], bitorder="little"), capi.FLOAT64, 0, int(keep.sum()), -1)]
    o2 = [orc.Column(tsm[keep].copy(), None, orc.INT64), orc.Column(v[keep].copy(), np.packbits(valid[keep], bitorder="little"), orc.FLOAT64)]
    assert check_pipeline(c2, o2, 50, ip, aggs, 49, [0, 0, 0], 500, "the window that starts at -1, empty") == 3
    ts3 = np.concatenate([np.array([-15, -12], dtype=np.int64), ts + 1100])  # Go's truncating division: s0 = -11 above the first two rows
    v3, valid3 = np.concatenate([[1.5, 2.5], v]), np.concatenate([[True, True], valid])
    bm3 = np.packbits(valid3, bitorder="little")
    c3 = [capi.Column(ts3, None, capi.INT64), capi.Column(v3, bm3, capi.FLOAT64, 0, n + 2, -1)]
    o3 = [orc.Column(ts3, None, orc.INT64), orc.Column(v3, bm3, orc.FLOAT64)]
    assert orc.plan_windows(o3[0], 10, 9)[0] > ts3[0]
    # (the pipeline is declined - one device interpolates - and the Aggregate over the interpolated frame, an ordinary frame in that device's
    # memory, is what the three ranks of THIS list serve: a list of one device throughout shares device-resident buffers)
    assert check_pipeline(c3, o3, 10, ip, aggs, 9, [0, 0, 0], 500, "rows below s0 that belong to no window") == 3
    # ... and rows below s0 that ride in window 0 (a row at or above s0 inside its first interval): the interpolated frame starts with its
    # synthetic row at s0 and goes on BELOW it - Aggregate refuses it as not ascending, with the device list in force exactly as without
    assert orc.plan_windows(ocols[0], 50, 49)[0] > ts[0]
    for ids in ([0], [0, 0, 0]):
        with capi.devices(ids, min_rows=500):
            with pytest.raises(capi.BowGpuError) as e:
                capi.rolling_interpolate_aggregate(ccols, 0, 50, ip, aggs, offset=49)
            assert e.value.code == -14 and capi.last_call_ranks() == 1
    # nine columns
    ts2 = (ts + 5000).astype(np.int64)
    cc = [capi.Column(ts2, None, capi.INT64)] + [capi.Column(v * (i + 1), bm, capi.FLOAT64, 0, n, -1) for i in range(8)]
    oc = [orc.Column(ts2, None, orc.INT64)] + [orc.Column(v * (i + 1), bm, orc.FLOAT64) for i in range(8)]
    ip9 = [{"kind": "WindowStart", "col": 0}] + [{"kind": "Linear", "col": i + 1} for i in range(8)]
    assert check_pipeline(cc, oc, 50, ip9, [("WindowStart", 0), ("ArithmeticMean", 8)], 0, [0, 0], 500, "nine columns") == 1
    # ... and eight are served by the ranks
    assert check_pipeline(cc[:8], oc[:8], 50, ip9[:8], [("WindowStart", 0), ("ArithmeticMean", 7)], 0, [0, 0], 500, "eight columns") == 2
