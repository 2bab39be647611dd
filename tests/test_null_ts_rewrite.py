"""The device path for an interval column with nulls (bow_amd/csrc/ts_nulls.hip) rewrites the call onto a dense interval column:
timestamps forward-filled, value validity ANDed with "the row belongs to a window" (and with the interval column's validity for the
time-weighted reducers).  CPU check of that REWRITE RULE against the oracle's literal window walk (rolling.go:177-239 restated in
oracle/bow_oracle.c): the oracle on the rewritten, dense frame must equal the oracle on the original frame with its null
timestamps - for every reducer the device path serves.  (The GPU tests compare the kernels with the oracle directly.)"""
import numpy as np
import pytest

from oracle import pyoracle as orc

PLAIN = ["Sum", "ArithmeticMean", "Min", "Max", "Count", "First", "Last"]
TW = ["IntegralStep", "WeightedAverageStep"]
LINEAR = ["IntegralTrapezoid", "WeightedAverageLinear"]   # the two reducers that ask for inclusive windows (aggregation.go:183-185)


def go_div(a, b):
    q = abs(a) // b
    return -q if a < 0 else q


def rewrite(ts, tvalid, s0, interval):
    """ts_eff, keep - what ts_nullfill_kernel computes"""
    n = len(ts)
    idx = np.arange(n)
    prev = np.maximum.accumulate(np.where(tvalid, idx, -1))
    nxt = np.minimum.accumulate(np.where(tvalid, idx, n)[::-1])[::-1]
    ts_eff = ts[np.maximum(prev, 0)].copy()
    wid = lambda t: np.where(t < s0, 0, (t - s0) // interval)      # rows below s0 ride in window 0
    keep = tvalid.copy()
    nul = ~tvalid
    ok = nul & (prev >= 0) & (nxt < n)
    keep[ok] = wid(ts[prev[ok]]) == wid(ts[np.minimum(nxt[ok], n - 1)])
    return ts_eff, keep


@pytest.mark.parametrize("null_frac", [0.03, 0.3, 0.8])
def test_the_rewritten_dense_frame_gives_the_oracles_answer(null_frac):
    rng = np.random.default_rng(int(null_frac * 1000))
    for case in range(60):
        n = int(rng.integers(1, 400))
        ts = np.cumsum(rng.integers(0, 7, n)).astype(np.int64) + int(rng.integers(-60, 60))
        tvalid = rng.random(n) >= null_frac
        tvalid[0] = tvalid[-1] = True
        vals = np.round(rng.standard_normal(n) * 10, 2)
        vvalid = rng.random(n) >= 0.25
        interval = int(rng.choice([1, 2, 5, 10, 40, 1000]))
        offset = int(rng.integers(-interval, 2 * interval))
        tbm, vbm = np.packbits(tvalid, bitorder="little"), np.packbits(vvalid, bitorder="little")
        ocols = [orc.Column(ts, tbm, orc.INT64), orc.Column(vals, vbm, orc.FLOAT64)]
        try:
            s0, W = orc.plan_windows(ocols[0], interval, offset)
        except orc.OracleError:
            continue
        aggs = [("WindowStart", 0)] + [(k, 1) for k in PLAIN + TW] + [("Sum", 0), ("Last", 0), ("IntegralStep", 0)]
        want, _ = orc.aggregate(ocols, 0, interval, aggs, offset=offset)
        ts_eff, keep = rewrite(ts, tvalid, s0, interval)
        assert (np.diff(ts_eff) >= 0).all()
        # columns of the rewritten call: [ts_eff, val & keep, val & tsvalid, ts & tsvalid(& keep), ts & tsvalid]
        dense = [orc.Column(ts_eff, None, orc.INT64),
                 orc.Column(vals, np.packbits(vvalid & keep, bitorder="little"), orc.FLOAT64),
                 orc.Column(vals, np.packbits(vvalid & tvalid, bitorder="little"), orc.FLOAT64),
                 orc.Column(ts, np.packbits(tvalid & keep, bitorder="little"), orc.INT64),
                 orc.Column(ts, np.packbits(tvalid, bitorder="little"), orc.INT64)]
        aggs2 = [("WindowStart", 0)] + [(k, 1) for k in PLAIN] + [(k, 2) for k in TW] + [("Sum", 3), ("Last", 3), ("IntegralStep", 4)]
        got, _ = orc.aggregate(dense, 0, interval, aggs2, offset=offset)
        label = (case, n, interval, offset)
        for (k, _c), g, w in zip(aggs, got, want):
            assert g.length == w.length == W, (label, k)
            gm, wm = g.valid_mask(), w.valid_mask()
            assert np.array_equal(gm, wm), (label, k, list(ts), list(tvalid.astype(int)))
            gv, wv = g.values[:W].view(np.uint64), w.values[:W].view(np.uint64)
            assert np.array_equal(gv[gm], wv[wm]), (label, k)


def test_a_null_last_timestamp_produces_no_window_at_all():
    ts = np.array([10, 11, 20, 21, 30], dtype=np.int64)
    vals = np.arange(5.0)
    for tail in (1, 2):
        tvalid = np.ones(5, bool)
        tvalid[-tail:] = False
        cols = [orc.Column(ts, np.packbits(tvalid, bitorder="little"), orc.INT64), orc.Column(vals, None, orc.FLOAT64)]
        s0, W = orc.plan_windows(cols[0], 10, 0)
        assert (s0, W) == (10, 2)             # countWindows measures from the last VALID timestamp (rolling.go:143-154)
        out, _ = orc.aggregate(cols, 0, 10, [("WindowStart", 0), ("Sum", 1), ("Count", 1)])
        assert all(o.length == W and not o.valid_mask().any() for o in out)   # ... and HasNext (:162-173) never lets a window start


def rewrite_inclusive(ts, tvalid, s0, interval):
    """ts_eff, keep, quirk - what ts_nullfill_kernel computes for an INCLUSIVE iteration (rolling.go:201-218).  quirk: a row on a window
    start (not window 0's), the first with that timestamp, with a null timestamp right behind it - the next window starts at
    `rowIndex - 1`, the last of the null rows, without this row."""
    n = len(ts)
    idx = np.arange(n)
    prev = np.maximum.accumulate(np.where(tvalid, idx, -1))
    nxt = np.minimum.accumulate(np.where(tvalid, idx, n)[::-1])[::-1]
    ts_eff = ts[np.maximum(prev, 0)].copy()
    wid = lambda t: 0 if t < s0 else (t - s0) // interval      # noqa: E731
    quirk = np.zeros(n, bool)
    for i in range(n - 1):
        if not tvalid[i] or tvalid[i + 1]:
            continue
        t = ts[i]
        if t < s0 + interval or (t - s0) % interval:
            continue
        p = prev[i - 1] if i > 0 else -1
        if p >= 0 and ts[p] == t:
            continue
        quirk[i] = True
    keep = tvalid & ~quirk
    for j in range(n):
        if tvalid[j]:
            continue
        p, q = prev[j], nxt[j]
        if p < 0 or q >= n:
            continue
        tp, tq = ts[p], ts[q]
        k = wid(tp) == wid(tq) or tq == s0 + (wid(tp) + 1) * interval      # ... or q is the row ON the end of p's window
        if quirk[p]:
            k = k and j == q - 1                                             # only the last null row opens the next window's slice
        keep[j] = k
    return ts_eff, keep, quirk


def test_the_inclusive_rewrite_gives_the_oracles_answer():
    """every reducer that reads windows through UnsetInclusive (window.go:23-31) - incl. NumRows as Count over the keep bits - is
    exact on the rewritten frame everywhere; the two that need inclusive windows are exact except in the windows BEHIND a quirk row,
    which the device path recomputes by a walk (ts_quirk_fix_kernel; the GPU tests compare those with the oracle)"""
    rng = np.random.default_rng(7)
    with_quirks = 0
    for case in range(700):
        null_frac = rng.choice([0.03, 0.3, 0.6])
        n = int(rng.integers(1, 120))
        ts = np.cumsum(rng.integers(0, 7, n)).astype(np.int64) + int(rng.integers(-60, 60))
        tvalid = rng.random(n) >= null_frac
        tvalid[0] = tvalid[-1] = True
        vals = np.round(rng.standard_normal(n) * 10, 2)
        vvalid = rng.random(n) >= 0.25
        interval = int(rng.choice([1, 2, 5, 10, 40]))
        offset = int(rng.integers(-interval, 2 * interval))
        ocols = [orc.Column(ts, np.packbits(tvalid, bitorder="little"), orc.INT64), orc.Column(vals, np.packbits(vvalid, bitorder="little"), orc.FLOAT64)]
        try:
            s0, W = orc.plan_windows(ocols[0], interval, offset)
        except orc.OracleError:
            continue
        aggs = [("WindowStart", 0), ("NumRows", 1)] + [(k, 1) for k in PLAIN + TW + LINEAR] + [("Sum", 0), ("Last", 0), ("IntegralStep", 0)]
        want, _ = orc.aggregate(ocols, 0, interval, aggs, offset=offset, inclusive=bool(case & 1))   # (LINEAR implies it anyway)
        ts_eff, keep, quirk = rewrite_inclusive(ts, tvalid, s0, interval)
        with_quirks += bool(quirk.any())
        behind = {int((ts[i] - s0) // interval) for i in np.nonzero(quirk)[0]}
        pack = lambda m: np.packbits(m, bitorder="little")      # noqa: E731
        dense = [orc.Column(ts_eff, None, orc.INT64),
                 orc.Column(vals, pack(vvalid & keep), orc.FLOAT64),                 # 1 plain
                 orc.Column(vals, pack(vvalid & tvalid & ~quirk), orc.FLOAT64),      # 2 step integrals
                 orc.Column(ts, pack(tvalid & keep), orc.INT64),                     # 3 plain, the interval column itself
                 orc.Column(ts, pack(tvalid & ~quirk), orc.INT64),                   # 4 step, the interval column itself
                 orc.Column(vals, pack(keep), orc.FLOAT64),                          # 5 NumRows
                 orc.Column(vals, pack(vvalid & tvalid), orc.FLOAT64)]               # 6 the two inclusive reducers
        aggs2 = ([("WindowStart", 0), ("Count", 5)] + [(k, 1) for k in PLAIN] + [(k, 2) for k in TW] + [(k, 6) for k in LINEAR] +
                 [("Sum", 3), ("Last", 3), ("IntegralStep", 4)])
        got, _ = orc.aggregate(dense, 0, interval, aggs2, offset=offset, inclusive=True)
        for (k, _c), g, w in zip(aggs, got, want):
            gm, wm = g.valid_mask().copy(), w.valid_mask().copy()
            gv, wv = g.values[:W].copy(), w.values[:W].copy()
            if k == "NumRows":
                gv = gv.view(np.int64).astype(np.float64)
            if k in LINEAR:
                for f in behind:
                    if f < W:
                        gm[f] = wm[f] = False
            label = (case, k, list(ts), list(tvalid.astype(int)), interval, offset)
            assert np.array_equal(gm, wm), label
            assert np.array_equal(gv.view(np.uint64)[gm], wv.view(np.uint64)[wm]), label
    assert with_quirks > 100


def test_interpolate_on_the_kept_rows_plus_a_marker_column_gives_the_oracles_answer():
    """Rolling.Interpolate over an interval column with nulls (exclusive iteration) as extras.cpp interp_null_ts makes it: the kept
    rows compacted (timestamps forward-filled, value validity ANDed with the interval column's), one more Int64 column under
    interpolation.None that is valid exactly in the null-timestamp rows and holds their row number; in the output, wherever that
    column is valid the row is a copy of such a row: its timestamp becomes null, its value gets its own validity back."""
    rng = np.random.default_rng(11)
    pack = lambda m: np.packbits(m, bitorder="little")      # noqa: E731
    for case in range(600):
        null_frac = rng.choice([0.03, 0.3, 0.6])
        n = int(rng.integers(1, 100))
        ts = np.cumsum(rng.integers(0, 7, n)).astype(np.int64) + int(rng.integers(-60, 60))
        tvalid = rng.random(n) >= null_frac
        tvalid[0] = tvalid[-1] = True
        vals = np.round(rng.standard_normal(n) * 10, 2)
        vvalid = rng.random(n) >= 0.25
        interval = int(rng.choice([1, 2, 5, 10, 40]))
        offset = int(rng.integers(-interval, 2 * interval))
        ocols = [orc.Column(ts, pack(tvalid), orc.INT64), orc.Column(vals, pack(vvalid), orc.FLOAT64)]
        try:
            s0, _W = orc.plan_windows(ocols[0], interval, offset)
        except orc.OracleError:
            continue
        kind = ["Linear", "StepPrevious", "None"][case % 3]
        ip = [{"kind": "WindowStart", "col": 0}, {"kind": kind, "col": 1}]
        if case % 2:
            ip[1]["prev"] = (float(ts[0] - 3), True, 42.5, True, 42)
        want = orc.interpolate(ocols, 0, interval, ip, offset=offset)
        ts_eff, keep = rewrite(ts, tvalid, s0, interval)
        k = np.nonzero(keep)[0]
        cts, cv, ctv, cvv = ts_eff[k], vals[k], tvalid[k], vvalid[k]
        dense = [orc.Column(cts, None, orc.INT64), orc.Column(cv, pack(cvv & ctv), orc.FLOAT64),
                 orc.Column(np.arange(len(k), dtype=np.int64), pack(~ctv), orc.INT64)]
        got = orc.interpolate(dense, 0, interval, ip + [{"kind": "None", "col": 2}], offset=offset)
        m_out = got[0].length
        gts_m, gv_m = got[0].valid_mask().copy(), got[1].valid_mask().copy()
        gv = got[1].values[:m_out].copy()
        for j in np.nonzero(got[2].valid_mask())[0]:
            r = got[2].values[j]
            gts_m[j] = False
            gv_m[j] = cvv[r]
            if cvv[r]:
                gv[j] = cv[r]
        label = (case, list(ts), list(tvalid.astype(int)), list(vvalid.astype(int)), interval, offset, kind)
        assert m_out == want[0].length, label
        wm0, wm1 = want[0].valid_mask(), want[1].valid_mask()
        assert np.array_equal(gts_m, wm0) and np.array_equal(gv_m, wm1), label
        assert np.array_equal(got[0].values[:m_out].view(np.uint64)[gts_m], want[0].values[:m_out].view(np.uint64)[wm0]), label
        assert np.array_equal(gv.view(np.uint64)[gv_m], want[1].values[:m_out].view(np.uint64)[wm1]), label


def _synth(kind, sk, pp, np_, prevrow):
    # pp / np_: (t, v) or None ; float64 column
    if kind == "None": return None
    if kind == "StepPrevious":
        if pp is not None: return pp[1]
        if prevrow is not None and prevrow[3]: return prevrow[2]
        return None
    if kind == "Linear":
        if pp is not None: t0, v0 = float(pp[0]), pp[1]
        elif prevrow is not None and prevrow[1] and prevrow[3]: t0, v0 = prevrow[0], prevrow[2]
        else: return None
        if np_ is None: return None
        t2, v2 = float(np_[0]), np_[1]
        with np.errstate(all='ignore'):
            coef = (np.float64(sk) - np.float64(t0)) / (np.float64(t2) - np.float64(t0))
            return float((np.float64(v2) - np.float64(v0)) * coef + np.float64(v0))


def test_inclusive_interpolate_on_the_kept_rows_gives_the_oracles_answer():
    """Rolling.Interpolate over an interval column with nulls after an INCLUSIVE iteration, as extras.cpp interp_null_ts makes it: the
    kept rows - the inclusive keep rule, the rows on a window start with a null timestamp behind them included - compacted, the
    marker column valid in the null-timestamp rows AND in those rows; the ordinary inclusive call then copies such a row twice (the
    end of its window, the start of the next); the second copy stands where the reference has the next window's synthetic start row
    (that window begins at the last null row, without the row on its start: rolling.go:214-218) and is replaced by the
    interpolators' values there.  Two shapes are declined: an equal timestamp right behind the null rows (the next window then has
    its start and adds no row), and such a row on -1 (interpolation.go:119-127's "no first value")."""
    pack = lambda m: np.packbits(m, bitorder="little")      # noqa: E731
    synth = _synth
    nq = 0
    rng = np.random.default_rng(21)
    for case in range(600):
        null_frac = rng.choice([0.03, 0.3, 0.6])
        n = int(rng.integers(1, 100))
        ts = np.cumsum(rng.integers(0, 7, n)).astype(np.int64) + int(rng.integers(-60, 60))
        tvalid = rng.random(n) >= null_frac
        tvalid[0] = tvalid[-1] = True
        vals = np.round(rng.standard_normal(n) * 10, 2)
        vvalid = rng.random(n) >= 0.25
        interval = int(rng.choice([1, 2, 5, 10, 40]))
        offset = int(rng.integers(-interval, 2 * interval))
        ocols = [orc.Column(ts, pack(tvalid), orc.INT64), orc.Column(vals, pack(vvalid), orc.FLOAT64)]
        try:
            s0, W = orc.plan_windows(ocols[0], interval, offset)
        except orc.OracleError:
            continue
        kind = ["Linear", "StepPrevious", "None"][case % 3]
        ip = [{"kind": "WindowStart", "col": 0}, {"kind": kind, "col": 1}]
        prevrow = None
        if case % 2:
            prevrow = (float(ts[0] - 3), True, 42.5, True, 42)
            ip[1]["prev"] = prevrow
        want = orc.interpolate(ocols, 0, interval, ip, offset=offset, inclusive=True)
        ts_eff, keep, quirk = rewrite_inclusive(ts, tvalid, s0, interval)
        keep = keep & ~quirk
        nq += int(quirk.any())
        outside = False                      # the two shapes interp_null_ts declines (ts_nulls.hip ts_nullfill_kernel, mode 2)
        for i in np.nonzero(quirk)[0]:
            b = i + 1
            while b < n and not tvalid[b]:
                b += 1
            if (b < n and ts[b] == ts[i]) or ts[i] == -1:
                outside = True
        if outside:
            continue
        k = np.nonzero(keep | quirk)[0]
        cts, cv, ctv, cvv, cq = ts_eff[k], vals[k], tvalid[k], vvalid[k], quirk[k]
        m = len(k)
        marker = np.arange(m, dtype=np.int64)
        dense = [orc.Column(cts, None, orc.INT64), orc.Column(cv, pack(cvv & ctv), orc.FLOAT64), orc.Column(marker, pack(~ctv | cq), orc.INT64)]
        got = orc.interpolate(dense, 0, interval, ip + [{"kind": "None", "col": 2}], offset=offset, inclusive=True)
        mo = got[0].length
        gts_m, gv_m = got[0].valid_mask().copy(), got[1].valid_mask().copy()
        gts, gv = got[0].values[:mo].copy(), got[1].values[:mo].copy()
        mm = got[2].valid_mask(); mv = got[2].values[:mo]
        both = tvalid & vvalid
        for j in np.nonzero(mm)[0]:
            r = mv[j]
            if not ctv[r]:      # copy of a null-timestamp row
                gts_m[j] = False
                gv_m[j] = cvv[r]
                if cvv[r]: gv[j] = cv[r]
            elif cq[r] and j > 0 and mm[j - 1] and mv[j - 1] == r:   # the second copy of a quirk row: the next window's synthetic start row
                i = k[r]                      # original row
                sk = ts[i]
                pi = i
                while pi >= 0 and not both[pi]: pi -= 1
                b = i + 1
                while b < n and not tvalid[b]: b += 1
                ni = b
                while ni < n and not both[ni]: ni += 1
                pp = (ts[pi], vals[pi]) if pi >= 0 else None
                nn = (ts[ni], vals[ni]) if ni < n else None
                x = synth(kind, sk, pp, nn, prevrow)
                gts[j] = sk; gts_m[j] = True
                if x is None: gv_m[j] = False
                else: gv_m[j] = True; gv[j] = x
        info = (case, [int(x) for x in ts], list(tvalid.astype(int)), list(vvalid.astype(int)), interval, offset, kind)
        assert mo == want[0].length, (mo, want[0].length, info)
        wm0, wm1 = want[0].valid_mask(), want[1].valid_mask()
        assert np.array_equal(gts_m, wm0), info
        assert np.array_equal(gv_m, wm1), (np.nonzero(gv_m != wm1)[0], info)
        assert np.array_equal(gts.view(np.uint64)[gts_m], want[0].values[:mo].view(np.uint64)[wm0]), info
        assert np.array_equal(gv.view(np.uint64)[gv_m], want[1].values[:mo].view(np.uint64)[wm1]), info

    assert nq > 100
