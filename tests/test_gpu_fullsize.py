"""BASELINE.json's full-size configurations on the GPU (configs[1..3] at 1e8 rows, the headline at 1e9 rows):
the HIP path through the C ABI, inputs generated in HBM by the counter-based generators (same generators restated
in oracle/bow_oracle.c).  Where the oracle finishes in seconds the whole output is compared with it bit for bit;
beyond that, size-independent properties: conservation of rows / counts, the arithmetic progression of window
starts, idempotence of Interpolate, sharded == whole, and oracle comparisons on random row ranges."""
import os

import numpy as np
import psutil
import pytest

from bow_amd import capi, sharded
from oracle import pyoracle as orc
from test_gpu_aggregate import compare
from test_gpu_callers import cmp_out

pytestmark = pytest.mark.gpu

N8 = 100_000_000
N9 = 1_000_000_000
# the MI355X boxes have hundreds of host cores' worth of RAM: the oracle then checks EVERY output slot even at 1e8 / 1e9 rows
# (it runs ~250 M rows/s per reducer there); on a small host the same tests fall back to sampled row ranges
HOST_RAM = psutil.virtual_memory().total
BIG_HOST = HOST_RAM > 200e9
# which branch ran is part of the evidence: it goes to the test log (pytest -rA / -s shows it; captured output is kept on failure)
# and to gpurun_out/fullsize_mode.txt on the GPU box
_MODE = "host RAM %.0f GB -> full-size outputs are checked on %s" % (HOST_RAM / 1e9, "EVERY slot against the oracle" if BIG_HOST else
                                                                     "SAMPLED row ranges (small host)")
print("[test_gpu_fullsize] " + _MODE)
try:
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fullsize_mode.txt"), "w") as _f:
        _f.write(_MODE + "\n")
except OSError:
    pass


def test_headline_check_is_not_sampled_on_a_big_host():
    """a >= 200 GB host must take the every-slot branch of the headline check (the sampled fallback exists for small dev boxes only)"""
    import warnings
    if HOST_RAM >= 200e9:
        assert BIG_HOST
    else:
        warnings.warn("small host (%.0f GB RAM): the 1e8 / 1e9-row outputs are checked on sampled row ranges only" % (HOST_RAM / 1e9))


def _paths():
    """the three tile kernels over the same call (auto = simple kernel where it applies)"""
    # (the lean / general runs also switch the long-only shortcut off, so windows of thousands of rows take both long paths)
    for label, env in (("auto", {}), ("classic-long", {"BOWGPU_LONG_CLASSIC": "1"}), ("stream-all", {"BOWGPU_LONG_STREAM_ALL": "1"}),
                       ("small-list", {"BOWGPU_SIMPLE_DENSE": "0"}), ("large-list", {"BOWGPU_SIMPLE_DENSE": "1"}),
                       ("lean", {"BOWGPU_NO_SIMPLE": "1", "BOWGPU_NO_LONG_ONLY": "1"}),
                       ("general", {"BOWGPU_FORCE_GENERAL": "1", "BOWGPU_NO_LONG_ONLY": "1"})):
        for k in ("BOWGPU_NO_SIMPLE", "BOWGPU_FORCE_GENERAL", "BOWGPU_NO_LONG_ONLY", "BOWGPU_LONG_CLASSIC", "BOWGPU_LONG_STREAM_ALL"):
            os.environ[k] = env.get(k, "0")
        os.environ.pop("BOWGPU_SIMPLE_DENSE", None)      # (the simple kernel's two head-list sizes: by the plan unless forced)
        if "BOWGPU_SIMPLE_DENSE" in env:
            os.environ["BOWGPU_SIMPLE_DENSE"] = env["BOWGPU_SIMPLE_DENSE"]
        try:
            yield label
        finally:
            os.environ["BOWGPU_NO_SIMPLE"] = "0"
            os.environ["BOWGPU_FORCE_GENERAL"] = "0"
            os.environ["BOWGPU_NO_LONG_ONLY"] = "0"
            os.environ["BOWGPU_LONG_CLASSIC"] = "0"
            os.environ["BOWGPU_LONG_STREAM_ALL"] = "0"
            os.environ.pop("BOWGPU_SIMPLE_DENSE", None)


def test_config1_dense_1e8_sum_mean_min_max():
    """configs[1]: 100 M dense float64 rows, fixed-interval rolling Sum/Mean/Min/Max - every output slot against the oracle"""
    ts, val = capi.gen_dense(0, N8, seed=42)
    ts_o, val_o = orc.gen_dense(0, N8, seed=42)
    aggs = [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1), ("Count", 1)]
    want, _ = orc.aggregate([orc.Column(ts_o, None, orc.INT64), orc.Column(val_o, None, orc.FLOAT64)], 0, 10, aggs)
    for label in _paths():
        got, info = capi.rolling_aggregate([ts, val], 0, 10, aggs, out_residency=capi.DEVICE)
        assert info.long_windows == 0
        for (k, _), g, w in zip(aggs, got, want):
            compare("config1 %s path=%s" % (k, label), g, w)


def test_config3_eight_columns_1e8():
    """configs[3]: 8 float64 columns x 100 M rows in ONE call (one pass per column over the shared interval column)"""
    ts, v0 = capi.gen_dense(0, N8, seed=42)
    cols = [ts, v0] + [capi.gen_dense(0, N8, seed=42 + c)[1] for c in range(1, 8)]
    aggs = [("WindowStart", 0)] + [("ArithmeticMean", 1 + c) for c in range(8)] + [("Max", 3), ("Sum", 8)]
    got, info = capi.rolling_aggregate(cols, 0, 10, aggs, out_residency=capi.DEVICE)
    assert capi.last_kernel_name() == "rolling_simple_kernel" and info.long_windows == 0
    ts_o = None
    for c in range(8):
        ts_o, val_o = orc.gen_dense(0, N8, seed=42 + c)
        mine = [(0, aggs[0])] + [(i, a) for i, a in enumerate(aggs) if a[1] == 1 + c]  # the interval column must be kept
        want, _ = orc.aggregate([orc.Column(ts_o, None, orc.INT64), orc.Column(val_o, None, orc.FLOAT64)], 0, 10,
                                [(k, 0 if k == "WindowStart" else 1) for _, (k, _) in mine])
        for (i, (k, _)), w in zip(mine, want):
            compare("config3 col %d %s" % (c, k), got[i], w)


def test_config2_sparse_nulls_linear_fill_then_mean():
    """configs[2]: irregular int64 ts, 30 % nulls, Linear fill (Rolling.Interpolate) then the rolling mean.
    2e7 rows against the oracle in full; 1e8 rows through properties."""
    ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
    aggs = [("WindowStart", 0), ("ArithmeticMean", 1), ("Count", 1), ("Min", 1)]
    n = N8 if BIG_HOST else 20_000_000
    ts, val = capi.gen_sparse(0, n, seed=5)
    ts_o, val_o, bm_o = orc.gen_sparse(0, n, seed=5)
    ocols = [orc.Column(ts_o, None, orc.INT64), orc.Column(val_o, bm_o, orc.FLOAT64)]
    for offset in (0, 7):
        filled = capi.rolling_interpolate([ts, val], 0, 100, ip, offset=offset, out_residency=capi.DEVICE)
        want = orc.interpolate(ocols, 0, 100, ip, offset=offset)
        # (values under null rows are whatever the input held - AppendBows copies them as they are, bowappend.go:44-51)
        cmp_out("config2 filled ts off=%d" % offset, filled[0], want[0])
        cmp_out("config2 filled val off=%d" % offset, filled[1], want[1])
        m = filled[0].length
        cols2 = [capi.Column(filled[0].values, None, capi.INT64, 0, m, 0),
                 capi.Column(filled[1].values, filled[1].validity, capi.FLOAT64, 0, m, -1)]
        got, _ = capi.rolling_aggregate(cols2, 0, 100, aggs, offset=offset, out_residency=capi.DEVICE)
        exp, _ = orc.aggregate([want[0], want[1]], 0, 100, aggs, offset=offset)
        for (k, _), g, w in zip(aggs, got, exp):
            compare("config2 %s off=%d" % (k, offset), g, w)
    del ts, val, filled, cols2, got

    # ---- full size: properties
    ts, val = capi.gen_sparse(0, N8, seed=5)
    s0, W = capi.plan_windows(ts, 100, 7)
    filled = capi.rolling_interpolate([ts, val], 0, 100, ip, offset=7, out_residency=capi.DEVICE)
    m = filled[0].length
    assert N8 <= m <= N8 + W
    f_ts = capi.Column(filled[0].values, None, capi.INT64, 0, m, 0)
    f_val = capi.Column(filled[1].values, filled[1].validity, capi.FLOAT64, 0, m, -1)
    assert capi.is_col_sorted(f_ts)
    # same windows before and after (interpolation only adds rows AT window starts)
    assert capi.plan_windows(f_ts, 100, 7) == (s0, W)
    # idempotence: every non-empty window now starts with a row on its start => nothing left to add, same bits back
    again = capi.rolling_interpolate([f_ts, f_val], 0, 100, ip, offset=7, out_residency=capi.DEVICE)
    assert again[0].length == m
    assert capi.checksum64(again[0].values, m) == capi.checksum64(filled[0].values, m)
    assert capi.checksum64(again[1].values, m) == capi.checksum64(filled[1].values, m)
    assert again[1].null_count == filled[1].null_count
    # conservation: rows and valid values per window add up; added rows = m - N
    aggs2 = [("WindowStart", 0), ("NumRows", 1), ("Count", 1), ("ArithmeticMean", 1)]
    before, _ = capi.rolling_aggregate([ts, val], 0, 100, aggs2, offset=7)
    after, _ = capi.rolling_aggregate([f_ts, f_val], 0, 100, aggs2, offset=7)
    nb, na = before[1].host_arrays()[0], after[1].host_arrays()[0]
    assert nb.sum() == N8 and na.sum() == m
    added = na - nb
    assert ((added == 0) | (added == 1)).all() and added.sum() == m - N8
    cb, ca = before[2].host_arrays()[0], after[2].host_arrays()[0]
    assert cb.sum() == capi.aggregate_whole([ts, val], 0, [("Count", 1)])[0].to_list()[0]  # = valid rows of the column
    assert ((ca - cb) >= 0).all() and ((ca - cb) <= added).all()
    assert np.array_equal(before[0].host_arrays()[0], after[0].host_arrays()[0])


def test_headline_1e9_every_window_properties_and_sharded():
    """the benched configuration (1e9 rows, interval 10, WindowStart + ArithmeticMean and friends)"""
    ts, val = capi.gen_dense(0, N9, seed=42)
    aggs = [("WindowStart", 0), ("ArithmeticMean", 1), ("Count", 1), ("Sum", 1), ("NumRows", 1)]
    got, info = capi.rolling_aggregate([ts, val], 0, 10, aggs, out_residency=capi.DEVICE)
    assert capi.last_kernel_name() == "rolling_simple_kernel" and info.long_windows == 0
    W = N9 // 10
    assert [g.length for g in got] == [W] * 5 and [g.null_count for g in got] == [0] * 5
    ws = got[0].host_arrays()[0]
    assert ws[0] == 0 and ws[-1] == 10 * (W - 1) and (np.diff(ws) == 10).all()
    cnt = got[2].host_arrays()[0]
    assert (cnt == 10).all()
    nr = got[4].host_arrays()[0]
    assert (nr == 10.0).all()
    mean, sm = got[1].host_arrays()[0], got[3].host_arrays()[0]
    assert np.array_equal(mean, sm / 10.0)  # the same left-to-right sum feeds both (arithmeticmean.go:17-28)
    assert 0.0 <= mean.min() and mean.max() < 1.0
    if BIG_HOST:  # every window of the benched call against the oracle, bit for bit
        ts_o, val_o = orc.gen_dense(0, N9, seed=42)
        want, _ = orc.aggregate([orc.Column(ts_o, None, orc.INT64), orc.Column(val_o, None, orc.FLOAT64)], 0, 10, aggs[:2])
        assert np.array_equal(ws, want[0].values[:W])
        assert np.array_equal(mean.view(np.uint64), want[1].values[:W].view(np.uint64))
        del ts_o, val_o, want
    # random row ranges against the oracle, bit for bit
    rng = np.random.default_rng(1)
    for a in [0, N9 - 2_000_000] + [int(x) * 10 for x in rng.integers(0, (N9 - 2_000_000) // 10, 6)]:
        ts_o, val_o = orc.gen_dense(a, 2_000_000, seed=42)
        want, _ = orc.aggregate([orc.Column(ts_o, None, orc.INT64), orc.Column(val_o, None, orc.FLOAT64)], 0, 10, aggs[:2])
        k0 = a // 10
        assert np.array_equal(ws[k0:k0 + 200_000], want[0].values[:200_000])
        assert np.array_equal(mean[k0:k0 + 200_000].view(np.uint64), want[1].values[:200_000].view(np.uint64))
    del ws, cnt, nr, sm

    # sharded == whole: 8 simulated ranks, interval 10 with offset 3 so a window straddles every shard boundary
    whole, _ = capi.rolling_aggregate([ts, val], 0, 10, aggs[:2], offset=3)
    w_ts, w_mean = whole[0].host_arrays()[0], whole[1].host_arrays()[0]
    del ts, val, got, whole
    world, R = 8, N9 // 8
    provs = []
    for r in range(world):
        cols = list(capi.gen_dense(r * R, R, seed=42))
        provs.append(sharded.GpuProvider(cols, 0, 10, aggs[:2], offset=3))
    owned = [(d.first_slot_window_id, d.windows_owned) for d in sharded.run_local(provs)]
    assert capi.last_kernel_name() == "rolling_simple_kernel"
    covered = 0
    for r, (fs, nwin) in enumerate(owned):
        assert fs == covered  # each rank owns the next contiguous run of windows
        g_ts, g_mean = provs[r].outs[0].host_arrays()[0][:nwin], provs[r].outs[1].host_arrays()[0][:nwin]
        assert np.array_equal(g_ts, w_ts[fs:fs + nwin]), r
        assert np.array_equal(g_mean.view(np.uint64), w_mean[fs:fs + nwin].view(np.uint64)), r
        covered += nwin
    assert covered == len(w_ts)


def test_fills_and_is_col_sorted_at_full_size():
    """the four fills and IsColSorted on configs[2]'s columns (irregular int64 ts, 30 % nulls) at 1e8 rows (2e7 on small hosts),
    every row against the oracle"""
    n = N8 if BIG_HOST else 20_000_000
    ts, val = capi.gen_sparse(0, n, seed=9)
    ts_o, val_o, bm_o = orc.gen_sparse(0, n, seed=9)
    oval, ots = orc.Column(val_o, bm_o, orc.FLOAT64), orc.Column(ts_o, None, orc.INT64)
    for method in ("Previous", "Next", "Mean"):
        got, gu = capi.fill(val, method, out_residency=capi.DEVICE)
        want, wu = orc.fill(oval, method)
        assert gu == wu
        cmp_out("full-size Fill%s" % method, got, want)
        del got, want
    got, gu = capi.fill_linear([ts, val], 0, 1, out_residency=capi.DEVICE)
    want, wu = orc.fill_linear([ots, oval], 0, 1)
    assert gu == wu
    cmp_out("full-size FillLinear", got, want)
    assert capi.is_col_sorted(ts) and orc.is_col_sorted(ots)
    # a filled column has no nulls left (ts is sorted, the first and the last value of gen_sparse columns may be null)
    assert got.null_count == want.length - int(want.valid_mask().sum())


def test_mode_at_scale():
    """aggregation.Mode on 2e7 rows of configs[2]'s generator (ten distinct values, 30 % nulls: ties in nearly every window):
    10-row windows through the lane-per-window class, 1000-row windows through the workgroup class, every window against the oracle"""
    n = 20_000_000
    ts, val = capi.gen_sparse(0, n, seed=3)
    ts_o, val_o, bm_o = orc.gen_sparse(0, n, seed=3)
    ocols = [orc.Column(ts_o, None, orc.INT64), orc.Column(val_o, bm_o, orc.FLOAT64)]
    for interval, rows in ((100, 20_000_000), (10_000, 2_000_000)):   # (the oracle's Mode is quadratic in a window's rows)
        cc = [capi.Column(ts.values, None, capi.INT64, 0, rows, 0), capi.Column(val.values, val.validity, capi.FLOAT64, 0, rows, -1)]
        oc = [orc.Column(ts_o, None, orc.INT64, length=rows), orc.Column(val_o, bm_o, orc.FLOAT64, length=rows)]
        aggs = [("WindowStart", 0), ("Mode", 1), ("Count", 1)]
        got, info = capi.rolling_aggregate(cc, 0, interval, aggs, offset=3, out_residency=capi.DEVICE)
        want, _ = orc.aggregate(oc, 0, interval, aggs, offset=3)
        for (k, _), g, w in zip(aggs, got, want):
            compare("Mode at scale I=%d %s" % (interval, k), g, w)
