"""SURVEY §8(b) "Threading": cgo calls arrive on arbitrary OS threads, several at once.  The ABI keeps its state per thread
(context, stream, scratch, error string): concurrent calls from many threads give the same results as the oracle, an error in
one thread is reported there and nowhere else, and a column uploaded by one thread serves calls made by another."""
import threading

import numpy as np
import pytest

from bow_amd import capi
from oracle import pyoracle as orc
from test_gpu_aggregate import compare
from tolerance import order_free_bounds
from test_gpu_fuzz import cmp_out

pytestmark = pytest.mark.gpu

AGGS = [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1), ("Count", 1), ("First", 1), ("Last", 1),
        ("WeightedAverageStep", 1), ("Mode", 1)]


def _case(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2000, 60_000))
    ts = np.cumsum(rng.integers(0, 4, n)).astype(np.int64) - int(rng.integers(0, 500))
    vals = np.round(np.abs(rng.standard_normal(n)), 1)   # (one sign: the long-window path's tree sums are compared relatively)
    valid = rng.random(n) > 0.3
    interval = int(rng.choice([3, 10, 64, 1000]))
    return ts, vals, valid, interval


def test_concurrent_calls_from_many_threads():
    n_threads, per_thread = 8, 12
    cases = {(t, j): _case(100 * t + j) for t in range(n_threads) for j in range(per_thread)}
    want = {}
    for key, (ts, vals, valid, interval) in cases.items():
        bm = np.packbits(valid, bitorder="little")
        ocols = [orc.Column(ts, None, orc.INT64), orc.Column(vals, bm, orc.FLOAT64)]
        want[key] = (orc.aggregate(ocols, 0, interval, AGGS, offset=1)[0], orc.fill(ocols[1], "Previous")[0])
    shared_ts, shared_vals, _, _ = _case(7)
    shared = [capi.Column(shared_ts, None, capi.INT64).to_device(), capi.Column(shared_vals, None, capi.FLOAT64).to_device()]  # uploaded HERE
    shared_want = orc.aggregate([orc.Column(shared_ts, None, orc.INT64), orc.Column(shared_vals, None, orc.FLOAT64)], 0, 10, AGGS[:8])[0]
    errors = []
    start = threading.Barrier(n_threads)

    def worker(t):
        try:
            start.wait()
            for j in range(per_thread):
                ts, vals, valid, interval = cases[(t, j)]
                bm = np.packbits(valid, bitorder="little")
                cols = [capi.Column(ts, None, capi.INT64), capi.Column(vals, bm, capi.FLOAT64, 0, len(vals), -1)]
                if (t + j) % 2:
                    cols = [c.to_device() for c in cols]
                outs, info = capi.rolling_aggregate(cols, 0, interval, AGGS, offset=1)
                exp, exp_fill = want[(t, j)]
                tol = None
                if info.long_windows:
                    tol = order_free_bounds([orc.Column(ts, None, orc.INT64), orc.Column(vals, bm, orc.FLOAT64)], 0, interval, AGGS, offset=1, ref=exp)
                for i, ((k, _), g, w) in enumerate(zip(AGGS, outs, exp)):
                    exact = k not in ("Sum", "ArithmeticMean", "WeightedAverageStep") or info.long_windows == 0
                    compare("thread %d case %d %s" % (t, j, k), g, w, exact=exact, bound=None if exact else tol[i])
                g, _ = capi.fill(cols[1], "Previous")
                cmp_out("thread %d case %d FillPrevious" % (t, j), g, exp_fill)
                if j % 4 == t % 4:   # an error of this thread's own, between good calls: message and code stay here
                    bad = ts.copy(); bad[len(bad) // 2] -= 10_000
                    with pytest.raises(capi.BowGpuError) as e:
                        capi.rolling_aggregate([capi.Column(bad, None, capi.INT64), cols[1]], 0, interval, AGGS[:3])
                    assert "TS_UNSORTED" in str(e.value) and "ascending" in str(e.value)
                outs, _ = capi.rolling_aggregate(shared, 0, 10, AGGS[:8])   # columns another thread put on the device
                for (k, _), g, w in zip(AGGS[:8], outs, shared_want):
                    compare("thread %d shared %s" % (t, k), g, w)
        except BaseException as ex:   # noqa: BLE001 - reported by the main thread
            errors.append((t, repr(ex)[:600]))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=500)
    assert not any(th.is_alive() for th in threads), "a worker hangs"
    assert not errors, errors[:3]


def test_a_thread_that_exits_gives_its_device_memory_back():
    """Per-thread state (stream, pools, pinned block, the cache of freed scratch blocks) must not outlive its thread: a cgo host
    whose calls hop OS threads would otherwise multiply the footprint by the number of threads it ever used."""
    n = 20_000_000
    ts = np.arange(n, dtype=np.int64)
    vals = np.ones(n)
    capi.rolling_aggregate([capi.Column(ts[:1000], None, capi.INT64), capi.Column(vals[:1000], None, capi.FLOAT64)], 0, 10, AGGS[:3])
    capi.trim(True)
    free0, total = capi.mem_info()
    seen = {}

    def work():
        # host-resident columns: the call stages 2 x 160 MB through this thread's scratch cache
        capi.rolling_aggregate([capi.Column(ts, None, capi.INT64), capi.Column(vals, None, capi.FLOAT64)], 0, 10, AGGS[:3])
        seen["during"] = capi.mem_info()[0]

    th = threading.Thread(target=work)
    th.start()
    th.join()
    # (Thread.join returns when the Python side of the thread is done; the OS thread runs its thread-local destructors a moment later)
    import time
    deadline = time.monotonic() + 10.0
    while True:
        free1, _ = capi.mem_info()
        if free0 - free1 < 64 << 20 or time.monotonic() > deadline:
            break
        time.sleep(0.05)
    assert free0 - seen["during"] > 200 << 20          # the worker did cache its staging blocks while it lived ...
    assert free0 - free1 < 64 << 20, (free0, free1)    # ... and they are gone with it


def test_trim_frees_every_threads_cache():
    n = 10_000_000
    ts = np.arange(n, dtype=np.int64)
    vals = np.ones(n)
    capi.trim(True)
    free0, _ = capi.mem_info()
    hold, done = threading.Event(), threading.Event()

    def work():
        capi.rolling_aggregate([capi.Column(ts, None, capi.INT64), capi.Column(vals, None, capi.FLOAT64)], 0, 10, AGGS[:3])
        done.set()
        hold.wait(60)      # stay alive, cache populated

    th = threading.Thread(target=work)
    th.start()
    assert done.wait(120)
    assert free0 - capi.mem_info()[0] > 100 << 20
    freed = capi.trim(True)             # from THIS thread: trims the worker's cache too
    assert freed > 100 << 20
    # (the blocks were freed by a thread that did not allocate them: on some boxes hipMemGetInfo shows them as free a moment later)
    import time
    deadline = time.time() + 10.0
    while free0 - capi.mem_info()[0] >= 64 << 20 and time.time() < deadline:
        time.sleep(0.05)
    assert free0 - capi.mem_info()[0] < 64 << 20
    hold.set()
    th.join()
