"""The multi-rank protocol of bow_amd/sharded.py under torch.distributed (gloo, world_size 2 and 3)
on CPU.  Compute is replaced by a numpy provider that follows the same provider interface as the
HIP one (the HIP provider itself is covered by tests/test_gpu_sharded.py); what is tested here is
the exchange (ONE all_gather of fixed-size records per call) and the ownership decisions bowgpu_shard_plan takes from the
gathered records - pure host arithmetic inside libbowgpu.so, so it runs here without a GPU."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

AGGS = [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1), ("Count", 1)]


class NumpyProvider:
    """Reference-order reducers over this rank's rows with numpy / python floats (test double for GpuProvider: same
    begin / finish interface, every ownership decision taken from bowgpu_shard_plan)."""

    def __init__(self, ts, vals, interval, offset=0):
        self.ts, self.vals, self.interval, self.offset = ts, vals, interval, offset
        self.out = None
        self.events = []
        self._early = None

    def pass_begin(self, record):
        """the rank's own windows from its own record alone, BEFORE the gathered records exist (what bowgpu_shard_pass_begin
        enqueues on the device): local numbering, slot 0 = the window of the first row on the offset-aligned grid"""
        from bow_amd import capi
        self.events.append("pass_begin")
        rec = capi.ShardRecord.from_buffer_copy(bytes(record))
        if rec.nrows == 0 or rec.first_ts < 0:
            return False
        I, off = self.interval, self.offset % self.interval
        base = (rec.first_ts - off) // I * I + off
        lw = (self.ts - base) // I
        empty = dict(sum=0.0, vmin=0.0, vmax=0.0, count=0, nrows=0, has=0)
        self._early = (base, [self._state(self.vals[lw == k]) if (lw == k).any() else empty for k in range(int(lw[-1]) + 1)])
        return True

    def _state(self, rows, seed=None):
        st = dict(sum=0.0, vmin=0.0, vmax=0.0, count=0, nrows=0, has=0) if seed is None else dict(seed)
        for x in rows:
            x = float(x)
            st["sum"] += x
            if st["has"]:
                if x < st["vmin"]: st["vmin"] = x
                if x > st["vmax"]: st["vmax"] = x
            else:
                st["vmin"] = st["vmax"] = x
                st["has"] = 1
            st["count"] += 1
            st["nrows"] += 1
        return st

    @staticmethod
    def _fill(cs, st):
        cs.sum, cs.vmin, cs.vmax = st["sum"], st["vmin"], st["vmax"]
        cs.nn_min, cs.nn_max, cs.has_nn = st["vmin"], st["vmax"], st["has"]
        cs.count, cs.nrows, cs.has_value = st["count"], st["nrows"], st["has"]

    @staticmethod
    def _unpack(a):
        return dict(sum=a.sum, vmin=a.vmin, vmax=a.vmax, count=a.count, nrows=a.nrows, has=a.has_value)

    def _emit(self, slot, wid, st):
        s0, I = self.s0, self.interval
        self.out[slot] = [s0 + wid * I, st["sum"] if st["nrows"] else 0.0,
                          (st["sum"] / st["count"]) if st["count"] else None,
                          st["vmin"] if st["has"] else None, st["vmax"] if st["has"] else None, st["count"]]

    def begin(self, global_s0=None):
        from bow_amd import capi
        self.events.append("begin")
        rec = capi.ShardRecord()
        n = len(self.ts)
        rec.nrows, rec.naggs = n, len(AGGS)
        if n == 0:
            return bytes(rec)
        I = self.interval
        rec.first_ts, rec.last_ts = int(self.ts[0]), int(self.ts[-1])
        off = self.offset % I                      # (non-negative offsets in these tests)
        start = (rec.last_ts - off) // I * I + off   # the window grid does not depend on the frame's first row
        rec.carry_from_ts = start
        st = self._state(self.vals[self.ts >= start])
        for i in range(len(AGGS)):
            self._fill(rec.last[i], st)
        return bytes(rec)

    def finish(self, records, rank):
        from bow_amd import capi, sharded
        self.events.append("finish")
        self.collected = False
        d = sharded.plan(records, rank, self.interval, self.offset)
        assert not d.retry_with_s0
        self.s0 = s0 = d.s0
        I = self.interval
        self.out = []
        if d.first_window_id < 0:
            return 0, d
        wid = (self.ts - s0) // I
        wf, wl, lead = d.first_window_id, d.last_window_id, d.lead_empty_windows
        assert (wf, wl) == (int(wid[0]), int(wid[-1]))
        self.out = [None] * d.windows_local
        empty = dict(sum=0.0, vmin=0.0, vmax=0.0, count=0, nrows=0, has=0)
        early, self._early = self._early, None
        if early is not None and lead == 0 and early[0] == s0 + wf * I:
            # the pass that ran while the records travelled assumed exactly this (no gap to the left neighbour): collect it
            assert len(early[1]) == wl - wf + 1
            for k, st in enumerate(early[1]):
                self._emit(k, wf + k, st)
            self.collected = True
        else:
            self.collected = False
            for k in range(wf - lead, wl + 1):
                rows = self.vals[wid == k]
                self._emit(k - (wf - lead), k, self._state(rows) if len(rows) else empty)
        if d.seed_first_rank >= 0:
            recs = [capi.ShardRecord.from_buffer_copy(b) for b in records]
            seed = self._unpack(recs[d.seed_first_rank].last[0])
            for q in range(d.seed_first_rank + 1, rank):
                if recs[q].nrows:
                    seed = self._merge(seed, self._unpack(recs[q].last[0]))
            self._emit(lead, wf, self._state(self.vals[wid == wf], seed))
        return 0, d

    @staticmethod
    def _merge(x, y):
        if not y["has"]:
            return dict(x, nrows=x["nrows"] + y["nrows"])
        if not x["has"]:
            return dict(y, nrows=x["nrows"] + y["nrows"])
        return dict(sum=x["sum"] + y["sum"], vmin=min(x["vmin"], y["vmin"]), vmax=max(x["vmax"], y["vmax"]),
                    count=x["count"] + y["count"], nrows=x["nrows"] + y["nrows"], has=1)


def _worker(rank, world, port, bounds, interval, q):
    import torch
    import torch.distributed as dist
    from bow_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ts, vals = _data()
        a, b = bounds[rank], bounds[rank + 1]
        prov = NumpyProvider(ts[a:b], vals[a:b], interval)
        gather = sharded.Gather(dist, torch, world, "cpu")
        d = sharded.sharded_aggregate(prov, gather, rank, world)        # the overlapped order: the pass starts before the records are in
        assert gather.calls == 1          # ONE exchange per call
        assert prov.events == ["begin", "pass_begin", "finish"]
        overlapped, was_collected = [list(r) if r is not None else None for r in prov.out], prov.collected
        # ... and the serial order (begin -> exchange -> finish) gives the same bytes, decision for decision
        prov.events = []
        d2 = sharded.sharded_aggregate(prov, gather, rank, world, overlap=False)
        assert gather.calls == 2 and prov.events == ["begin", "finish"] and not prov.collected
        assert bytes(d2) == bytes(d)
        assert [repr(r) for r in prov.out] == [repr(r) for r in overlapped]
        # a rank with rows and no gap in front of it must have USED its early pass
        assert was_collected == (b > a and d.lead_empty_windows == 0)
        q.put((rank, d.first_slot_window_id, d.windows_owned, prov.out[:max(d.windows_owned, 0)]))
    finally:
        dist.destroy_process_group()


def _data():
    rng = np.random.default_rng(5)
    n = 3000
    ts = np.cumsum(rng.integers(1, 9, n)).astype(np.int64)
    vals = np.round(rng.standard_normal(n) * 50, 3)
    return ts, vals


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,bounds,interval", [
    (2, [0, 1500, 3000], 7),          # one straddling window
    (2, [0, 1, 3000], 100),           # a one-row left shard
    (3, [0, 1000, 1004, 3000], 50),   # the middle shard lies inside ONE window: three ranks share it
    (3, [0, 1200, 1200, 3000], 13),   # an empty shard
])
def test_protocol_under_gloo(world, bounds, interval):
    import torch.multiprocessing as mp
    from oracle import pyoracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bounds, interval, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ts, vals = _data()
    exp, _ = orc.aggregate([orc.Column(ts, None, orc.INT64), orc.Column(vals, None, orc.FLOAT64)], 0, interval, AGGS)
    W = exp[0].length
    got = [None] * W
    for rank, first_slot, owned, rows in sorted(results):
        for k, row in enumerate(rows):
            assert got[first_slot + k] is None, "window %d owned twice" % (first_slot + k)
            got[first_slot + k] = row
    assert all(g is not None for g in got)
    exp_lists = [e.to_list() for e in exp]
    three_way = world == 3 and bounds[2] - bounds[1] in (4,)
    for k in range(W):
        for i in range(len(AGGS)):
            e, g = exp_lists[i][k], got[k][i]
            if three_way and AGGS[i][0] in ("Sum", "ArithmeticMean") and e is not None:
                assert abs(g - e) <= 1e-11 * max(1.0, abs(e)), (k, AGGS[i], g, e)
            else:
                assert g == e, (k, AGGS[i], g, e)


def _records(spans, interval, offset=0):
    """records as bowgpu_shard_begin would fill them, from (first_ts, last_ts, nrows) per rank"""
    from bow_amd import capi
    out = []
    for f, l, n in spans:
        r = capi.ShardRecord()
        r.nrows = n
        if n:
            r.first_ts, r.last_ts = f, l
            off = offset % interval
            r.carry_from_ts = (l - off) // interval * interval + off
        out.append(bytes(r))
    return out


def test_c_plan_equals_the_python_plan():
    """bowgpu_shard_plan (C, behind the ABI) against the round-1 Python rules on random shard layouts: empty ranks, one-row
    ranks, several ranks inside one window, gaps of many empty windows between ranks, offsets."""
    from bow_amd import sharded
    from legacy_shard_plan import ShardPlan as OldPlan
    rng = np.random.default_rng(2024)
    cases = 0
    for _ in range(3000):
        world = int(rng.integers(1, 9))
        interval = int(rng.choice([1, 3, 7, 10, 100, 1000]))
        offset = int(rng.integers(0, 3 * interval))
        t = int(rng.integers(0, 5000))
        spans = []
        for _r in range(world):
            kind = rng.integers(0, 6)
            if kind == 0:
                spans.append((0, 0, 0))
                continue
            t += int(rng.choice([0, 1, 2, interval, 5 * interval + 3, 1]))
            f = t
            n = 1 if kind == 1 else int(rng.integers(1, 50))
            t += 0 if n == 1 else int(rng.integers(0, 4 * interval + 1))
            spans.append((f, t, n))
        recs = _records(spans, interval, offset)
        new = sharded.ShardPlan(recs, interval, offset)
        first = next((f for f, _l, n in spans if n), None)
        if first is None:
            assert all(w == -1 for w in new.wf)
            continue
        s0 = sharded.first_window_start(first, interval, offset)
        assert new.s0 == s0
        old = OldPlan(s0, interval, [s[0] for s in spans], [s[1] for s in spans], [s[2] for s in spans])
        assert new.wf == old.wf and new.wl == old.wl, (spans, interval, offset)
        for r in range(world):
            assert new.lead_empty(r) == old.lead_empty(r), (r, spans, interval, offset)
            assert new.seed_ranks(r) == old.seed_ranks(r), (r, spans, interval, offset)
            assert new.drops_last(r) == old.drops_last(r), (r, spans, interval, offset)
            d = new.decisions[r]
            if old.wf[r] >= 0:
                W_local = old.wl[r] - old.wf[r] + 1 + old.lead_empty(r)
                assert d.windows_local == W_local and d.windows_owned == W_local - (1 if old.drops_last(r) else 0)
                assert d.first_slot_window_id == old.wf[r] - old.lead_empty(r)
                assert d.next_rank == old.right_nonempty(r)
            assert not d.retry_with_s0
        # every window of the frame is owned exactly once
        W = new.decisions[0].num_windows
        seen = np.zeros(W, dtype=int)
        for d in new.decisions:
            if d.first_slot_window_id >= 0:
                seen[d.first_slot_window_id:d.first_slot_window_id + d.windows_owned] += 1
        assert (seen == 1).all(), (spans, interval, offset)
        cases += 1
    assert cases > 2000


def test_c_plan_rejects_ranks_out_of_order_and_handles_rows_below_s0():
    from bow_amd import capi, sharded
    recs = _records([(0, 50, 10), (40, 90, 10)], 10)
    with pytest.raises(capi.BowGpuError) as e:
        sharded.plan(recs, 0, 10)
    assert e.value.code == -14      # BOWGPU_ERR_TS_UNSORTED
    # Go's truncating division: first ts -7, interval 5, offset 4 -> s0 = -6 ABOVE the first row; rows below s0 ride in
    # window 0 (rolling.go:96-99, :194-196).  Rank 0 holds rows in grid cells -1 and 0, rank 1 continues window 0: the
    # first-attempt state of rank 0 (cut on the grid: rows >= -6 only) cannot seed it -> every rank is told to retry with s0
    recs = _records([(-7, -3, 4), (-2, 20, 9)], 5, 4)
    for r in range(2):
        d = sharded.plan(recs, r, 5, 4)
        assert d.s0 == -6 and d.retry_with_s0 == 1
    assert sharded.first_window_start(-7, 5, 4) == -6
    # second attempt: flags bit 0 set, the state covers all rows of window 0
    fixed = []
    for b in recs:
        r = capi.ShardRecord.from_buffer_copy(b)
        r.flags = 1
        r.carry_from_ts = -(1 << 63) if r.last_ts < -6 + 5 else r.carry_from_ts
        fixed.append(bytes(r))
    d1 = sharded.plan(fixed, 1, 5, 4)
    assert d1.retry_with_s0 == 0 and d1.first_window_id == 0 and d1.seed_first_rank == 0
    d0 = sharded.plan(fixed, 0, 5, 4)
    assert d0.first_window_id == 0 and d0.last_window_id == 0 and d0.drops_last == 1 and d0.windows_owned == 0


def test_forced_collective_at_world_1_under_gloo():
    """Gather(force_collective=True) at world 1 issues the real exchange instead of returning the payload (the switch the GPU suite
    uses to run RCCL on a 1-GPU box, tests/test_gpu_rccl_world1.py); here over gloo on the CPU, list form, and then one whole
    protocol step through it: same decisions and bytes as the shortcut."""
    import torch
    import torch.distributed as dist
    from bow_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        plain = sharded.Gather(dist, torch, 1, "cpu")
        forced = sharded.Gather(dist, torch, 1, "cpu", force_collective=True)
        assert not plain.collective and forced.collective and not forced.on_gpu and not forced.single
        payload = bytes(range(256)) * 9 + bytes(forced.n - 2304)
        payload = payload[:forced.n]
        forced.start(payload)
        assert forced._work is not None                       # a collective is in flight
        assert forced.wait() == [payload] == plain(payload)
        assert forced.calls == 1 and plain.calls == 0         # (the shortcut does not count as an exchange)
        ts, vals = _data()
        outs = []
        for g in (plain, forced):
            prov = NumpyProvider(ts, vals, 7)
            d = sharded.sharded_aggregate(prov, g, 0, 1)
            outs.append((bytes(d), [repr(r) for r in prov.out]))
        assert outs[0] == outs[1]
    finally:
        dist.destroy_process_group()
