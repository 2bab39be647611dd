"""The multi-rank protocol of bow_amd/sharded.py under torch.distributed (gloo, world_size 2 and 3)
on CPU.  Compute is replaced by a numpy provider that follows the same provider interface as the
HIP one (the HIP provider itself is covered by tests/test_gpu_sharded.py); what is tested here is
the exchange: s0 broadcast, plan all_gather, carry all_gather, ownership of straddling windows."""
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
    """Reference-order reducers over this rank's rows with numpy / python floats (test double)."""

    def __init__(self, ts, vals, interval):
        self.ts, self.vals, self.interval = ts, vals, interval
        self.out = None

    def first_last_nrows(self):
        n = len(self.ts)
        return (int(self.ts[0]), int(self.ts[-1]), n) if n else (0, 0, 0)

    def plan_s0(self):
        from oracle import pyoracle as orc
        return orc.plan_windows(orc.Column(self.ts, None, orc.INT64), self.interval, 0)[0]

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

    def _pack(self, st):
        from bow_amd import capi
        arr = (capi.CarryState * capi.CARRY_MAX_AGGS)()
        for i in range(len(AGGS)):
            arr[i].sum, arr[i].vmin, arr[i].vmax = st["sum"], st["vmin"], st["vmax"]
            arr[i].nn_min, arr[i].nn_max, arr[i].has_nn = st["vmin"], st["vmax"], st["has"]
            arr[i].count, arr[i].nrows, arr[i].has_value = st["count"], st["nrows"], st["has"]
        return bytes(arr)

    def _unpack(self, b):
        from bow_amd import capi
        a = (capi.CarryState * capi.CARRY_MAX_AGGS).from_buffer_copy(b)[0]
        return dict(sum=a.sum, vmin=a.vmin, vmax=a.vmax, count=a.count, nrows=a.nrows, has=a.has_value)

    def _emit(self, slot, wid, st):
        s0, I = self.s0, self.interval
        self.out[slot] = [s0 + wid * I, st["sum"] if st["nrows"] else 0.0,
                          (st["sum"] / st["count"]) if st["count"] else None,
                          st["vmin"] if st["has"] else None, st["vmax"] if st["has"] else None, st["count"]]

    def shard_aggregate(self, s0, holds_row0, lead):
        from bow_amd import capi
        self.s0 = s0
        I = self.interval
        carry = capi.ShardCarry()
        n = len(self.ts)
        if n == 0:
            carry.first_window_id = carry.last_window_id = -1
            self.out = []
            return bytes(carry)
        wid = (self.ts - s0) // I
        wf, wl = int(wid[0]), int(wid[-1])
        self.wf = wf
        self.out = [None] * (wl - wf + 1 + lead)
        empty = dict(sum=0.0, vmin=0.0, vmax=0.0, count=0, nrows=0, has=0)
        for k in range(wf - lead, wl + 1):
            rows = self.vals[wid == k]
            self._emit(k - (wf - lead), k, self._state(rows) if len(rows) else empty)
        carry.first_window_id, carry.last_window_id = wf, wl
        carry.first_ts, carry.last_ts, carry.nrows, carry.naggs = int(self.ts[0]), int(self.ts[-1]), n, len(AGGS)
        last = (capi.CarryState * capi.CARRY_MAX_AGGS).from_buffer_copy(self._pack(self._state(self.vals[wid == wl])))
        for i in range(capi.CARRY_MAX_AGGS):
            carry.last[i] = last[i]
        return bytes(carry)

    def fix_first(self, s0, lead, first_window_id, seed_bytes):
        wid = (self.ts - s0) // self.interval
        st = self._state(self.vals[wid == first_window_id], self._unpack(seed_bytes))
        self._emit(lead, first_window_id, st)
        return self._pack(st)

    def merge(self, a, b):
        x, y = self._unpack(a), self._unpack(b)
        if not y["has"]:
            x["nrows"] += y["nrows"]
            return self._pack(x)
        if not x["has"]:
            y["nrows"] += x["nrows"]
            return self._pack(y)
        return self._pack(dict(sum=x["sum"] + y["sum"], vmin=min(x["vmin"], y["vmin"]), vmax=max(x["vmax"], y["vmax"]),
                               count=x["count"] + y["count"], nrows=x["nrows"] + y["nrows"], has=1))


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
        first_slot, owned, plan = sharded.sharded_aggregate(prov, dist, torch, rank, world, interval)
        q.put((rank, first_slot, owned, prov.out[:max(owned, 0)]))
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
