"""Row-range sharded Rolling.Aggregate across the GPUs of one node - the host side, and it is thin.

One process per GPU; rank r holds rows [r*R, (r+1)*R) of every column.  Windows are disjoint row ranges, so every rank
reduces its own rows with no data-path collective.  The protocol lives BEHIND the C ABI (include/bowgpu.h, "the shard
protocol"); what is left here is the transport:

    record  = bowgpu_shard_begin(my columns)                    one small kernel, one read-back        (~2.8 KB)
    all_gather(record) STARTS                                    THE exchange of the call (RCCL over xGMI / gloo in tests)
    bowgpu_shard_pass_begin(my columns, record)                  the pass over my rows, enqueued: it runs while the records travel
    records = the all_gather's result
    bowgpu_shard_finish(my columns, records) -> my output slots  collects the pass + the boundary-window stitch

Every rank derives the same ownership decisions from the same gathered bytes (bowgpu_shard_plan: which rank outputs a
window that straddles a boundary, who emits the empty windows between two shards, whose running state seeds whose first
window, who needs whose first row for an inclusive window).  A window that straddles a boundary is finished by the right
rank: it re-walks its own rows of that window seeded with the left rank's running state, i.e. in the reference's row order
(bit-exact for two ranks per window; windows spanning three or more ranks merge partial sums).

One corner needs a second round: rows below the first window start (Go's truncating division on negative timestamps,
rolling.go:96-99) split across ranks - finish() then returns RETRY on every rank alike and the ranks repeat begin() with
the now-known global s0.

The compute sits behind a small provider interface so the transport can be tested on CPU under gloo with a numpy provider
(tests/test_sharded_gloo.py); the product provider is GpuProvider (HIP kernels through the C ABI, no CPU path).
"""
import ctypes as C
import time

import numpy as np

RETRY = 1  # BOWGPU_SHARD_RETRY


def first_window_start(first_ts, interval, offset=0):
    """s0 of rolling.IntervalRolling from the first timestamp alone (rolling.go:96-99 + enforceIntervalAndOffset :114-128),
    in Go's integer semantics (division and % truncate toward zero), integer arithmetic only."""
    if interval <= 0:
        raise ValueError("strictly positive interval required")
    if offset >= interval or offset <= -interval:
        offset = (abs(offset) % interval) * (1 if offset >= 0 else -1)   # Go's %: sign of the dividend
    if offset < 0:
        offset += interval
    q = abs(first_ts) // interval
    if first_ts < 0:
        q = -q
    s0 = q * interval + offset
    if s0 > first_ts:
        s0 -= interval
    return s0


def plan(records, rank, interval, offset=0):
    """bowgpu_shard_plan on gathered record bytes -> capi.ShardDecision (pure host arithmetic inside the library)."""
    from . import capi
    world = len(records)
    arr = (capi.ShardRecord * world)()
    for q, b in enumerate(records):
        C.memmove(C.byref(arr[q]), bytes(b), C.sizeof(capi.ShardRecord))
    d = capi.ShardDecision()
    capi.check(capi.lib().bowgpu_shard_plan(arr, world, rank, C.c_int64(interval), C.c_int64(offset), C.byref(d)))
    return d


class ShardPlan:
    """Every rank's decisions at once (what the old Python-side plan exposed; now a view of bowgpu_shard_plan)."""

    def __init__(self, records=None, interval=None, offset=0, decisions=None):
        self.decisions = decisions if decisions is not None else [plan(records, r, interval, offset) for r in range(len(records))]
        self.world = len(self.decisions)
        self.s0 = self.decisions[0].s0
        self.wf = [d.first_window_id for d in self.decisions]
        self.wl = [d.last_window_id for d in self.decisions]

    def lead_empty(self, r):
        return self.decisions[r].lead_empty_windows

    def drops_last(self, r):
        return bool(self.decisions[r].drops_last)

    def seed_ranks(self, r):
        sf = self.decisions[r].seed_first_rank
        return [] if sf < 0 else [q for q in range(sf, r) if self.wf[q] >= 0]


class Gather:
    """all_gather of one fixed-size record per rank, as bytes.  Buffers are allocated ONCE: a page-locked host pair and (nccl) a
    device pair, so a step hands the HIP runtime no pageable memory (runtime-pinned pageable pages faulted on
    fresh machines) and allocates nothing.  start() puts the exchange in flight on a stream of its own - upload, collective,
    download - and returns; wait() blocks the HOST until the bytes are there.  In between the caller enqueues the rank's pass on
    the library's stream (bowgpu_shard_pass_begin), which is how the exchange leaves the critical path.  The collective form is
    chosen once, from the backend, so every rank always issues the same collective and a failure surfaces as an error instead
    of a hang."""

    def __init__(self, dist, torch, world, device, nbytes=None, force_collective=False):
        from . import capi
        self.dist, self.torch, self.world, self.device = dist, torch, world, device
        self.n = nbytes if nbytes is not None else C.sizeof(capi.ShardRecord)
        # force_collective: run the real exchange even at world 1 (no shortcut) - how the RCCL transport is exercised on a 1-GPU box
        # (tests/test_gpu_rccl_world1.py): pinned pair, side stream, all_gather_into_tensor(async_op=True), stream-side wait, D2H
        self.collective = world > 1 or bool(force_collective)
        self.on_gpu = self.collective and str(device).startswith("cuda")
        self.single = self.collective and dist.get_backend() == "nccl"   # all_gather_into_tensor: RCCL; gloo takes the list form
        self.ms = 0.0          # host wall time start() -> wait() returned, summed (bench.py: the window the exchange had)
        self.wait_ms = 0.0     # ... of which the host spent blocked inside wait()
        self.calls = 0
        self._work = None
        if not self.collective:
            return
        pin = bool(self.on_gpu)
        self.h_send = torch.empty(self.n, dtype=torch.uint8, pin_memory=pin)
        self.h_recv = torch.empty(world * self.n, dtype=torch.uint8, pin_memory=pin)
        self.h_send_np, self.h_recv_np = self.h_send.numpy(), self.h_recv.numpy()
        if self.on_gpu:
            self.d_send = torch.empty(self.n, dtype=torch.uint8, device=device)
            self.d_recv = torch.empty(world * self.n, dtype=torch.uint8, device=device)
            self.stream = torch.cuda.Stream(device=device)
        if not self.single:
            src = self.d_recv if self.on_gpu else self.h_recv
            self.recv_views = [src[r * self.n:(r + 1) * self.n] for r in range(world)]

    def start(self, payload):
        if not self.collective:
            self._payload = bytes(payload)
            return
        assert self._work is None, "one exchange at a time"
        assert len(payload) == self.n
        torch, dist = self.torch, self.dist
        self._t0 = time.perf_counter()
        self.h_send_np[:] = np.frombuffer(payload, dtype=np.uint8)
        if self.on_gpu:
            with torch.cuda.stream(self.stream):
                self.d_send.copy_(self.h_send, non_blocking=True)
                if self.single:
                    self._work = dist.all_gather_into_tensor(self.d_recv, self.d_send, async_op=True)
                else:
                    self._work = dist.all_gather(self.recv_views, self.d_send, async_op=True)
        else:
            self._work = dist.all_gather(self.recv_views, self.h_send, async_op=True)

    def wait(self):
        if not self.collective:
            return [self._payload]
        torch = self.torch
        t1 = time.perf_counter()
        if self.on_gpu:
            with torch.cuda.stream(self.stream):
                self._work.wait()                     # (stream-side for nccl: the download below queues behind the collective)
                self.h_recv.copy_(self.d_recv, non_blocking=True)
            self.stream.synchronize()
        else:
            self._work.wait()
        self._work = None
        host = self.h_recv_np.tobytes()
        t2 = time.perf_counter()
        self.ms += (t2 - self._t0) * 1e3
        self.wait_ms += (t2 - t1) * 1e3
        self.calls += 1
        return [host[r * self.n:(r + 1) * self.n] for r in range(self.world)]

    def __call__(self, payload):
        self.start(payload)
        return self.wait()


def sharded_aggregate(provider, gather, rank, world, overlap=True):
    """begin -> exchange -> finish (-> once more in the rows-below-s0 corner).  Returns the rank's capi.ShardDecision:
    output slot k of the provider is global window first_slot_window_id + k, the first windows_owned of them are this rank's.
    provider:  begin(global_s0=None) -> record bytes ;  finish(records, rank) -> (rc, decision) ;  optionally
    pass_begin(record) - puts the rank's pass in flight while the records travel (overlap=False: the serial order, for A/B)."""
    rec = provider.begin()
    gather.start(rec)
    if overlap and hasattr(provider, "pass_begin"):
        provider.pass_begin(rec)
    recs = gather.wait()
    rc, d = provider.finish(recs, rank)
    if rc == RETRY:
        recs = gather(provider.begin(d.s0))
        rc, d = provider.finish(recs, rank)
        if rc != 0:
            raise RuntimeError("shard protocol did not settle after the second exchange")
    return d


def run_local(providers, overlap=True):
    """The protocol over ranks that live in ONE process (tests: K simulated ranks on one GPU; the all_gather is a list).
    Returns every rank's decision.  overlap: each rank's pass is put in flight from its own record (bowgpu_shard_pass_begin)
    before its finish - one thread holds one pass in flight, so begin-pass and finish alternate rank by rank here."""
    recs = [p.begin() for p in providers]
    res = []
    for r, p in enumerate(providers):
        if overlap and hasattr(p, "pass_begin"):
            p.pass_begin(recs[r])
        res.append(p.finish(recs, r))
    if any(rc == RETRY for rc, _ in res):
        assert all(rc == RETRY for rc, _ in res), "every rank must take the same decision from the same records"
        s0 = res[0][1].s0
        recs = [p.begin(s0) for p in providers]
        res = [p.finish(recs, r) for r, p in enumerate(providers)]
    assert all(rc == 0 for rc, _ in res)
    return [d for _, d in res]


class GpuProvider:
    """The product provider: HIP kernels through the C ABI (columns of any residency; outputs device-resident unless out_residency
    says otherwise - host-resident shards are staged through HBM per call, see include/bowgpu.h at bowgpu_shard_pass_begin)."""

    def __init__(self, cols, ts_col, interval, aggs, offset=0, out_capacity=None, strict_order=False, out_residency=None):
        from . import capi
        self.out_residency = capi.DEVICE if out_residency is None else out_residency
        self.capi = capi
        self.cols, self.ts_col, self.interval, self.aggs, self.offset = cols, ts_col, interval, aggs, offset
        self.n = cols[ts_col].length
        self.capacity = out_capacity
        self.outs = None
        self._carr = capi._cols(cols)
        self._aarr = capi._aggs(aggs)
        self._opts = capi.Options(offset, 0, int(bool(strict_order)))   # (strict_order: every window in the reference's row order, also across a shard boundary)
        self.info = capi.AggInfo()

    # ---- the protocol
    def begin(self, global_s0=None):
        capi = self.capi
        rec = capi.ShardRecord()
        s0 = C.byref(C.c_int64(global_s0)) if global_s0 is not None else None
        capi.check(capi.lib().bowgpu_shard_begin(self._carr, len(self.cols), self.ts_col, C.c_int64(self.interval),
                                                 C.byref(self._opts), self._aarr, len(self.aggs), s0, C.byref(rec)))
        self._f, self._l = rec.first_ts, rec.last_ts
        return bytes(rec)

    def pass_begin(self, record):
        """the pass over this rank's rows, enqueued from its own record (the exchange travels meanwhile); finish() collects it.
        Returns True when something was put in flight."""
        capi = self.capi
        rec = capi.ShardRecord.from_buffer_copy(bytes(record))
        cap = self.capacity or 0
        if rec.nrows > 0 and rec.last_ts >= rec.first_ts:
            cap = max(cap, (rec.last_ts - rec.first_ts) // self.interval + 3)
        self._pass_oarr = self._ensure_outs(max(cap, 1))
        rc = capi.lib().bowgpu_shard_pass_begin(self._carr, len(self.cols), self.ts_col, C.c_int64(self.interval), C.byref(self._opts),
                                                self._aarr, len(self.aggs), self._pass_oarr, C.byref(rec))
        if rc < 0:
            capi.check(rc)
        return rc == 0

    def finish(self, records, rank):
        capi = self.capi
        world = len(records)
        arr = (capi.ShardRecord * world)()
        for q, b in enumerate(records):
            C.memmove(C.byref(arr[q]), bytes(b), C.sizeof(capi.ShardRecord))
        d = plan(records, rank, self.interval, self.offset)
        oarr = self._ensure_outs(max(d.windows_local, 1))
        rc = capi.lib().bowgpu_shard_finish(self._carr, len(self.cols), self.ts_col, C.c_int64(self.interval), C.byref(self._opts),
                                            self._aarr, len(self.aggs), oarr, arr, world, rank, C.byref(d), C.byref(self.info))
        if rc < 0:
            capi.check(rc)
        if rc == 0:
            for i, o in enumerate(self.outs):
                o.absorb(oarr[i])
        return rc, d

    def _ensure_outs(self, need):
        capi = self.capi
        if self.outs is None or self.outs[0].slots < need:
            cap = max(need, self.capacity or 0)
            self.outs = [capi.OutColumn(cap, self.out_residency) for _ in self.aggs]
        oarr = (capi.Out * len(self.aggs))()
        for i, o in enumerate(self.outs):
            oarr[i] = o.c()
            oarr[i].length = o.slots
        return oarr

    # ---- the building blocks the protocol is made of (kept in the ABI; tests compare them with the fused path)
    def first_last_nrows(self):
        capi = self.capi
        if self.n == 0:
            return 0, 0, 0
        f, l, n = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        c = self.cols[self.ts_col].c()
        capi.check(capi.lib().bowgpu_shard_span(C.byref(c), C.byref(f), C.byref(l), C.byref(n)))
        self._f, self._l = f.value, l.value
        return self._f, self._l, self.n

    def shard_aggregate(self, s0, holds_row0, lead, next_row=None, finish_last=False):
        capi = self.capi
        if self.outs is None:
            self.first_last_nrows()
            need = (self._l - max(self._f, s0)) // self.interval + 2 + lead if self.n else 1
            self._ensure_outs(need)
        oarr = self._ensure_outs(1)
        carry = capi.ShardCarry()
        nr = capi.NextRow.from_buffer_copy(next_row) if next_row is not None else None
        capi.check(capi.lib().bowgpu_shard_aggregate(self._carr, len(self.cols), self.ts_col, C.c_int64(self.interval),
                                                     C.byref(self._opts), C.c_int64(s0), int(holds_row0), C.c_int64(lead),
                                                     self._aarr, len(self.aggs), oarr, C.byref(carry),
                                                     C.byref(nr) if nr is not None else None, int(bool(finish_last))))
        for i, o in enumerate(self.outs):
            o.absorb(oarr[i])
        return bytes(carry)

    def shard_carry_only(self, s0, holds_row0):
        """the carry shard_aggregate will return, from the rows of the last window alone"""
        capi = self.capi
        carry = capi.ShardCarry()
        capi.check(capi.lib().bowgpu_shard_carry_only(self._carr, len(self.cols), self.ts_col, C.c_int64(self.interval),
                                                      C.byref(self._opts), C.c_int64(s0), int(holds_row0),
                                                      self._aarr, len(self.aggs), C.byref(carry)))
        return bytes(carry)


class ShardedRolling:
    """bench.py's multi-GPU step: dense synthetic rows generated in this rank's HBM."""

    def __init__(self, rank, world, rows, interval, aggs, dist, torch, seed=42, offset=0, exchange_device="cuda", force_collective=False):
        from . import capi
        self.rank, self.world, self.interval, self.aggs = rank, world, interval, aggs
        ts, val = capi.gen_dense(rank * rows, rows, seed=seed)
        self.cols = [ts, val]
        cap = rows // interval + 3
        self.provider = GpuProvider(self.cols, 0, interval, aggs, offset=offset, out_capacity=cap)
        device = torch.device("cuda", torch.cuda.current_device()) if ((world > 1 or force_collective) and exchange_device == "cuda") else "cpu"
        self.gather = Gather(dist, torch, world, device, force_collective=force_collective)

    def step(self, overlap=True):
        d = sharded_aggregate(self.provider, self.gather, self.rank, self.world, overlap=overlap)
        self.decision = d
        self.first_slot, self.owned = d.first_slot_window_id, d.windows_owned
        return self.provider.info
