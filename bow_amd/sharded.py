"""Row-range sharded Rolling.Aggregate across the GPUs of one node.

One process per GPU; rank r holds rows [r*R, (r+1)*R) of every column.  Windows are disjoint row
ranges, so every rank reduces its own rows with no data-path collective.  The only exchange is
bookkeeping, as bytes through torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests):

  1. all_gather of (first_ts, last_ts, nrows) + each shard's first row (232 B/rank) so every rank knows all window
     ranges and, for inclusive windows, the row its last window may still need from the right; the first window start s0
     is host arithmetic on the first timestamp of the rank holding global row 0 (first_window_start)
  2. all_gather of each rank's carry: the running state of its LAST window            (~1.5 KB/rank)

A window that straddles a shard boundary is finished by the right rank: it re-walks its own rows
of that window seeded with the left rank's carry, i.e. in the reference's row order (bit-exact for
two ranks per window; windows spanning three or more ranks merge partial sums).  The left rank drops
that window from its output.

The compute lives behind a small provider interface so the protocol can be tested on CPU under
gloo with a numpy provider (tests/test_sharded_gloo.py); the product provider is GpuProvider
(HIP kernels through the C ABI, no CPU path).
"""
import ctypes as C

import numpy as np


class ShardPlan:
    """What every rank knows after step 2."""

    def __init__(self, s0, interval, firsts, lasts, nrows):
        self.s0, self.interval = s0, interval
        self.wf, self.wl = [], []
        for f, l, n in zip(firsts, lasts, nrows):
            if n == 0 or l < s0:
                self.wf.append(-1)
                self.wl.append(-1)
            else:
                ff = max(f, s0)
                self.wf.append((ff - s0) // interval)
                self.wl.append((l - s0) // interval)
        self.world = len(self.wf)

    def left_nonempty(self, r):
        q = r - 1
        while q >= 0 and self.wf[q] < 0:
            q -= 1
        return q

    def right_nonempty(self, r):
        q = r + 1
        while q < self.world and self.wf[q] < 0:
            q += 1
        return q if q < self.world else -1

    def lead_empty(self, r):
        """empty windows between the left neighbour's last window and this shard's first one"""
        if self.wf[r] < 0:
            return 0
        q = self.left_nonempty(r)
        if q < 0:
            return self.wf[r]  # nothing to the left: windows 0..wf-1 cannot exist (row 0 is in window 0) => 0
        return max(0, self.wf[r] - self.wl[q] - 1)

    def seed_ranks(self, r):
        """ranks (ascending) whose rows belong to this shard's FIRST window"""
        if self.wf[r] < 0:
            return []
        out = []
        q = self.left_nonempty(r)
        while q >= 0 and self.wl[q] == self.wf[r]:
            out.append(q)
            if self.wf[q] != self.wf[r]:
                break  # q only contributes its tail
            q = self.left_nonempty(q)
        return out[::-1]

    def drops_last(self, r):
        """this shard's last window continues on a rank to the right, which owns its output"""
        if self.wf[r] < 0:
            return False
        q = self.right_nonempty(r)
        return q >= 0 and self.wf[q] == self.wl[r]


def first_window_start(first_ts, interval, offset=0):
    """s0 of rolling.IntervalRolling from the first timestamp alone (rolling.go:96-99 + enforceIntervalAndOffset :114-128),
    with Go's integer semantics (division and % truncate toward zero).  Host arithmetic on three scalars: every rank derives
    it from the gathered first timestamp of the rank that holds global row 0, so no separate broadcast is needed."""
    if interval <= 0:
        raise ValueError("strictly positive interval required")
    if offset >= interval or offset <= -interval:
        offset = offset - int(offset / interval) * interval   # Go's %: sign of the dividend
    if offset < 0:
        offset += interval
    q = abs(first_ts) // interval
    if first_ts < 0:
        q = -q
    s0 = q * interval + offset
    if s0 > first_ts:
        s0 -= interval
    return s0


def _gather_bytes(dist, torch, payload, world, device):
    """all_gather of one fixed-size record per rank, as bytes.  One upload, one collective into ONE tensor, one download: with a
    list of per-rank output tensors every rank's record comes back through a copy + synchronisation of its own, and at 8 ranks
    those cost more than the collective (records are a few hundred bytes; a step of the benched workload is 3 ms)."""
    t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    n = t.numel()
    out = torch.empty(world * n, dtype=torch.uint8, device=device)
    try:
        dist.all_gather_into_tensor(out, t)
    except (RuntimeError, NotImplementedError, AttributeError):   # a backend without the single-tensor form
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        out = torch.cat(outs)
    host = out.cpu().numpy().tobytes()
    return [host[r * n:(r + 1) * n] for r in range(world)]


class ShardSession:
    """The protocol split at its two exchanges, so it can be driven by torch.distributed
    (sharded_aggregate below) or by an in-process loop over simulated ranks (tests)."""

    def __init__(self, provider, rank, world, interval):
        self.provider, self.rank, self.world, self.interval = provider, rank, world, interval
        self.plan = None

    def local_info(self):
        """(first_ts, last_ts, nrows) + this shard's first row (for the inclusive windows of the rank to the left), as bytes"""
        f, l, n = self.provider.first_last_nrows()
        rec = np.array([f, l, n], dtype=np.int64).tobytes()
        if hasattr(self.provider, "first_row_record"):
            rec += self.provider.first_row_record()
        return rec

    def _next_row(self, all_info):
        """first-row record of the next non-empty rank to the right (None: there is none / the provider has no such records)"""
        q = self.plan.right_nonempty(self.rank)
        if q < 0 or len(all_info[q]) <= 24:
            return None
        return all_info[q][24:]

    def phase1(self, s0, all_info):
        arr = [np.frombuffer(b[:24], dtype=np.int64) for b in all_info]
        self.s0 = s0
        self.plan = ShardPlan(s0, self.interval, [int(a[0]) for a in arr], [int(a[1]) for a in arr], [int(a[2]) for a in arr])
        self.lead = self.plan.lead_empty(self.rank)
        self.next_row = self._next_row(all_info)
        plan, rank = self.plan, self.rank
        # the shard folds the next shard's first row into its last window itself when it owns that window and the window is
        # not also its first one shared with ranks to the left (then phase 2 does it, after the seeds)
        owns_last = plan.wf[rank] >= 0 and not plan.drops_last(rank)
        seeded_single = plan.wf[rank] == plan.wl[rank] and bool(plan.seed_ranks(rank))
        if self.next_row is not None:
            return self.provider.shard_aggregate(s0, rank == 0, self.lead, self.next_row, owns_last and not seeded_single)
        return self.provider.shard_aggregate(s0, rank == 0, self.lead)  # carry bytes

    def phase2(self, all_carries):
        from . import capi
        plan, rank = self.plan, self.rank
        seeds = plan.seed_ranks(rank)
        if seeds:
            off = capi.ShardCarry.last.offset
            sz = C.sizeof(capi.CarryState) * capi.CARRY_MAX_AGGS
            state = all_carries[seeds[0]][off:off + sz]
            for q in seeds[1:]:
                state = self.provider.merge(state, all_carries[q][off:off + sz])
            if self.next_row is not None and plan.wf[rank] == plan.wl[rank] and not plan.drops_last(rank):
                self.provider.fix_first(self.s0, self.lead, plan.wf[rank], state, self.next_row)
            else:
                self.provider.fix_first(self.s0, self.lead, plan.wf[rank], state)
        W_local = 0 if plan.wf[rank] < 0 else plan.wl[rank] - plan.wf[rank] + 1 + self.lead
        owned = W_local - (1 if plan.drops_last(rank) else 0)
        first_slot = -1 if plan.wf[rank] < 0 else plan.wf[rank] - self.lead
        return first_slot, owned


def sharded_aggregate(provider, dist, torch, rank, world, interval, device="cpu"):
    """Runs the protocol over torch.distributed.  Returns (first_slot_window_id, n_windows_owned, plan).
    provider:
       first_last_nrows() -> (first_ts, last_ts, nrows)
       offset (attribute, optional)                      Options.Offset as given by the caller
       shard_aggregate(s0, holds_row0, lead) -> carry bytes (ShardCarry layout)
       fix_first(s0, lead, first_window_id, seed_bytes) -> merged carry-state bytes
       merge(a_bytes, b_bytes) -> bytes                  (array of CarryState, one per aggregator)
    """
    sess = ShardSession(provider, rank, world, interval)
    # 1. every rank's (first_ts, last_ts, nrows); s0 follows from the first timestamp of the rank holding global row 0
    mine = sess.local_info()
    all_info = _gather_bytes(dist, torch, mine, world, device) if world > 1 else [mine]
    s0 = 0
    for b in all_info:
        f, _, n = np.frombuffer(b[:24], dtype=np.int64)
        if n > 0:
            s0 = first_window_start(int(f), interval, getattr(provider, "offset", 0))
            break
    carry = sess.phase1(s0, all_info)
    # 2. carries.  (Putting this exchange in flight before the pass - the carry of a call without inclusive reducers follows from
    # the last window's rows alone, bowgpu_shard_carry_only - was measured: the early carry costs 0.06 ms, about what an RCCL
    # all_gather of a few hundred bytes does, so the plain order stays.)
    carries = _gather_bytes(dist, torch, carry, world, device) if world > 1 else [carry]
    first_slot, owned = sess.phase2(carries)
    return first_slot, owned, sess.plan


class GpuProvider:
    """The product provider: HIP kernels through the C ABI (device-resident columns and outputs)."""

    def __init__(self, cols, ts_col, interval, aggs, offset=0, out_capacity=None):
        from . import capi
        self.capi = capi
        self.cols, self.ts_col, self.interval, self.aggs, self.offset = cols, ts_col, interval, aggs, offset
        n = cols[ts_col].length
        # upper bound of local windows: span / interval + 2 (+ lead, bounded the same way); caller may pass capacity
        self.capacity = out_capacity
        self.outs = None
        self._carr = capi._cols(cols)
        self._aarr = capi._aggs(aggs)
        self._opts = capi.Options(offset, 0, 0)
        self.n = n
        self.has_inclusive = any(a[0] in ("IntegralTrapezoid", "WeightedAverageLinear") for a in aggs)

    def first_last_nrows(self):
        capi = self.capi
        ts = self.cols[self.ts_col]
        if self.n == 0:
            return 0, 0, 0
        f, l, n = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        c = ts.c()
        capi.check(capi.lib().bowgpu_shard_span(C.byref(c), C.byref(f), C.byref(l), C.byref(n)))
        self._f, self._l = f.value, l.value
        return self._f, self._l, self.n

    def plan_s0(self):
        return self.capi.plan_windows(self.cols[self.ts_col], self.interval, self.offset)[0]

    def _ensure_outs(self, s0, lead):
        capi = self.capi
        if self.outs is None:
            cap = self.capacity
            if cap is None:
                cap = (self._l - max(self._f, s0)) // self.interval + 2 + lead if self.n else 1
            self.outs = [capi.OutColumn(cap, capi.DEVICE) for _ in self.aggs]
        oarr = (capi.Out * len(self.aggs))()
        for i, o in enumerate(self.outs):
            oarr[i] = o.c()
        return oarr

    def first_row_record(self):
        capi = self.capi
        rec = capi.NextRow()
        if not any(a[0] in ("IntegralTrapezoid", "WeightedAverageLinear") for a in self.aggs):
            return bytes(rec)  # exclusive windows: nobody needs this shard's first row (present = 0, no device access)
        capi.check(capi.lib().bowgpu_shard_first_row(self._carr, len(self.cols), self.ts_col, self._aarr, len(self.aggs), C.byref(rec)))
        return bytes(rec)

    def shard_aggregate(self, s0, holds_row0, lead, next_row=None, finish_last=False):
        capi = self.capi
        oarr = self._ensure_outs(s0, lead)
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
        """the carry shard_aggregate will return, from the rows of the last window alone (no inclusive reducers)"""
        capi = self.capi
        carry = capi.ShardCarry()
        capi.check(capi.lib().bowgpu_shard_carry_only(self._carr, len(self.cols), self.ts_col, C.c_int64(self.interval),
                                                      C.byref(self._opts), C.c_int64(s0), int(holds_row0),
                                                      self._aarr, len(self.aggs), C.byref(carry)))
        return bytes(carry)

    def fix_first(self, s0, lead, first_window_id, seed_bytes, next_row=None):
        capi = self.capi
        nr = capi.NextRow.from_buffer_copy(next_row) if next_row is not None else None
        oarr = self._ensure_outs(s0, lead)
        for i, o in enumerate(self.outs):
            oarr[i].length = o.slots
        seeds = (capi.CarryState * capi.CARRY_MAX_AGGS).from_buffer_copy(seed_bytes)
        merged = (capi.CarryState * capi.CARRY_MAX_AGGS)()
        capi.check(capi.lib().bowgpu_shard_fix_first(self._carr, len(self.cols), self.ts_col, C.c_int64(self.interval),
                                                     C.byref(self._opts), C.c_int64(s0), C.c_int64(lead), self._aarr,
                                                     len(self.aggs), oarr, C.c_int64(first_window_id), seeds, merged,
                                                     C.byref(nr) if nr is not None else None))
        for i, o in enumerate(self.outs):
            o.absorb(oarr[i])
        return bytes(merged)

    def merge(self, a_bytes, b_bytes):
        capi = self.capi
        a = (capi.CarryState * capi.CARRY_MAX_AGGS).from_buffer_copy(a_bytes)
        b = (capi.CarryState * capi.CARRY_MAX_AGGS).from_buffer_copy(b_bytes)
        out = (capi.CarryState * capi.CARRY_MAX_AGGS)()
        for i in range(len(self.aggs)):
            capi.check(capi.lib().bowgpu_carry_merge(C.byref(a[i]), C.byref(b[i]), C.byref(out[i])))
        return bytes(out)


class ShardedRolling:
    """bench.py's multi-GPU step: dense synthetic rows generated in this rank's HBM."""

    def __init__(self, rank, world, rows, interval, aggs, dist, torch, seed=42, offset=0, exchange_device="cuda"):
        from . import capi
        self.rank, self.world, self.interval, self.aggs = rank, world, interval, aggs
        self.dist, self.torch = dist, torch
        ts, val = capi.gen_dense(rank * rows, rows, seed=seed)
        self.cols = [ts, val]
        cap = rows // interval + 3
        self.provider = GpuProvider(self.cols, 0, interval, aggs, offset=offset, out_capacity=cap)
        self.device = torch.device("cuda", torch.cuda.current_device()) if (world > 1 and exchange_device == "cuda") else "cpu"
        self.kernel_timer = capi.Timer()

    def step(self):
        from . import capi
        first_slot, owned, plan = sharded_aggregate(self.provider, self.dist, self.torch, self.rank, self.world,
                                                    self.interval, device=self.device)
        info = capi.AggInfo()
        info.kernel_ms = capi.last_kernel_ms()
        self.first_slot, self.owned = first_slot, owned
        return info
