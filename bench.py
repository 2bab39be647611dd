#!/usr/bin/env python3
"""bench.py — rows/s of rolling ArithmeticMean (+WindowStart key) on dense int64-ts / float64-value
columns resident in HBM (BASELINE.json metric; SURVEY.md §8d cfg-dense: ts=i, value=u01, interval 10).

One "step" = one full Rolling.Aggregate(WindowStart(time), ArithmeticMean(value)) over the rank's
rows through the C ABI (bitmap init + bucketing/reduction kernel + status readback + null counts).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: rows are range-partitioned (rank r owns rows [r*R, (r+1)*R): weak scaling); the only exchange of a step is ONE
all_gather of a fixed-size record per rank (first / last timestamp + the running state of the rank's last window) over RCCL,
started BEFORE the rank's pass is enqueued and collected after (bowgpu_shard_begin / _pass_begin / _finish: the exchange is off
the critical path; `exchange_ms` = one exchange alone, `exchange_hidden_ms` = what the overlap saves per step against the serial
order, both measured after the timed region); the ownership rules live behind the C ABI.  Without a launcher (no WORLD_SIZE in
the environment) `--gpus N` starts its own N rank processes.  A preflight (device count, one exchange under a watchdog) turns a
rendezvous that cannot complete into a non-zero exit with a message instead of a hang.
Rank 0 prints ONE JSON line; it carries `parity_check`: the outputs of the timed call compared, outside the timed region, with
the oracle on the first and the last 2e6 rows of rank 0's rows bit for bit, plus the WindowStart progression of every window -
a mismatch is a non-zero exit.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_ROW = 16.0   # algorithmic read bytes/row: 8 (ts) + 8 (value); null_count == 0 => no bitmap
INTERVAL = 10
HEADLINE_ROWS = 1_000_000_000
# checksum64 (xor, sum) of the 1e8 means of the benched call at its full size: the value test_gpu_fullsize.py's
# test_headline_1e9_every_window_against_the_oracle obtains from outputs it has just compared with the oracle on EVERY window
HEADLINE_MEAN_CHECKSUM64 = "bed3073b06e80b6f8a90ff85ba2864c7"
# N > 1: same interval (same per-row work as N = 1), but Options.Offset = 3 so that every shard boundary
# (a multiple of 10 rows) falls INSIDE a window and the boundary-window stitch really runs
OFFSET_MULTI = 3


def cpu_baseline(capi, rows_sample):
    """The oracle (C restatement of the reference's algorithm, 1 thread — the reference is
    single-goroutine on this path) timed on a bounded sample of the same workload."""
    import numpy as np
    from oracle import pyoracle as orc
    ts_d, val_d = capi.gen_dense(0, rows_sample, seed=42)
    ts = ts_d.values.to_numpy(np.int64, rows_sample)
    val = val_d.values.to_numpy(np.float64, rows_sample)
    del ts_d, val_d
    cols = [orc.Column(ts, None, orc.INT64), orc.Column(val, None, orc.FLOAT64)]
    # repeat the pass over the sample until ~12 s of CPU work have been timed
    reps, dt = 0, 0.0
    while dt < 12.0 and reps < 200:
        t0 = time.perf_counter()
        orc.aggregate(cols, 0, INTERVAL, [("WindowStart", 0), ("ArithmeticMean", 1)])
        dt += time.perf_counter() - t0
        reps += 1
    return {"value": rows_sample * reps / dt, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": "%d passes over %d rows of the same dense workload (ts=i, value=u01, interval %d) with the "
                      "literal scan of oracle/bow_oracle.c (C restatement of the reference; a lower bound on the Go "
                      "reference's time), %.1f s" % (reps, rows_sample, INTERVAL, dt)}


def cpu_baseline_parallel(capi, rows_sample):
    """The same oracle on ALL host cores (SURVEY §8d "parallel mode"): the sample is cut into row ranges on window
    boundaries, one range per core, each range scanned by the single-threaded oracle (ctypes releases the GIL).  The
    reference itself is single-goroutine on this path: this is what a multi-core restatement of it could reach."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle as orc
    cores = os.cpu_count() or 1
    ts_d, val_d = capi.gen_dense(0, rows_sample, seed=42)
    ts = ts_d.values.to_numpy(np.int64, rows_sample)
    val = val_d.values.to_numpy(np.float64, rows_sample)
    del ts_d, val_d
    per = -(-rows_sample // cores)
    per += (-per) % INTERVAL  # ts = i: a multiple of the interval is a window boundary
    ranges = [(a, min(a + per, rows_sample)) for a in range(0, rows_sample, per)]

    def one(r):
        a, b = r
        cols = [orc.Column(ts[a:b], None, orc.INT64), orc.Column(val[a:b], None, orc.FLOAT64)]
        orc.aggregate(cols, 0, INTERVAL, [("WindowStart", 0), ("ArithmeticMean", 1)])

    reps, dt = 0, 0.0
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(one, ranges))  # warm-up (page faults of the output buffers)
        while dt < 5.0 and reps < 200:
            t0 = time.perf_counter()
            list(ex.map(one, ranges))
            dt += time.perf_counter() - t0
            reps += 1
    return {"value": rows_sample * reps / dt, "unit": "rows/s", "cores": min(cores, len(ranges)), "kind": "port",
            "sample": "%d passes over %d rows, %d row ranges cut on window boundaries, one oracle scan per range on a "
                      "thread pool, %.1f s" % (reps, rows_sample, len(ranges), dt)}


# the instantiation the benched call runs (one Float64 column without nulls, 10-row windows: the unpadded small-list form);
# tests/test_gpu_fullsize.py asserts it on the GPU, tests/test_profiles_fresh.py that the committed counters are its
BENCH_KERNEL_INSTANCE = "rolling_simple_kernel<0, false, false, false, false, false, false>"


def kernel_sha(instance):
    """identity of one rolling_simple_kernel instantiation as the build left it next to the library (bow_amd/csrc/kernel_sha.py:
    sha256 over the kernel's machine code + descriptor): ties a committed counter file to the code that is running.  Round 4 hashed
    three source files; an edit to common.h that changed nothing the kernel is compiled from made the evidence look stale and the
    driver's line lost its counter traffic."""
    try:
        with open(os.path.join(ROOT, "bow_amd", "libbowgpu.kernel_sha.json")) as fh:
            return json.load(fh)["kernels"].get(instance, {}).get("sha")
    except (OSError, ValueError, KeyError):
        return None


def newest_traffic_file():
    import glob
    import re
    files = glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic_bench_1e9.csv"))
    return max(files, key=lambda f: int(re.search(r"r(\d+)_pmc", os.path.basename(f)).group(1))) if files else None


def measured_traffic(rows, instance=None):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/r*_pmc_hbm_traffic_bench_1e9.csv:
    FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, in KB) - only when they were collected for this row count, for the instantiation that ran
    and from the machine code this library holds (each row carries the full kernel signature and the kernel's sha); None otherwise
    (PMC cannot be collected from inside the timed run)."""
    import csv
    f_ = newest_traffic_file()
    sha = kernel_sha(instance) if instance else None
    if rows != HEADLINE_ROWS or not f_ or not sha:
        return None
    f, w = [], []
    for r in csv.DictReader(open(f_)):
        if r.get("kernel_sha") != sha or instance not in r["kernel"]:
            continue  # counters of another build of the kernel / of another instantiation: not this run's traffic
        (f if r["counter"] == "FETCH_SIZE" else w).append(float(r["value_KB"]))
    if not f or not w:
        return None
    return (2.0 * sum(f) / len(f) + sum(w) / len(w)) * 1024.0


def host_pinned_rate(capi, dev_cols, sample, aggs):
    import numpy as np
    ts = capi.page_aligned(sample, np.int64)
    val = capi.page_aligned(sample, np.float64)
    ts[:] = dev_cols[0].values.to_numpy(np.int64, sample)
    val[:] = dev_cols[1].values.to_numpy(np.float64, sample)
    # first as they are - pageable, what a cgo caller holding plain Go-heap Arrow buffers passes: staged through HBM by the library
    pageable = None
    try:
        pcols = [capi.Column(ts), capi.Column(val)]
        Wp = capi.plan_windows(pcols[0], INTERVAL, 0)[1]
        pouts = [capi.OutColumn(Wp, capi.HOST) for _ in aggs]
        for _ in range(3):
            t0 = time.perf_counter()
            capi.rolling_aggregate(pcols, 0, INTERVAL, aggs, outs=pouts)
            dt = time.perf_counter() - t0
            pageable = dt if pageable is None or dt < pageable else pageable
        del pouts
    except Exception:
        pageable = None
    cols = [capi.Column(ts).pin(), capi.Column(val).pin()]
    try:
        W = capi.plan_windows(cols[0], INTERVAL, 0)[1]
        outs = [capi.OutColumn(W, capi.HOST_PINNED) for _ in aggs]
        best = None
        for _ in range(4):
            t0 = time.perf_counter()
            capi.rolling_aggregate(cols, 0, INTERVAL, aggs, outs=outs)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        moved = sample * BYTES_PER_ROW + 2 * 8.125 * W
        # ... and the SAME call with the library's multi-device fan-out on (bowgpu_set_devices: one call, row ranges over the listed
        # devices, each reading its range in place over its own host link).  On a one-GPU box the device is listed twice: the path
        # runs (two ranks, records in host memory, stitched outputs) but there is only one link to share.
        multi = None
        try:
            ndev = capi.device_count()
            ids = list(range(ndev)) if ndev > 1 else [0, 0]
            with capi.devices(ids):
                bm = None
                for _ in range(4):
                    t0 = time.perf_counter()
                    capi.rolling_aggregate(cols, 0, INTERVAL, aggs, outs=outs)
                    dt = time.perf_counter() - t0
                    bm = dt if bm is None or dt < bm else bm
                ranks = capi.last_call_ranks()
            multi = {"value": sample / bm, "unit": "rows/s", "ms_per_call": bm * 1e3, "rows": sample, "devices": ids, "ranks": ranks,
                     "pcie_gb_per_s": moved / bm / 1e9,
                     "what": "the same registered columns through ONE bowgpu_rolling_aggregate call with bowgpu_set_devices(%s) in force: "
                             "row ranges on one library thread per listed device, boundary windows stitched in row order, outputs placed "
                             "in the caller's buffers (bit-identical to the one-device call: tests/test_gpu_multi.py)" % ids}
        except Exception as e:
            multi = {"error": repr(e)}
        return {"multi": multi, "value": sample / best, "unit": "rows/s", "ms_per_call": best * 1e3, "rows": sample, "pcie_gb_per_s": moved / best / 1e9,
                "pageable_rows_per_s": (sample / pageable) if pageable else None,
                "pageable_pcie_gb_per_s": (moved / pageable / 1e9) if pageable else None,
                "what": "PCIe-inclusive: registered host columns read in place by the kernels (zero-copy) + outputs by DMA to registered "
                        "host buffers; bounded by the host link (PCIe Gen5 x16), not by the kernel.  pageable_*: the same call on "
                        "unregistered (pageable) buffers, staged through HBM by the library"}
    finally:
        for c in cols:
            c.unpin()


def parity_check(capi, outs, rows, row0, interval, offset, owned, first_slot, s0):
    """The outputs the timed steps produced (rank 0's), checked OUTSIDE the timed region: WindowStart of every owned window is the
    arithmetic progression; the windows of the first and of the last 2e6 rows against the oracle (the checker), bit for bit.
    Returns the `parity_check` object of the JSON line; ok == False makes the run fail."""
    aggs = [("WindowStart", 0), ("ArithmeticMean", 1)]
    res = {"ok": True, "windows_total": int(owned), "checked_against_oracle": 0, "progression_checked": 0}
    try:
        import numpy as np
        from oracle import pyoracle as orc   # (raises when oracle/libbow_oracle.so has not been built: reported below, the line is still printed)
        from bow_amd.sharded import first_window_start
        CH = 50_000_000
        for k0 in range(0, owned, CH):
            m = min(CH, owned - k0)
            ws = outs[0].values.to_numpy(np.int64, m, first=k0)
            if ws[0] != s0 + interval * (first_slot + k0) or not (np.diff(ws) == interval).all():
                raise AssertionError("WindowStart is not the arithmetic progression in slots [%d, %d)" % (k0, k0 + m))
            res["progression_checked"] += int(m)
        span = min(2_000_000, rows)
        for a in sorted({row0, row0 + rows - span}):
            ts_o, val_o = orc.gen_dense(a, span, seed=42)
            want, _ = orc.aggregate([orc.Column(ts_o, None, orc.INT64), orc.Column(val_o, None, orc.FLOAT64)], 0, interval, aggs, offset=offset)
            g0 = (first_window_start(a, interval, offset) - s0) // interval - first_slot   # output slot of the range's first window
            nw = want[0].length
            lo = 0 if a == row0 and row0 == 0 else 1                 # (a range cut out of the frame: its first window lacks rows)
            hi = nw if (a + span == row0 + rows and g0 + nw <= owned and offset == 0) else nw - 1
            lo, hi = max(lo, -g0), min(hi, owned - g0)
            for i in (0, 1):
                gv = outs[i].values.to_numpy(np.uint64, hi - lo, first=g0 + lo)
                if not np.array_equal(gv, want[i].values[:nw].view(np.uint64)[lo:hi]):
                    raise AssertionError("output %d differs from the oracle on rows [%d, %d)" % (i, a, a + span))
            res["checked_against_oracle"] += int(hi - lo)
        x, s = capi.checksum64(outs[1].values, owned)
        res["mean_checksum64"] = "%016x%016x" % (x, s)
        res["what"] = ("outputs of the timed steps, after the timed region: WindowStart progression over every owned window; windows "
                       "of the first and the last %d rows of rank 0 against oracle/bow_oracle.c, bit for bit" % span)
        # the benched configuration at its full size: every one of its 1e8 means is validated against the oracle by
        # tests/test_gpu_fullsize.py::test_headline_1e9_every_window_against_the_oracle, which pins this checksum of them
        if (rows, row0, interval, offset, first_slot) == (HEADLINE_ROWS, 0, INTERVAL, 0, 0) and owned == HEADLINE_ROWS // INTERVAL:
            res["mean_checksum64_expected"] = HEADLINE_MEAN_CHECKSUM64
            if res["mean_checksum64"] != HEADLINE_MEAN_CHECKSUM64:
                raise AssertionError("checksum of the %d means is %s, the oracle-validated value is %s" %
                                     (owned, res["mean_checksum64"], HEADLINE_MEAN_CHECKSUM64))
            res["what"] += "; checksum of all %d means equal to the value pinned by the every-window oracle test" % owned
    except Exception as e:   # (anything: a missing oracle library, a failed copy - the JSON line is still written, the run exits non-zero)
        res["ok"] = False
        res["error"] = "%s: %s" % (type(e).__name__, e)
    return res


def preflight(dist, torch, world, rank, backend, seconds=120):
    """N > 1, before anything is timed: every rank must see its device, and ONE exchange of the record-sized buffer must complete.
    A rendezvous that cannot (a rank on the wrong device, a transport that never connects) ends the run with a message and a
    non-zero exit code instead of hanging until the driver's limit."""
    import threading
    from bow_amd import sharded
    done = threading.Event()

    def watchdog():
        if not done.wait(seconds):
            sys.stderr.write("bench.py: rank %d: the preflight exchange (%s, world %d) did not complete within %d s - "
                             "giving up\n" % (rank, backend, world, seconds))
            sys.stderr.flush()
            os._exit(3)

    threading.Thread(target=watchdog, daemon=True).start()
    device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else "cpu"
    g = sharded.Gather(dist, torch, world, device, nbytes=64)
    got = g(bytes([rank % 251]) * 64)
    done.set()
    if [b[0] for b in got] != [r % 251 for r in range(world)]:
        raise SystemExit("bench.py: rank %d: the preflight all_gather returned the wrong bytes" % rank)


def _count(s):
    """row counts as the shell writes them: 100000000, 1e8, 2.5e7"""
    return int(float(s))


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, env=None, script=None, timeout=None):
    """`bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): this process becomes the launcher.
    It starts one fresh child per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as
    torch.distributed.run would) and never touches the GPU itself - nothing here imports torch.  Rank 0's stdout is the
    only thing forwarded to stdout (the one JSON line); the other ranks' stdout goes to stderr.  A rank that dies takes
    the others with it and the exit code is non-zero."""
    import subprocess
    import threading
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base.setdefault("MASTER_PORT", str(free_port()))
    base["WORLD_SIZE"] = str(n)
    base["LOCAL_WORLD_SIZE"] = str(n)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    script = script or os.path.abspath(__file__)
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=e,
                                      stdout=subprocess.PIPE, stderr=None, text=True))

    def pump(p, r):
        for line in p.stdout:
            (sys.stdout if r == 0 else sys.stderr).write(line)
            (sys.stdout if r == 0 else sys.stderr).flush()

    threads = [threading.Thread(target=pump, args=(p, r), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    rc = 0
    t_end = None if timeout is None else time.monotonic() + timeout
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, code))
                for q in live:
                    procs[q].terminate()   # exactly the PIDs started above
        if t_end is not None and time.monotonic() > t_end and live:
            sys.stderr.write("bench.py: ranks %s still running after %.0f s; stopping them\n" % (sorted(live), timeout))
            for q in live:
                procs[q].kill()
            rc = rc or 124
        time.sleep(0.05)
    for t in threads:
        t.join(timeout=5)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=_count, default=1_000_000_000, help="rows per GPU")
    ap.add_argument("--cpu-sample", type=_count, default=100_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity_check of the timed outputs against the oracle")
    ap.add_argument("--no-pinned", action="store_true",
                    help="skip the host_pinned aside (profiling: it launches the same kernel over PCIe, which would pollute per-kernel averages)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: be the launcher (before anything touches the GPU in this process)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # the ONE JSON line is the only thing this process may write to stdout: native libraries (gloo's "[Gloo] Rank 0 is connected
    # ..." goes to fd 1) write into stderr from here on, the line itself goes to the saved descriptor
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    from bow_amd import capi

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the bowgpu path has no CPU fallback")
    # BOW_BENCH_SINGLE_DEVICE=1 + BOW_BENCH_BACKEND=gloo: exercise the N>1 code path on a 1-GPU box (testing aid)
    if os.environ.get("BOW_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d needs GPU %d, this box has %d (BOW_BENCH_SINGLE_DEVICE=1 BOW_BENCH_BACKEND=gloo puts "
                         "every rank on GPU 0 to exercise the N>1 path on a 1-GPU box)" % (rank, local_rank, torch.cuda.device_count()))
    backend = os.environ.get("BOW_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    capi.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=300))
        preflight(dist, torch, world, rank, backend)

    rows = args.rows
    aggs = [("WindowStart", 0), ("ArithmeticMean", 1)]
    if world == 1:
        ts, val = capi.gen_dense(0, rows, seed=42)
        cols = [ts, val]
        s0, W = capi.plan_windows(ts, INTERVAL, 0)
        outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]

        def step():
            return capi.rolling_aggregate(cols, 0, INTERVAL, aggs, outs=outs)[1]
        interval = INTERVAL
    else:
        from bow_amd import sharded
        runner = sharded.ShardedRolling(rank, world, rows, INTERVAL, aggs, dist, torch, offset=OFFSET_MULTI,
                                        exchange_device="cuda" if backend == "nccl" else "cpu")
        step = runner.step
        interval = INTERVAL

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        capi.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if world > 1:
        runner.gather.ms, runner.gather.wait_ms, runner.gather.calls = 0.0, 0.0, 0
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        info = step()
        kernel_ms.append(info.kernel_ms)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    exch = None
    if world > 1:
        # after the timed region: the same steps in the SERIAL order (begin -> exchange -> pass) and the exchange alone
        g = runner.gather
        exch = {"window_ms": g.ms / max(g.calls, 1), "host_wait_ms": g.wait_ms / max(g.calls, 1), "per_step": g.calls / args.steps}
        ks = max(3, min(args.steps, 10))
        runner.step(overlap=False)
        barrier()
        t1 = time.perf_counter()
        for _ in range(ks):
            runner.step(overlap=False)
        barrier()
        ts_ = torch.tensor([(time.perf_counter() - t1) / ks], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(ts_, op=dist.ReduceOp.MAX)
        exch["serial_ms_per_step"] = float(ts_.item()) * 1e3
        rec = runner.provider.begin()
        g(rec)
        barrier()
        t1 = time.perf_counter()
        for _ in range(20):
            g(rec)
        te = torch.tensor([(time.perf_counter() - t1) / 20], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        exch["alone_ms"] = float(te.item()) * 1e3
        runner.step()          # (leave the outputs of an ordinary step behind for the parity check)
        barrier()

    parity = None
    if rank == 0 and not args.no_parity:
        # BEFORE the probes below (the read + write probe stores into the output buffers)
        if world == 1:
            parity = parity_check(capi, outs, rows, 0, INTERVAL, 0, W, 0, s0)
        else:
            d = runner.decision
            parity = parity_check(capi, runner.provider.outs, rows, 0, INTERVAL, OFFSET_MULTI, d.windows_owned, d.first_slot_window_id, d.s0)

    if rank == 0:
        total_rows = rows * world * args.steps
        value = total_rows / dt
        k_ms = sum(kernel_ms) / len(kernel_ms)
        kernel_name = capi.last_kernel_name()
        kernel_instance = capi.last_kernel_instance()
        achieved = rows * BYTES_PER_ROW / (k_ms * 1e-3) / 1e9
        line = {
            "metric": "rows/sec rolling-mean on 1B-row float64",
            "value": value,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "IntervalRolling(interval=%d, offset=%d)+Aggregate(WindowStart,ArithmeticMean), dense int64 ts=i + "
                                   "float64 u01 values, %d rows/GPU resident in HBM, null_count=0"
                                   % (interval, 0 if world == 1 else OFFSET_MULTI, rows),
                       "rows_per_gpu": rows, "windows_per_gpu": rows // interval, "parallelism": "rows range-partitioned x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": kernel_name, "kernel_instance": kernel_instance, "kernel_sha": kernel_sha(kernel_instance), "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": rows * BYTES_PER_ROW},
            "device": capi.device_name(),
            "parity_check": parity,
        }
        line["roofline"]["traffic"] = measured_traffic(rows, kernel_instance)
        if line["roofline"]["traffic"] is not None:
            line["roofline"]["traffic_source"] = "profiles/" + os.path.basename(newest_traffic_file()) + " (2 x FETCH_SIZE + WRITE_SIZE, same kernel_sha)"
        if world > 1:
            # THE exchange of a step (upload + all_gather of one 2424-byte record per rank + download, pinned buffers allocated once):
            # alone it takes exchange_ms; in the timed steps it is in flight while the pass runs, and the step is exchange_hidden_ms
            # shorter than the same step in the serial order
            line["exchange_ms"] = exch["alone_ms"]
            line["exchange_hidden_ms"] = max(0.0, exch["serial_ms_per_step"] - dt / args.steps * 1e3)
            line["exchange"] = {"alone_ms": exch["alone_ms"], "serial_ms_per_step": exch["serial_ms_per_step"],
                                "overlapped_ms_per_step": dt / args.steps * 1e3, "host_wait_ms_per_step": exch["host_wait_ms"],
                                "window_ms_per_step": exch["window_ms"], "exchanges_per_step": exch["per_step"],
                                "record_bytes": runner.gather.n, "buffers": "pinned host + device, allocated once"}
            line["exchanges_per_step"] = exch["per_step"]
        if world == 1:
            # the achievable line: a trivial streaming sum over the same two columns, measured in this run
            try:
                ceil = capi.stream_read_ceiling(cols[0].values, cols[1].values, rows * 8)
                line["roofline"]["stream_read_ceiling"] = {"value": ceil, "unit": "GB/s", "frac_of_ceiling": achieved / ceil}
            except Exception as e:
                line["roofline"]["stream_read_ceiling"] = {"error": repr(e)}
            # ... and for this TRAFFIC MIX: the same reads plus the same output bytes (two 8-byte slots per window) written in the
            # same pattern by a kernel that does no work (HBM writes are not free next to the reads: bus turnarounds)
            try:
                rw, rw_ms = capi.stream_rw_probe(cols[0].values, cols[1].values, rows * 8, outs[0].values, outs[1].values, interval)
                line["roofline"]["stream_rw_probe"] = {"value": rw, "unit": "GB/s", "ms": rw_ms, "product_over_probe": achieved / rw,
                                                       "what": "trivial kernel, same bytes read (16 B/row) and written (2 x 8 B/window), same store "
                                                               "pattern, no arithmetic: ONE launch shape, a probe of what the writes cost beside "
                                                               "the reads - not a bound (the product kernel is usually a little faster)"}
            except Exception as e:
                line["roofline"]["stream_rw_probe"] = {"error": repr(e)}
        if world == 1 and not args.no_pinned:
            # PCIe-inclusive aside (never `value`): the same call on HOST-resident columns of a 1e8-row sample - registered buffers
            # read in place by the kernels (BOWGPU_HOST_PINNED, zero-copy), outputs by DMA into registered buffers
            try:
                line["host_pinned"] = host_pinned_rate(capi, cols, min(rows, 100_000_000), aggs)
                line["host_pinned_multi"] = line["host_pinned"].pop("multi", None)
            except Exception as e:
                line["host_pinned"] = {"error": repr(e)}
        if not args.no_cpu and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(capi, min(args.cpu_sample, rows))
                # the Go reference itself would be the baseline of choice; it needs a Go toolchain plus the module's
                # un-vendored dependencies (arrow/go/v8, parquet-go), neither of which exists on these boxes
                import shutil
                line["cpu_baseline"]["go_reference"] = ("go toolchain found at %s, but github.com/metronlab/bow and its modules are not "
                                                        "available offline: not run" % shutil.which("go")) if shutil.which("go") \
                    else "not runnable on this box (no go toolchain)"
            except Exception as e:  # the baseline is a reported aside; never lose the GPU number over it
                line["cpu_baseline"] = {"error": repr(e)}
            try:
                line["cpu_baseline_all_cores"] = cpu_baseline_parallel(capi, min(args.cpu_sample, rows))
            except Exception as e:
                line["cpu_baseline_all_cores"] = {"error": repr(e)}
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
        if parity is not None and not parity["ok"]:
            sys.stderr.write("bench.py: PARITY CHECK FAILED: %s\n" % parity.get("error"))
            if dist is not None:
                dist.destroy_process_group()
            raise SystemExit(4)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
