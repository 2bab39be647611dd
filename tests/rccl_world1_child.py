"""Child process of tests/test_gpu_rccl_world1.py: RCCL at world 1 on the box's one GPU.

Started FRESH (nothing has touched the GPU in this process before init_process_group) with the environment bench.py's launcher
builds for its ranks.  Drives the real transport of bow_amd/sharded.py - Gather with force_collective=True: pinned host pair,
device pair, side stream, all_gather_into_tensor(async_op=True), stream-side wait, D2H - and then ShardedRolling.step() through it,
overlapped and serial, checked with bench.py's parity_check.  A failed init / a hang never outlives the watchdog: os._exit, no re-exec.
Prints one JSON line; exit code 0 = everything ran and matched."""
import datetime
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    done = threading.Event()
    limit = float(os.environ.get("BOW_RCCL_CHILD_LIMIT", "240"))

    def watchdog():
        if not done.wait(limit):
            sys.stderr.write("rccl_world1_child: not finished after %.0f s (RCCL init or a collective hangs) - giving up\n" % limit)
            sys.stderr.flush()
            os._exit(3)

    threading.Thread(target=watchdog, daemon=True).start()
    assert os.environ.get("WORLD_SIZE") == "1" and os.environ.get("RANK") == "0"
    res = {"env_HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    import torch
    import torch.distributed as dist
    try:
        torch.cuda.set_device(0)
        t0 = time.perf_counter()
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=120))
        res["init_s"] = time.perf_counter() - t0
        res["backend"] = dist.get_backend()
    except Exception as e:   # the RCCL error text, a non-zero exit - never a hang
        sys.stderr.write("rccl_world1_child: init_process_group(nccl) failed: %s: %s\n" % (type(e).__name__, e))
        sys.stderr.flush()
        os._exit(2)
    from bow_amd import capi, sharded
    import bench
    capi.set_device(0)
    dev = torch.device("cuda", 0)
    # 1. the transport alone: the record-sized exchange, twice (buffers are allocated once), bytes back unchanged
    g = sharded.Gather(dist, torch, 1, dev, force_collective=True)
    assert g.collective and g.on_gpu and g.single and g.h_send.is_pinned() and g.h_recv.is_pinned()
    for k in range(2):
        payload = bytes([(7 * i + k) % 251 for i in range(g.n)])
        g.start(payload)
        assert g._work is not None            # really in flight: an async collective on the side stream
        got = g.wait()
        assert got == [payload], "all_gather_into_tensor returned other bytes"
    res["gather_calls"], res["gather_ms_per_call"] = g.calls, g.ms / g.calls
    # 2. one sharded step through it (world 1: this rank holds global row 0 and owns every window), overlapped and serial
    rows, interval, offset = 2_000_000, bench.INTERVAL, bench.OFFSET_MULTI
    aggs = [("WindowStart", 0), ("ArithmeticMean", 1)]
    runner = sharded.ShardedRolling(0, 1, rows, interval, aggs, dist, torch, offset=offset, exchange_device="cuda", force_collective=True)
    assert runner.gather.collective and runner.gather.single
    sums = []
    for overlap in (True, False):
        runner.step(overlap=overlap)
        d = runner.decision
        par = bench.parity_check(capi, runner.provider.outs, rows, 0, interval, offset, d.windows_owned, d.first_slot_window_id, d.s0)
        assert par["ok"], par
        sums.append(par["mean_checksum64"])
        res["parity_overlap_%s" % overlap] = {k: par[k] for k in ("ok", "windows_total", "checked_against_oracle", "mean_checksum64")}
    assert sums[0] == sums[1], "overlapped and serial steps differ"
    res["exchanges_in_steps"] = runner.gather.calls
    assert runner.gather.calls == 2
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    done.set()
    res["ok"] = True
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
