"""RCCL executes at least once before the first multi-GPU run: every other GPU test and the 2-rank bench
aside use gloo, and Gather used to return early at world 1, so the `nccl` branch of bow_amd/sharded.py had never run anywhere.
One fresh child process (tests/rccl_world1_child.py), WORLD_SIZE=1, with exactly the environment bench.py's launcher gives its ranks
(HSA_ENABLE_IPC_MODE_LEGACY=0 included): init_process_group("nccl", device_id=...), the forced collective path of Gather, one
ShardedRolling.step() overlapped and one serial, bench.py's parity_check on both.  Still unexercised after this: the N > 1
rendezvous and the ring over xGMI (no multi-GPU box is available to the builder)."""
import json
import os
import subprocess
import sys

import pytest

import bench

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_world1_transport_and_one_sharded_step():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    # what launch_ranks() sets for a rank (bench.py): the same keys, through the same defaults
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = str(bench.free_port())
    env.update(WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_child.py")], env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, "exit %d\n--- stdout\n%s\n--- stderr\n%s" % (p.returncode, p.stdout[-3000:], p.stderr[-3000:])
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["ok"] and res["backend"] == "nccl" and res["gather_calls"] == 2 and res["exchanges_in_steps"] == 2
    assert res["env_HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert res["parity_overlap_True"]["mean_checksum64"] == res["parity_overlap_False"]["mean_checksum64"]
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "rccl_world1.json"), "w") as f:
            f.write(line + "\n")
