"""bench.py as its own launcher: `python bench.py --gpus N` with no WORLD_SIZE in the environment must start N rank
processes itself (before anything touches the GPU in the parent), forward only rank 0's stdout, and fail when a rank dies."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run_launcher(tmp_path, body, n=3, timeout=60):
    stub = tmp_path / "stub.py"
    stub.write_text(textwrap.dedent(body))
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "raise SystemExit(bench.launch_ranks(%d, ['--x', '1'], script=%r, timeout=%d))" % (ROOT, n, str(stub), timeout))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=timeout + 30)


def test_launcher_forwards_rank0_only_and_sets_env(tmp_path):
    r = _run_launcher(tmp_path, """
        import json, os, sys
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
                         | {"argv": sys.argv[1:]}))
    """)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout           # rank 0's line is the only stdout line
    rec = json.loads(lines[0])
    assert rec["RANK"] == "0" and rec["LOCAL_RANK"] == "0" and rec["WORLD_SIZE"] == "3"
    assert rec["MASTER_ADDR"] == "127.0.0.1" and int(rec["MASTER_PORT"]) > 0
    assert rec["argv"] == ["--x", "1"]
    others = [json.loads(l) for l in r.stderr.splitlines() if l.startswith("{")]
    assert sorted(o["RANK"] for o in others) == ["1", "2"]
    assert {o["MASTER_PORT"] for o in others} == {rec["MASTER_PORT"]}


def test_launcher_dead_rank_fails_the_run(tmp_path):
    r = _run_launcher(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(30)      # the surviving ranks are stopped by the launcher, long before this ends
        print("{}")
    """, timeout=25)
    assert r.returncode == 7, (r.returncode, r.stderr)
    assert r.stdout.strip() == ""
    assert "rank 1 exited with code 7" in r.stderr


def test_bench_without_gpu_fails_loudly_for_n2():
    """No GPU in this container: both ranks must refuse (no CPU fallback), and the launcher must report it."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by test_bench_two_ranks_on_one_gpu")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "1e5", "--steps", "1",
                        "--warmup", "0", "--no-cpu"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "needs a GPU" in r.stderr


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """The driver's `python bench.py --gpus 2` form on a 1-GPU box: both ranks on GPU 0, exchange over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BOW_BENCH_SINGLE_DEVICE="1", BOW_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "1e7", "--steps", "3",
                        "--warmup", "1", "--no-cpu"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak"
    assert rec["config"]["parallelism"] == "rows range-partitioned x2"
    assert rec["value"] > 0 and "exchange_ms" in rec
