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
    assert rec["value"] > 0 and rec["exchange_ms"] > 0 and rec["exchange_hidden_ms"] >= 0
    assert rec["exchange"]["exchanges_per_step"] == 1 and rec["exchange"]["record_bytes"] > 2000
    assert rec["parity_check"]["ok"] is True and rec["parity_check"]["checked_against_oracle"] > 100_000


@pytest.mark.gpu
def test_bench_line_contract_single_gpu():
    """`python bench.py` at N = 1 (a reduced row count, everything else as the default run): ONE JSON line on stdout with the keys of
    the bench contract - the metric of BASELINE.json, the roofline object (live kernel time on the library's stream, the in-run
    ceilings) and the cpu_baseline object - and values that hang together."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "2e7", "--steps", "4", "--warmup", "1",
                        "--cpu-sample", "2e6"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert rec["metric"] == "rows/sec rolling-mean on 1B-row float64" and rec["metric"] in base["metric"]
    assert rec["unit"] == "rows/s" and rec["n_gpus"] == 1 and rec["steps"] == 4 and rec["warmup"] == 1
    assert rec["higher_is_better"] is True and rec["scaling"] == "weak" and rec["vs_baseline"] is None
    assert rec["dtype"] == "f64" and rec["data"] == "synthetic" and "workload" in rec["config"] and "model" not in rec["config"]
    assert abs(rec["value"] - 2e7 / (rec["ms_per_step"] * 1e-3)) / rec["value"] < 1e-6
    roof = rec["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert abs(roof["achieved"] - 2e7 * 16 / (roof["kernel_ms"] * 1e-3) / 1e9) / roof["achieved"] < 1e-6
    assert roof["kernel"] == "rolling_simple_kernel" and 0 < roof["kernel_ms"] <= rec["ms_per_step"]
    assert roof["traffic"] is None          # (PMC traffic is committed for the 1e9-row launch only)
    assert roof["stream_read_ceiling"]["value"] > 0 and roof["stream_rw_probe"]["value"] > 0
    cpu = rec["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["unit"] == "rows/s" and cpu["value"] > 0 and cpu["sample"]
    assert rec["host_pinned"]["value"] > 0 and rec["host_pinned"]["pageable_rows_per_s"] > 0
    par = rec["parity_check"]
    assert par["ok"] is True and par["progression_checked"] == 2_000_000 and par["checked_against_oracle"] >= 2_000_000 // 10 * 2 - 2
