"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers run on the CPU build only - GPU ASan is
not available on this pool): the whole golden-vector suite and a seeded sweep over every entry point, in a child process that
preloads the sanitizer runtime.  A heap overflow, a misaligned / out-of-range access or signed overflow outside -fwrapv's
contract in the restatement would make every parity claim above it worthless."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWEEP = r"""
import numpy as np
from oracle import pyoracle as orc
rng = np.random.default_rng(5)
AGGS = ["WindowStart", "Sum", "ArithmeticMean", "Min", "Max", "Count", "First", "Last", "NumRows", "IntegralStep",
        "IntegralTrapezoid", "WeightedAverageStep", "WeightedAverageLinear", "Mode"]
for case in range(120):
    n = int(rng.integers(0, 400))
    ts = (np.cumsum(rng.integers(0, 6, n)) - int(rng.integers(0, 300))).astype(np.int64)
    pad = int(rng.integers(0, 20))
    vals = np.concatenate([np.zeros(pad), np.round(rng.standard_normal(n), 1), np.zeros(3)])
    valid = np.concatenate([np.ones(pad, bool), rng.random(n) > rng.choice([0.0, 0.3, 1.0]), np.ones(3, bool)])
    bm = np.packbits(valid, bitorder="little")
    tsb = np.concatenate([np.zeros(pad, np.int64), ts, np.zeros(3, np.int64)])
    cols = [orc.Column(tsb, None, orc.INT64, offset=pad, length=n), orc.Column(vals, bm, orc.FLOAT64, offset=pad, length=n)]
    interval = int(rng.choice([1, 3, 10, 100]))
    offset = int(rng.integers(-2 * interval, 2 * interval + 1))
    aggs = [("WindowStart", 0)] + [(AGGS[int(rng.integers(0, len(AGGS)))], int(rng.integers(0, 2))) for _ in range(5)]
    if n:
        orc.aggregate(cols, 0, interval, aggs, offset=offset)
        orc.aggregate_whole(cols, 0, aggs)
        ip = [{"kind": "WindowStart", "col": 0}, {"kind": ["Linear", "StepPrevious", "None"][case % 3], "col": 1}]
        orc.interpolate(cols, 0, interval, ip, offset=offset)
        orc.fill_linear(cols, 0, 1)
    for m in ("Previous", "Next", "Mean"):
        orc.fill(cols[1], m)
    orc.is_col_sorted(cols[1])
print("sweep ok")
"""


def _env():
    env = dict(os.environ)
    env["BOW_ORACLE_SANITIZED"] = "1"
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env["LD_PRELOAD"] = asan
    # python itself "leaks" by design; every other finding aborts the child
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1:halt_on_error=1"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


def test_golden_vectors_under_asan_ubsan():
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-q", "-x", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-3000:]


def test_seeded_sweep_under_asan_ubsan():
    p = subprocess.run([sys.executable, "-c", SWEEP], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "sweep ok" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-3000:]
