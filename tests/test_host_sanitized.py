"""The HOST side of libbowgpu.so under AddressSanitizer + UndefinedBehaviorSanitizer (GPU ASan is not available on this pool, so
the device code is covered by parity tests only; the host code is plain C++ and runs here).  The sanitized build
(make -C bow_amd/csrc asan) holds api.cpp / extras.cpp / parquet.cpp compiled with g++ and stubs for everything that lives in the
.hip files; a child process preloads the sanitizer runtime, loads it through BOWGPU_LIB and drives every entry point that works
without a GPU: plans and ctor errors on host-resident columns, aggregation / interpolation validation (which runs before any device
is touched), the shard plan on thousands of random layouts, carry merges, the Parquet footer and page-header parser on the
reference's files and on hostile footers."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_LIB = os.path.join(ROOT, "bow_amd", "libbowgpu_host_asan.so")

SWEEP = r"""
import ctypes as C, glob, os, sys
import numpy as np
from bow_amd import capi, sharded
L = capi.lib()
assert capi.LIB_PATH.endswith("libbowgpu_host_asan.so")
rng = np.random.default_rng(11)

# ---- plans / ctor errors on host columns (newIntervalRolling, enforceIntervalAndOffset, countWindows)
for case in range(400):
    n = int(rng.integers(0, 200))
    ts = (np.cumsum(rng.integers(0, 7, n)) - int(rng.integers(0, 500))).astype(np.int64)
    pad = int(rng.integers(0, 9))
    buf = np.concatenate([np.zeros(pad, np.int64), ts, np.zeros(2, np.int64)])
    valid = np.ones(len(buf), bool)
    if n and rng.random() < 0.3:
        valid[pad + int(rng.integers(0, n))] = False
    bm = np.packbits(valid, bitorder="little")
    col = capi.Column(buf, bm if rng.random() < 0.5 else None, capi.INT64, pad, n, -1)
    interval = int(rng.choice([-3, 0, 1, 5, 10, 1000, 1 << 40]))
    offset = int(rng.integers(-3000, 3000))
    try:
        s0, W = capi.plan_windows(col, interval, offset)
        p = capi.plan_windows_ex(col, interval, offset)
        assert (p.s0, p.num_windows) == (s0, W)
    except capi.BowGpuError as e:
        assert e.code in (-1, -3, -9), e
# a float64 interval column
try:
    capi.plan_windows(capi.Column(np.zeros(4), None, capi.FLOAT64), 5, 0)
    raise SystemExit("float64 interval column accepted")
except capi.BowGpuError as e:
    assert e.code == -2
out = C.c_int64(0)
for interval, off in [(5, 8), (5, -2), (5, 5), (7, -(1 << 62)), (1, 1 << 62)]:
    capi.check(L.bowgpu_enforce_interval_and_offset(C.c_int64(interval), C.c_int64(off), C.byref(out)))
    assert 0 <= out.value < interval

# ---- the CPU side of the pageable staging: a memcpy split over the helper threads, odd sizes and alignments, from two threads at once
import threading
def copies(seed):
    r = np.random.default_rng(seed)
    for n in [0, 1, 63, 4096, (1 << 20) - 1, (1 << 20) + 7, (4 << 20), (4 << 20) + 12345, 9_999_999]:
        a = r.integers(0, 256, n + 16, dtype=np.uint8)
        b = np.zeros(n + 16, dtype=np.uint8)
        off = int(r.integers(0, 16))
        capi.check(L.bowgpu_debug_host_copy(C.c_void_p(b.ctypes.data + off), C.c_void_p(a.ctypes.data + off), C.c_int64(n)))
        assert np.array_equal(b[off:off + n], a[off:off + n]) and not b[:off].any() and not b[off + n:].any(), n
th = [threading.Thread(target=copies, args=(s,)) for s in (1, 2)]
[t.start() for t in th]; [t.join() for t in th]

# ---- validation that runs before any device is touched (aggregation.go:147-188, interpolation.go:40-96)
ts = capi.Column(np.arange(10, dtype=np.int64))
val = capi.Column(np.arange(10, dtype=np.float64))
for aggs, code in [([], -4), ([("Sum", 1)], -5), ([("WindowStart", 0), ("Sum", 7)], -6)]:
    try:
        capi.rolling_aggregate([ts, val], 0, 5, aggs, outs=[capi.OutColumn(4) for _ in aggs])
        raise SystemExit("accepted %r" % (aggs,))
    except capi.BowGpuError as e:
        assert e.code == code, (aggs, e)
try:   # valid arguments: the call then needs a GPU and must say so (no CPU fallback)
    capi.rolling_aggregate([ts, val], 0, 5, [("WindowStart", 0), ("Sum", 1)])
    raise SystemExit("aggregate ran without a GPU")
except capi.BowGpuError as e:
    assert e.code == -11, e
for ip, code in [([{"kind": "Linear", "col": 1}], -5), ([{"kind": "WindowStart", "col": 1}, {"kind": "WindowStart", "col": 0}], -7)]:
    try:
        capi.rolling_interpolate([ts, val], 0, 5, ip)
        raise SystemExit("accepted %r" % (ip,))
    except capi.BowGpuError as e:
        assert e.code == code, (ip, e)

# ---- the shard plan (pure host arithmetic) on random layouts, negative timestamps and offsets included
def rec(f, l, n, interval, offset, flags=0):
    r = capi.ShardRecord()
    r.nrows, r.flags = n, flags
    if n:
        r.first_ts, r.last_ts = f, l
        off = offset % interval
        r.carry_from_ts = (l - off) // interval * interval + off
    return bytes(r)
for case in range(3000):
    world = int(rng.integers(1, 9))
    interval = int(rng.choice([1, 3, 7, 10, 100, 1000, 1 << 33]))
    offset = int(rng.integers(-3 * min(interval, 1 << 20), 3 * min(interval, 1 << 20)))
    t = int(rng.integers(-5000, 5000))
    recs = []
    for _ in range(world):
        if rng.random() < 0.2:
            recs.append(rec(0, 0, 0, interval, offset)); continue
        t += int(rng.choice([0, 1, 2, 9, 5 * min(interval, 1 << 20) + 3]))
        f = t
        n = int(rng.integers(1, 50))
        t += int(rng.integers(0, 4 * min(interval, 1 << 20) + 1)) if n > 1 else 0
        recs.append(rec(f, t, n, interval, offset))
    ds = [sharded.plan(recs, r, interval, offset) for r in range(world)]
    W = ds[0].num_windows
    if not any(d.retry_with_s0 for d in ds):
        seen = np.zeros(max(W, 0), dtype=int)
        for d in ds:
            if d.first_slot_window_id >= 0:
                seen[d.first_slot_window_id:d.first_slot_window_id + d.windows_owned] += 1
        assert (seen == 1).all(), (case, recs)
try:
    sharded.plan([rec(0, 50, 10, 10, 0), rec(40, 90, 10, 10, 0)], 0, 10)
    raise SystemExit("out-of-order ranks accepted")
except capi.BowGpuError as e:
    assert e.code == -14

# ---- carry merges
a, b, o = capi.CarryState(), capi.CarryState(), capi.CarryState()
for case in range(500):
    for s in (a, b):
        s.has_value = int(rng.random() < 0.7); s.has_nn = s.has_value; s.has_point = int(rng.random() < 0.6); s.has_pair = s.has_point
        s.sum, s.vmin, s.vmax = float(rng.standard_normal()), -1.0, 2.0
        s.nn_min, s.nn_max = s.vmin, s.vmax
        s.count, s.nrows = int(rng.integers(0, 9)), int(rng.integers(0, 9))
        s.pt, s.pv, s.first_pt, s.first_pv = 3.0, 1.5, 1.0, 0.5
    capi.check(L.bowgpu_carry_merge(C.byref(a), C.byref(b), C.byref(o)))
    assert o.nrows == a.nrows + b.nrows

# ---- Parquet: the reference's own files, and hostile footers
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.parquet"))):
    f = capi.ParquetFile(path)
    assert f.num_rows > 0 and len(f.columns) >= 2
    f.close()
import tempfile
hostile = [bytes([0x15, 0x02, 0x18]) + b"\xff" * 9 + b"\x01", bytes([0x19]) * 201 + b"\x00", bytes([0x1c]) * 100000,
           bytes([0x1b]) + b"\xff" * 8 + b"\x3f" + bytes([0x11]), bytes([0x19, 0xf5]) + b"\xff" * 8 + b"\x0f", b"", b"\x00" * 64]
hostile += [bytes(rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8)) for _ in range(300)]
with tempfile.TemporaryDirectory() as d:
    for i, footer in enumerate(hostile):
        p = os.path.join(d, "h%d.parquet" % i)
        with open(p, "wb") as fh:
            fh.write(b"PAR1" + footer + len(footer).to_bytes(4, "little") + b"PAR1")
        try:
            capi.ParquetFile(p).close()     # (a random footer may even parse: what matters is that nothing is read out of bounds)
        except capi.BowGpuError:
            pass
    # truncations of a real file
    blob = open(sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.parquet")))[0], "rb").read()
    for cut in list(range(0, 64)) + [len(blob) // 2, len(blob) - 9, len(blob) - 1]:
        p = os.path.join(d, "t%d.parquet" % cut)
        with open(p, "wb") as fh:
            fh.write(blob[:cut])
        try:
            capi.ParquetFile(p).close()
        except capi.BowGpuError:
            pass
print("host sweep ok")
"""


def _env():
    env = dict(os.environ)
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    ubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    env["LD_PRELOAD"] = asan + (" " + ubsan if os.path.exists(ubsan) else "")
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1:halt_on_error=1:protect_shadow_gap=0"   # (python itself leaks by design)
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    env["BOWGPU_LIB"] = ASAN_LIB
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


def test_host_side_of_the_library_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "bow_amd", "csrc"), "asan", "-s"])
    code = "import os\nROOT = %r\n" % ROOT + SWEEP
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "host sweep ok" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]
