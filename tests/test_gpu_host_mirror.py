"""Runs the C++ mirror of the reference's interface (bow_amd/host/bow_rolling.hpp) against the reference's
own table-driven tests (tests/cpp/test_rolling.cpp) - through the C ABI, on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_mirror_replays_reference_tests():
    exe = os.path.join(ROOT, "tests", "cpp", "test_rolling")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "bow_amd", "host")])
    p = subprocess.run([exe, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=600)
    print(p.stdout[-4000:])
    print(p.stderr[-2000:])
    assert p.returncode == 0, p.stdout[-4000:]


@pytest.mark.gpu
def test_cpp_mirror_replays_reference_tests_through_the_fan_out():
    """the same unmodified binary with BOWGPU_DEVICES naming this box's device three times and one row per rank allowed: every table of the
    reference's tests that the record protocol serves runs as row ranges on worker threads (the rest falls back to the one-device path), and
    every expectation still holds"""
    exe = os.path.join(ROOT, "tests", "cpp", "test_rolling")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "bow_amd", "host")])
    env = dict(os.environ, BOWGPU_DEVICES="0,0,0", BOWGPU_FANOUT_MIN_ROWS="1")
    p = subprocess.run([exe, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=600, env=env)
    print(p.stdout[-4000:])
    print(p.stderr[-2000:])
    assert p.returncode == 0, p.stdout[-4000:]
    import re
    m = re.search(r"fan-out: (\d+) of (\d+) Rolling.Aggregate calls", p.stdout)
    assert m and int(m.group(2)) > 0 and int(m.group(1)) * 2 >= int(m.group(2)), p.stdout[-400:]   # (most tables are served; the declines fall back)


def test_the_initial_device_list_from_the_environment():
    # CPU: parsed once, no device needed to read it back; an explicit bowgpu_set_devices wins afterwards (needs a device: test_gpu_multi.py)
    import sys
    code = "from bow_amd import capi; print(capi.get_devices())"
    for val, want in (("0,1,2", "[0, 1, 2]"), ("3", "[]"), ("", "[]"), ("1,x", "[]"), ("2,2", "[2, 2]")):
        env = dict(os.environ, BOWGPU_DEVICES=val)
        out = subprocess.check_output([sys.executable, "-c", code], cwd=ROOT, env=env, text=True).strip()
        assert out == want, (val, out)


def test_cpp_mirror_builds():
    # CPU: the header-only mirror + its test compile and link against libbowgpu.so
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "bow_amd", "host")])
    assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "test_rolling"))
