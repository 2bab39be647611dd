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


def test_cpp_mirror_builds():
    # CPU: the header-only mirror + its test compile and link against libbowgpu.so
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "bow_amd", "host")])
    assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "test_rolling"))
