"""CPU-side checks of the drop-in boundary: libbowgpu.so loads, exports every symbol that
include/bowgpu.h declares, the O(1) planning entry points work on host buffers without a GPU,
and — on a box with no GPU — the compute entry points fail loudly instead of falling back."""
import os
import re

import pytest

from bow_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "bowgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bowgpu_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    L = capi.lib()
    names = header_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(capi.SYMBOLS) == names
    assert L.bowgpu_abi_version() == capi.ABI_VERSION == int(re.search(r'#define BOWGPU_ABI_VERSION (\d+)', open(os.path.join(ROOT, 'include', 'bowgpu.h')).read()).group(1))


def test_plan_on_host_buffers(golden):
    for o in golden["offsets"]:
        assert capi.enforce_interval_and_offset(o["interval"], o["offset"]) == o["norm"]
    for v in golden["num_windows"]:
        ts = capi.Column.from_list(v["time"], "int64")
        assert capi.plan_windows(ts, v["interval"], v["offset"])[1] == v["W"], v["name"]
    for v in golden["iterate"]:
        ts = capi.Column.from_list(v["time"], "int64")
        s0, W = capi.plan_windows(ts, v["interval"], v["offset"])
        assert s0 == v["windows"][0]["start"] and W == len(v["windows"]), v["name"]


def test_ctor_errors(golden):
    # same error text as the reference (rolling_test.go:70-98) modulo the wrapping prefix the shim adds
    with pytest.raises(capi.BowGpuError) as e:
        capi.plan_windows(capi.Column.from_list([0], "int64"), 0)
    assert e.value.code == -1 and e.value.message == "strictly positive interval required"
    with pytest.raises(capi.BowGpuError) as e:
        capi.plan_windows(capi.Column.from_list([0.0], "float64"), 1)
    assert e.value.message == "impossible to create a new intervalRolling on column of type float64"
    with pytest.raises(capi.BowGpuError) as e:
        capi.plan_windows(capi.Column.from_list([None, 3], "int64"), 1)
    assert e.value.code == -3


def test_no_cpu_fallback_without_gpu():
    try:
        n = capi.device_count()
    except capi.BowGpuError:
        n = 0
    if n > 0:
        pytest.skip("a GPU is present")
    ts = capi.Column.from_list([10, 15, 16], "int64")
    val = capi.Column.from_list([1.0, 2.0, 3.0], "float64")
    with pytest.raises(capi.BowGpuError) as e:
        capi.rolling_aggregate([ts, val], 0, 10, [("WindowStart", 0), ("Sum", 1)])
    assert e.value.code == -11


def test_route_bits_of_the_binding_equal_the_headers():
    """two public bits in include/bowgpu.h, the test / A-B bits in bow_amd/csrc/debug_routes.h (not part of the ABI): capi.py holds copies"""
    pub = open(os.path.join(ROOT, "include", "bowgpu.h")).read()
    prv = open(os.path.join(ROOT, "bow_amd", "csrc", "debug_routes.h")).read()
    bits = {}
    for src in (pub, prv):
        for name, val in re.findall(r"\b(BOWGPU_ROUTE_\w+) = (\d+)", src):
            bits[name] = int(val)
    assert set(re.findall(r"\b(BOWGPU_ROUTE_\w+) =", pub)) == {"BOWGPU_ROUTE_PINNED_STAGE", "BOWGPU_ROUTE_STRICT_ORDER"}
    for name, val in bits.items():
        if name == "BOWGPU_ROUTE__ALL":
            continue
        assert getattr(capi, name[len("BOWGPU_"):]) == val, name
    assert bits["BOWGPU_ROUTE__ALL"] == sum(v for k, v in bits.items() if k != "BOWGPU_ROUTE__ALL")
