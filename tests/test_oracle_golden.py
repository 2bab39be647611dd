"""Pins the CPU oracle against every golden vector the reference's own tests hold for the
rolling path (tests/golden/reference_vectors.json).  Exact equality,
as the reference's tests use (Bow.Equal => reflect.DeepEqual, bow.go:227-275)."""
import math

import numpy as np
import pytest

from oracle import pyoracle as orc

T = {"float64": orc.FLOAT64, "int64": orc.INT64, "bool": orc.BOOLEAN}


def same(a, b):
    """exact list equality with None; floats compared bitwise-equal (no NaN in the vectors)."""
    assert len(a) == len(b), (a, b)
    for x, y in zip(a, b):
        if x is None or y is None:
            assert x is None and y is None, (a, b)
        else:
            assert type(x) == type(y) or isinstance(x, (int, float)) and isinstance(y, (int, float)), (a, b)
            assert x == y, (a, b)


def two_col(time, value, vtype="float64"):
    return [orc.Column.from_list(time, orc.INT64), orc.Column.from_list(value, T[vtype])]


def test_offsets(golden):
    for o in golden["offsets"]:
        assert orc.enforce_interval_and_offset(o["interval"], o["offset"]) == o["norm"]


def test_num_windows(golden):
    for v in golden["num_windows"]:
        ts = orc.Column.from_list(v["time"], orc.INT64)
        s0, W = orc.plan_windows(ts, v["interval"], v["offset"])
        assert W == v["W"], v["name"]


def test_ctor_errors(golden):
    # interval <= 0 -> error; float64 interval column -> error (rolling_test.go:70-98)
    with pytest.raises(orc.OracleError) as e:
        orc.plan_windows(orc.Column.from_list([0], orc.INT64), 0, 0)
    assert e.value.code == -1
    with pytest.raises(orc.OracleError) as e:
        orc.plan_windows(orc.Column.from_list([0.0], orc.FLOAT64), 1, 0)
    assert e.value.code == -2


def test_iterate(golden):
    for v in golden["iterate"]:
        ts = orc.Column.from_list(v["time"], orc.INT64)
        wins = orc.iterate_windows(ts, v["interval"], v["offset"], v["inclusive"])
        assert len(wins) == len(v["windows"]), v["name"]
        for got, exp in zip(wins, v["windows"]):
            assert got["first_value"] == exp["start"], v["name"]
            assert got["last_value"] == exp["end"], v["name"]
            assert got["first_index"] == exp["first_index"], v["name"]
            rows = list(range(got["slice_begin"], got["slice_end"]))
            assert [v["time"][r] for r in rows] == exp["time_rows"], (v["name"], exp)
            assert [v["value"][r] for r in rows] == exp["value_rows"], (v["name"], exp)


def test_reducers(golden):
    for v in golden["reducers"]:
        b = golden["bows"][v["bow"]]
        cols = two_col(b["time"], b["value"], b["value_type"])
        outs, nic = orc.aggregate(cols, 0, v["interval"], [("WindowStart", 0), (v["reducer"], 1, v["factors"])],
                                  offset=v["offset"])
        assert nic == 0
        assert outs[0].type == orc.INT64
        same(outs[0].to_list(), v["expect_time"])
        assert outs[1].type == T[v["expect_type"]], (v["reducer"], v["name"])
        same(outs[1].to_list(), v["expect_value"])


KIND = {"FirstValue": ("WindowStart", None), "NumRows": ("NumRows", None), "NumRows2x": ("NumRows", [2.0])}


def test_driver(golden):
    for v in golden["driver"]:
        cols = two_col(v["time"], v["value"])
        names = {"time": 0, "value": 1}
        if "error" in v:
            aggs = []
            for name, kind, _ in v["aggs"]:
                aggs.append((KIND.get(kind, ("WindowStart", None))[0], names.get(name, 7)))
            with pytest.raises(orc.OracleError) as e:
                orc.aggregate(cols, 0, v["interval"], aggs)
            assert e.value.code == (-5 if "must keep" in v["error"] else -6)
            continue
        aggs = [(KIND[k][0], names[n], KIND[k][1]) for n, k, _ in v["aggs"]]
        outs, nic = orc.aggregate(cols, 0, v["interval"], aggs)
        for o, typ, exp in zip(outs, v["expect_types"], v["expect"]):
            assert o.type == T[typ]
            same(o.to_list(), exp)
        # new interval column = last aggregator reading the interval column (aggregation.go:158-160)
        assert nic == max(i for i, a in enumerate(v["aggs"]) if a[0] == "time")


def test_whole(golden):
    for v in golden["whole"]:
        cols = two_col(v["time"], v["value"])
        aggs = []
        for a in v["aggs"]:
            k, cname = a.split(":")
            aggs.append((k, 0 if cname == "time" else 1))
        outs = orc.aggregate_whole(cols, 0, aggs)
        for o, exp in zip(outs, v["expect"]):
            same(o.to_list(), exp)


def _interps(spec):
    out = []
    for i, s in enumerate(spec):
        if s.startswith("Const:"):
            out.append({"kind": "Const", "col": i, "const": float(s.split(":")[1])})
        else:
            out.append({"kind": s, "col": i})
    return out


def test_interpolate(golden):
    for v in golden["interpolate"]:
        cols = two_col(v["time"], v["value"])
        outs = orc.interpolate(cols, 0, v["interval"], _interps(v["interps"]), offset=v["offset"])
        same(outs[0].to_list(), v["expect_time"])
        same(outs[1].to_list(), v["expect_value"])


def test_interpolate_errors(golden):
    cols = two_col([10, 15], [True, False], "bool")
    with pytest.raises(orc.OracleError) as e:
        orc.interpolate(cols, 0, 2, _interps(["WindowStart", "Linear"]))
    assert e.value.code == -7  # type whitelist: linear_test.go:146-162
    cols = two_col([10, 13], [1.0, 1.3])
    with pytest.raises(orc.OracleError) as e:
        orc.interpolate(cols, 0, 2, [{"kind": "Const", "col": 1, "const": 9.9}])
    assert e.value.code == -5  # must keep interval column: interpolation_test.go:37-47


def test_fill_linear(golden):
    names = ["a", "b", "c", "d", "e"]
    for v in golden["fill_linear"]:
        cols = [orc.Column.from_list([None if x is None else (float(x) if v["type"] == "float64" else x)
                                      for x in golden["fill_bow"][n]], T[v["type"]]) for n in names]
        ref, fill = names.index(v["ref"]), names.index(v["fill"])
        if v.get("error"):
            with pytest.raises(orc.OracleError) as e:
                orc.fill_linear(cols, ref, fill)
            assert e.value.code == -8
            continue
        out, unchanged = orc.fill_linear(cols, ref, fill)
        assert not unchanged
        same(out.to_list(), v["expect"])
    m = golden["fill_linear_meta"]
    cols = [orc.Column.from_list(m["ref"], T[m["ref_type"]]), orc.Column.from_list(m["fill"], T[m["fill_type"]])]
    out, _ = orc.fill_linear(cols, 0, 1)
    same(out.to_list(), m["expect"])


def test_fill_previous_next_mean(golden):
    # bowfill_test.go:29-154, :204-330 - every column of newFreshBow under each method
    n = 0
    for typ, methods in golden["fill_methods"].items():
        for method, expect in methods.items():
            for name in ["a", "b", "c", "d", "e"]:
                data = [None if x is None else (float(x) if typ == "float64" else x) for x in golden["fill_bow"][name]]
                out, unchanged = orc.fill(orc.Column.from_list(data, T[typ]), method)
                assert not unchanged and out.type == T[typ]
                same(out.to_list(), expect[name])
                n += 1
    assert n == 30
    out, unchanged = orc.fill(orc.Column.from_list([1, 2, 3], orc.INT64), "Mean")
    assert unchanged and out.to_list() == [1, 2, 3]


def test_factor(golden):
    # Factor(0.1): int64 11 -> 1, float64 11. -> 1.1 (factor_test.go:24-34), via a 1-window Last aggregation
    cols = [orc.Column.from_list([0], orc.INT64), orc.Column.from_list([11], orc.INT64)]
    outs, _ = orc.aggregate(cols, 0, 10, [("WindowStart", 0), ("Last", 1, [0.1])])
    assert outs[1].type == orc.INT64 and outs[1].to_list() == [1]
    cols = [orc.Column.from_list([0], orc.INT64), orc.Column.from_list([11.0], orc.FLOAT64)]
    outs, _ = orc.aggregate(cols, 0, 10, [("WindowStart", 0), ("Last", 1, [0.1])])
    assert outs[1].to_list() == [1.1]


def test_fixture_file_is_exactly_what_the_committed_script_writes():
    """tests/golden/reference_vectors.json must be reproducible: it equals, value for value, what
    tests/golden/transcribe_reference_vectors.py (the hand transcription, with its file:line citations) writes."""
    import importlib.util
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("transcribe_reference_vectors", os.path.join(here, "transcribe_reference_vectors.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(os.path.join(here, "reference_vectors.json")) as f:
        committed = json.load(f)
    assert json.loads(json.dumps(mod.out)) == committed
