"""shim/go/ holds the cgo binding a maintainer of the reference adds.  No Go toolchain exists in this image, so the
files cannot be compiled here; what CAN be checked is that they and the C ABI have not drifted apart: every C function, constant and
record the Go files name exists in include/bowgpu.h, the kind tags equal the header's enum values, and call sites pass the number of
arguments the prototypes take."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = open(os.path.join(ROOT, "include", "bowgpu.h")).read()
GO = {f: open(f).read() for f in glob.glob(os.path.join(ROOT, "shim", "go", "**", "*.go"), recursive=True)}


def _enums():
    vals = {}
    for body in re.findall(r"enum\s*\{(.*?)\};", HDR, flags=re.S):
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        nxt = 0
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            m = re.match(r"(\w+)\s*(?:=\s*(-?\d+))?$", item)
            assert m, item
            nxt = int(m.group(2)) if m.group(2) is not None else nxt
            vals[m.group(1)] = nxt
            nxt += 1
    for m in re.finditer(r"#define\s+(BOWGPU_\w+)\s+(-?\d+)", HDR):
        vals[m.group(1)] = int(m.group(2))
    return vals


def _prototypes():
    flat = re.sub(r"/\*.*?\*/", "", HDR, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|const char \*)\s*(bowgpu_\w+)\s*\(([^;{]*?)\)\s*;", flat, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return protos


def test_shim_files_exist_and_are_go_shaped():
    names = {os.path.relpath(f, os.path.join(ROOT, "shim", "go")) for f in GO}
    assert {"bowfill_gpu.go", "rolling/gpu_cgo.go", "rolling/gpu_kinds.go", "rolling/aggregation/gpu_kinds.go",
            "rolling/interpolation/gpu_kinds.go", "rolling/transformation/gpu_factor.go"} <= names
    for f, src in GO.items():
        assert re.search(r"^package \w+$", src, flags=re.M), f
        code = re.sub(r"//.*", "", src)
        assert code.count("{") == code.count("}") and code.count("(") == code.count(")"), f
        assert "\t" in src or "gpu_kinds.go" in f, f        # gofmt indents with tabs


def test_every_c_name_the_shim_uses_is_in_the_header():
    enums, protos = _enums(), _prototypes()
    structs = set(re.findall(r"typedef struct (bowgpu_\w+)", HDR))
    used_fn, used_const = set(), set()
    for f, src in GO.items():
        code = re.sub(r"//.*", "", src)
        for name in re.findall(r"\bC\.(bowgpu_\w+)\s*\(", code):
            used_fn.add(name)
        for name in re.findall(r"\bC\.(BOWGPU_\w+)", code):
            used_const.add(name)
        for name in re.findall(r"\bC\.sizeof_(bowgpu_\w+)", code):
            assert name in structs, (f, name)
        for name in re.findall(r"\bC\.(bowgpu_\w+)\b(?!\s*\()", code):
            assert name in structs or name in protos, (f, name)
    assert used_fn and used_fn <= set(protos), used_fn - set(protos)
    assert used_const <= set(enums), used_const - set(enums)
    # the entry points of the hot path are bound
    assert {"bowgpu_rolling_aggregate_planned", "bowgpu_plan_windows_ex", "bowgpu_rolling_interpolate_count", "bowgpu_rolling_interpolate_fill",
            "bowgpu_fill_linear", "bowgpu_fill", "bowgpu_is_col_sorted", "bowgpu_shard_begin", "bowgpu_shard_pass_begin", "bowgpu_shard_finish",
            "bowgpu_host_register", "bowgpu_last_error"} <= used_fn


def test_call_sites_pass_as_many_arguments_as_the_prototypes_take():
    protos = _prototypes()
    for f, src in GO.items():
        code = re.sub(r"//.*", "", src)
        for m in re.finditer(r"\bC\.(bowgpu_\w+)\s*\(", code):
            depth, i, args, cur = 1, m.end(), 0, ""
            while depth:
                ch = code[i]
                depth += ch in "([{"
                depth -= ch in ")]}"
                if ch == "," and depth == 1:
                    args += 1
                elif depth:
                    cur += ch
                i += 1
            n = 0 if not cur.strip() and args == 0 else args + 1
            assert n == protos[m.group(1)], (os.path.basename(f), m.group(1), n, protos[m.group(1)])


def test_kind_tags_equal_the_header_enums():
    enums = _enums()
    kinds = GO[os.path.join(ROOT, "shim", "go", "rolling", "gpu_kinds.go")]
    got = {k: int(v) for k, v in re.findall(r"\b(GPU\w+)\s+int32\s*=\s*(-?\d+)", kinds)}
    want = {"GPUKindWindowStart": "BOWGPU_AGG_WINDOW_START", "GPUKindSum": "BOWGPU_AGG_SUM", "GPUKindArithmeticMean": "BOWGPU_AGG_MEAN",
            "GPUKindMin": "BOWGPU_AGG_MIN", "GPUKindMax": "BOWGPU_AGG_MAX", "GPUKindCount": "BOWGPU_AGG_COUNT", "GPUKindFirst": "BOWGPU_AGG_FIRST",
            "GPUKindLast": "BOWGPU_AGG_LAST", "GPUKindIntegralStep": "BOWGPU_AGG_INTEGRAL_STEP",
            "GPUKindIntegralTrapezoid": "BOWGPU_AGG_INTEGRAL_TRAPEZOID", "GPUKindWeightedAvgStep": "BOWGPU_AGG_WAVG_STEP",
            "GPUKindWeightedAvgLinear": "BOWGPU_AGG_WAVG_LINEAR", "GPUKindMode": "BOWGPU_AGG_MODE",
            "GPUInterpWindowStart": "BOWGPU_INTERP_WINDOW_START", "GPUInterpLinear": "BOWGPU_INTERP_LINEAR",
            "GPUInterpStepPrevious": "BOWGPU_INTERP_STEP_PREVIOUS", "GPUInterpNone": "BOWGPU_INTERP_NONE"}
    for g, h in want.items():
        assert got[g] == enums[h], (g, got[g], h, enums[h])
    assert got["GPUKindNone"] == -1
    # bow.Type values the shim passes through as bowgpu_col.type (bowtypes.go:21-23)
    assert enums["BOWGPU_FLOAT64"] == 1 and enums["BOWGPU_INT64"] == 2
