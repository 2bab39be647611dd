#!/usr/bin/env python3
"""Generates shim/go/patches/*.patch - the edits a maintainer applies to the reference's OWN files - and
tests/golden/go_type_shapes.json - the struct fields / method sets of the reference types the shim touches.

Run in the build container, where the reference tree is at /root/reference (it never travels: the GPU box sees only the
committed patches and the JSON).  Each patch is a unified diff with two lines of context; nothing else of the reference is stored.
tests/test_go_shim.py re-runs this script when the reference is present and requires the committed files to be reproduced
byte for byte, then applies the patches to a scratch copy with `patch --dry-run`.

    python3 shim/go/make_patches.py [--check]
"""
import difflib
import json
import os
import re
import sys

REF = os.environ.get("BOW_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

AGG_TAGS = {"WindowStart": "GPUKindWindowStart", "Sum": "GPUKindSum", "ArithmeticMean": "GPUKindArithmeticMean", "Min": "GPUKindMin",
            "Max": "GPUKindMax", "Count": "GPUKindCount", "First": "GPUKindFirst", "Last": "GPUKindLast",
            "IntegralStep": "GPUKindIntegralStep", "IntegralTrapezoid": "GPUKindIntegralTrapezoid",
            "WeightedAverageStep": "GPUKindWeightedAvgStep", "WeightedAverageLinear": "GPUKindWeightedAvgLinear", "Mode": "GPUKindMode"}
INTERP_TAGS = {"WindowStart": "GPUInterpWindowStart", "Linear": "GPUInterpLinear", "StepPrevious": "GPUInterpStepPrevious",
               "None": "GPUInterpNone"}


def sub_once(src, old, new, path):
    assert src.count(old) == 1, (path, old, src.count(old))
    return src.replace(old, new)


def edit_aggregation_go(src, path):
    src = sub_once(src, "\toutputName string\n\ttyp        bow.Type\n}",
                   "\toutputName string\n\ttyp        bow.Type\n\n"
                   "\tgpuKind int32 // bowgpu: 0 = no tag (a user closure), else BOWGPU_AGG_* + 1 (gpu_kinds.go); a field, so copies keep it\n}", path)
    src = sub_once(src, "func (r *intervalRolling) aggregateWindows(aggrs []ColAggregation) (bow.Bow, error) {\n",
                   "func (r *intervalRolling) aggregateWindows(aggrs []ColAggregation) (bow.Bow, error) {\n"
                   "\tif b, err := r.aggregateWindowsGPU(aggrs); err != errDeclined { // bowgpu: gpu_cgo.go / gpu_off.go\n"
                   "\t\treturn b, err\n\t}\n\n", path)
    return src


def edit_interpolation_go(src, path):
    src = sub_once(src, "\tcolIndex int\n}", "\tcolIndex int\n\n"
                   "\tgpuKind int32 // bowgpu: 0 = no tag, else BOWGPU_INTERP_* + 1 (gpu_kinds.go)\n}", path)
    src = sub_once(src, "func (r *intervalRolling) interpolateWindows(interps []ColInterpolation) (bow.Bow, error) {\n",
                   "func (r *intervalRolling) interpolateWindows(interps []ColInterpolation) (bow.Bow, error) {\n"
                   "\tif b, err := r.interpolateWindowsGPU(interps); err != errDeclined { // bowgpu: gpu_cgo.go / gpu_off.go\n"
                   "\t\treturn b, err\n\t}\n\n", path)
    return src


def edit_interpolation_go_lazy(src, path):
    # Interpolate: once the interpolators are validated and the interval column is known to be kept, in front of interpolateWindows
    src = sub_once(src, "\tb, err := rCopy.interpolateWindows(interps)\n",
                   "\tif lazy := rCopy.lazyInterpolationGPU(interps, newIntervalCol); lazy != nil { // bowgpu: gpu_lazy.go, gpu_cgo.go / gpu_off.go\n"
                   "\t\treturn lazy\n\t}\n\n"
                   "\tb, err := rCopy.interpolateWindows(interps)\n", path)
    return src


def edit_bowfill_go(src, path):
    # FillLinear: behind ALL of its argument checks (bowfill.go:15-55), in front of the loop
    src = sub_once(src, "\tif b.Column(toFillColIndex).NullN() == 0 {\n\t\treturn b, nil\n\t}\n\tbuf := b.NewBufferFromCol(toFillColIndex)\n",
                   "\tif b.Column(toFillColIndex).NullN() == 0 {\n\t\treturn b, nil\n\t}\n"
                   "\tif filled, err := b.fillLinearGPU(refColIndex, toFillColIndex); err != errGPUDeclined { // bowgpu: bowfill_gpu.go / bowfill_gpu_off.go\n"
                   "\t\treturn filled, err\n\t}\n"
                   "\tbuf := b.NewBufferFromCol(toFillColIndex)\n", path)
    # FillMean: first thing in the per-column goroutine
    src = sub_once(src, "\t\t\tdefer wg.Done()\n\n\t\t\tbuf := b.NewBufferFromCol(colIndex)\n\t\t\tfor rowIndex := 0; rowIndex < b.NumRows(); rowIndex++ {\n",
                   "\t\t\tdefer wg.Done()\n\n"
                   "\t\t\tif s, ok := b.fillGPU(colIndex, \"Mean\"); ok { // bowgpu\n"
                   "\t\t\t\tfilledSeries[colIndex] = s\n\t\t\t\treturn\n\t\t\t}\n"
                   "\t\t\tbuf := b.NewBufferFromCol(colIndex)\n\t\t\tfor rowIndex := 0; rowIndex < b.NumRows(); rowIndex++ {\n", path)
    # fill (FillPrevious / FillNext): the same
    src = sub_once(src, "\t\t\tdefer wg.Done()\n\n\t\t\tdata := b.Column(colIndex).Data()\n",
                   "\t\t\tdefer wg.Done()\n\n"
                   "\t\t\tif s, ok := b.fillGPU(colIndex, method); ok { // bowgpu\n"
                   "\t\t\t\tfilledSeries[colIndex] = s\n\t\t\t\treturn\n\t\t\t}\n"
                   "\t\t\tdata := b.Column(colIndex).Data()\n", path)
    return src


def edit_bowassertion_go(src, path):
    src = sub_once(src, "\tif b.IsColEmpty(colIndex) {\n\t\treturn false\n\t}\n",
                   "\tif b.IsColEmpty(colIndex) {\n\t\treturn false\n\t}\n"
                   "\tif sorted, err := b.isColSortedGPU(colIndex); err != errGPUDeclined { // bowgpu: bowfill_gpu.go / bowfill_gpu_off.go\n"
                   "\t\treturn sorted\n\t}\n", path)
    return src


def edit_whole_go(src, path):
    src = sub_once(src, "\tintervalColIndex, err := b.ColumnIndex(intervalColName)\n\tif err != nil {\n\t\treturn nil, err\n\t}\n",
                   "\tintervalColIndex, err := b.ColumnIndex(intervalColName)\n\tif err != nil {\n\t\treturn nil, err\n\t}\n"
                   "\tif res, err := aggregateWholeGPU(b, intervalColIndex, aggrs); err != rolling.ErrGPUDeclined { // bowgpu: whole_gpu.go\n"
                   "\t\treturn res, err\n\t}\n", path)
    return src


def error_strings():
    """the texts of the reference's errors that the shim words itself (the rest come through bowgpu_last_error, which api.cpp words)"""
    out = {}
    src = open(os.path.join(REF, "bowfill.go")).read()
    m = re.search(r'fmt\.Errorf\("(refColIndex \'%d\' is empty or not sorted)",', src)
    out["bowfill.go FillLinear not sorted"] = m.group(1)
    src = open(os.path.join(REF, "rolling", "aggregation.go")).read()
    m = re.search(r'"(must keep interval column \'%s\')"', src)
    out["rolling/aggregation.go keep interval"] = m.group(1)
    return out


def edit_constructors(src, path, ctor, tags, ret):
    """every `func Name(col string) rolling.<ret> { ... rolling.<ctor>(... ) }`: the call becomes <ctor>GPU(..., rolling.<tag>)"""
    out, pos, done = [], 0, []
    for m in re.finditer(r"^func (\w+)\(\w+ string\) rolling\.%s \{\n" % ret, src, flags=re.M):
        name = m.group(1)
        if name not in tags:
            continue
        end = src.index("\n}\n", m.end()) + 3          # the function's closing brace (gofmt: column 0)
        body = src[m.start():end]
        body = sub_once(body, "rolling.%s(" % ctor, "rolling.%sGPU(" % ctor, path)
        lines = body.rstrip("\n").split("\n")
        assert lines[-1] == "}" and lines[-2].rstrip().endswith(")"), (path, name, lines[-2:])
        call_end = lines[-2]
        if call_end.strip() == ")":                       # `\t\t},\n\t)` form (linear.go)
            assert lines[-3].rstrip().endswith(","), (path, name)
            lines.insert(-2, lines[-3][:len(lines[-3]) - len(lines[-3].lstrip())] + "rolling.%s," % tags[name])
        else:                                             # `\t\t})` form
            lines[-2] = call_end.rstrip()[:-1] + ", rolling.%s)" % tags[name]
        out.append(src[pos:m.start()])
        out.append("\n".join(lines) + "\n")
        pos = end
        done.append(name)
    out.append(src[pos:])
    assert done, path
    return "".join(out), done


def diff(rel, old, new):
    d = difflib.unified_diff(old.splitlines(True), new.splitlines(True), "a/" + rel, "b/" + rel, n=2)
    return "".join(d)


def go_shapes():
    """struct fields and method names of the reference types the shim reads or extends"""
    shapes = {}

    def struct_fields(path, name):
        src = open(os.path.join(REF, path)).read()
        m = re.search(r"^type %s struct \{\n(.*?)^\}" % name, src, flags=re.M | re.S)
        fields = []
        for line in m.group(1).split("\n"):
            line = re.sub(r"//.*", "", line).strip()
            if line:
                fields.append(line.split()[0])
        return fields

    def iface_methods(path, name):
        src = open(os.path.join(REF, path)).read()
        m = re.search(r"^type %s interface \{\n(.*?)^\}" % name, src, flags=re.M | re.S)
        return sorted(set(re.findall(r"^\t(\w+)\(", m.group(1), flags=re.M)))

    def methods_of(paths, recv):
        names = set()
        for p in paths:
            names |= set(re.findall(r"^func \(\w+ \*?%s\) (\w+)\(" % recv, open(os.path.join(REF, p)).read(), flags=re.M))
        return sorted(names)

    def funcs(path):
        return sorted(set(re.findall(r"^func (\w+)\(", open(os.path.join(REF, path)).read(), flags=re.M)))

    shapes["rolling.colAggregation"] = {"kind": "struct", "file": "rolling/aggregation.go", "fields": struct_fields("rolling/aggregation.go", "colAggregation"),
                                        "methods": methods_of(["rolling/aggregation.go"], "colAggregation")}
    shapes["rolling.ColAggregation"] = {"kind": "interface", "file": "rolling/aggregation.go", "methods": iface_methods("rolling/aggregation.go", "ColAggregation")}
    shapes["rolling.ColInterpolation"] = {"kind": "struct", "file": "rolling/interpolation.go", "fields": struct_fields("rolling/interpolation.go", "ColInterpolation"),
                                          "methods": methods_of(["rolling/interpolation.go"], "ColInterpolation")}
    shapes["rolling.intervalRolling"] = {"kind": "struct", "file": "rolling/rolling.go", "fields": struct_fields("rolling/rolling.go", "intervalRolling"),
                                         "methods": methods_of(["rolling/rolling.go", "rolling/aggregation.go", "rolling/interpolation.go"], "intervalRolling")}
    shapes["rolling.Rolling"] = {"kind": "interface", "file": "rolling/rolling.go", "methods": iface_methods("rolling/rolling.go", "Rolling")}
    shapes["rolling.Options"] = {"kind": "struct", "file": "rolling/rolling.go", "fields": struct_fields("rolling/rolling.go", "Options")}
    shapes["bow.Bow"] = {"kind": "interface", "file": "bow.go", "methods": iface_methods("bow.go", "Bow")}
    shapes["transformation"] = {"kind": "package", "file": "rolling/transformation/factor.go", "funcs": funcs("rolling/transformation/factor.go"),
                                "Func": re.search(r"^type Func (.*)$", open(os.path.join(REF, "rolling/transformation/factor.go")).read(), flags=re.M).group(1)}
    shapes["rolling.funcs"] = {"kind": "package", "funcs": sorted(set(funcs("rolling/aggregation.go") + funcs("rolling/interpolation.go") + funcs("rolling/rolling.go")))}
    shapes["bow.funcs"] = {"kind": "package", "funcs": sorted(set(funcs("bow.go") + funcs("bowseries.go") + funcs("bowbuffer.go") + funcs("bowmetadata.go")))}
    shapes["bow.bow"] = {"kind": "struct", "file": "bow.go", "fields": struct_fields("bow.go", "bow"),
                         "methods": methods_of([f for f in sorted(os.listdir(REF)) if f.endswith(".go") and not f.endswith("_test.go")], "bow")}
    shapes["error_strings"] = error_strings()
    shapes["go.mod"] = {"go": re.search(r"^go (\S+)$", open(os.path.join(REF, "go.mod")).read(), flags=re.M).group(1)}
    return shapes


def generate():
    files = {}
    p1 = ""
    for rel, fn in (("rolling/aggregation.go", edit_aggregation_go), ("rolling/interpolation.go", edit_interpolation_go)):
        old = open(os.path.join(REF, rel)).read()
        p1 += diff(rel, old, fn(old, rel))
    files["patches/0001-rolling-gpu-kind-fields-and-hooks.patch"] = p1
    p2, tagged = "", []
    for rel in sorted(os.listdir(os.path.join(REF, "rolling", "aggregation"))):
        if rel.endswith("_test.go") or not rel.endswith(".go") or rel == "whole.go":
            continue
        rel = "rolling/aggregation/" + rel
        old = open(os.path.join(REF, rel)).read()
        new, done = edit_constructors(old, rel, "NewColAggregation", AGG_TAGS, "ColAggregation")
        tagged += done
        p2 += diff(rel, old, new)
    assert sorted(tagged) == sorted(AGG_TAGS), sorted(set(AGG_TAGS) - set(tagged))
    files["patches/0002-aggregation-constructors-carry-their-kind.patch"] = p2
    p3, tagged = "", []
    for rel in sorted(os.listdir(os.path.join(REF, "rolling", "interpolation"))):
        if rel.endswith("_test.go") or not rel.endswith(".go"):
            continue
        rel = "rolling/interpolation/" + rel
        old = open(os.path.join(REF, rel)).read()
        new, done = edit_constructors(old, rel, "NewColInterpolation", INTERP_TAGS, "ColInterpolation")
        tagged += done
        p3 += diff(rel, old, new)
    assert sorted(tagged) == sorted(INTERP_TAGS)
    files["patches/0003-interpolation-constructors-carry-their-kind.patch"] = p3
    p4 = ""
    for rel, fn in (("bowassertion.go", edit_bowassertion_go), ("bowfill.go", edit_bowfill_go)):
        old = open(os.path.join(REF, rel)).read()
        p4 += diff(rel, old, fn(old, rel))
    files["patches/0004-fill-and-is-col-sorted-hooks.patch"] = p4
    old = open(os.path.join(REF, "rolling/aggregation/whole.go")).read()
    files["patches/0005-whole-frame-aggregate-hook.patch"] = diff("rolling/aggregation/whole.go", old, edit_whole_go(old, "rolling/aggregation/whole.go"))
    # 0006 comes on top of 0001 (both edit rolling/interpolation.go): the diff is taken against the file as 0001 leaves it
    base = edit_interpolation_go(open(os.path.join(REF, "rolling/interpolation.go")).read(), "rolling/interpolation.go")
    files["patches/0006-interpolate-returns-a-lazy-rolling.patch"] = diff("rolling/interpolation.go", base, edit_interpolation_go_lazy(base, "rolling/interpolation.go"))
    return files, json.dumps(go_shapes(), indent=1, sort_keys=True) + "\n"


def main():
    files, shapes = generate()
    targets = {os.path.join(HERE, k): v for k, v in files.items()}
    targets[os.path.join(ROOT, "tests", "golden", "go_type_shapes.json")] = shapes
    if "--check" in sys.argv:
        bad = [p for p, v in targets.items() if not os.path.exists(p) or open(p).read() != v]
        if bad:
            sys.exit("stale: %s (run shim/go/make_patches.py)" % ", ".join(os.path.relpath(b, ROOT) for b in bad))
        return
    for p, v in targets.items():
        os.makedirs(os.path.dirname(p), exist_ok=True)
        open(p, "w").write(v)
        print("wrote", os.path.relpath(p, ROOT), len(v), "bytes")


if __name__ == "__main__":
    main()
