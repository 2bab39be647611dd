"""The counter evidence under profiles/ must belong to the code this tree builds.

bench.py quotes `roofline.traffic` from the newest profiles/r*_pmc_hbm_traffic_bench_1e9.csv, and only when the rows carry the
identity (sha over the machine code: bow_amd/csrc/kernel_sha.py) of the kernel instantiation that ran.  In round 4 the last
product commit of the round changed the hash after the counters had been collected, nothing failed, and the driver's bench line
went out with "traffic": null.  This test fails instead: a change to the benched kernel (rolling_simple.hip, agg_device.h, anything
that alters its instructions) makes the CPU suite red until scratch/profile_bench.sh has been re-run on a GPU box and its
summaries committed (profiles/README.md)."""
import csv
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bow_amd", "csrc")


def _tree_sha():
    # (compiles rolling_simple.hip for gfx950 when its object is missing or older than its sources: ~30 s, nothing when current)
    subprocess.check_call(["make", "-s", "-C", CSRC, "../libbowgpu.kernel_sha.json"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    import bench
    sha = bench.kernel_sha(bench.BENCH_KERNEL_INSTANCE)
    assert sha, "bow_amd/libbowgpu.kernel_sha.json does not list %s" % bench.BENCH_KERNEL_INSTANCE
    return bench, sha


def test_the_committed_counter_traffic_belongs_to_the_kernel_this_tree_builds():
    bench, sha = _tree_sha()
    f = bench.newest_traffic_file()
    assert f, "no profiles/r*_pmc_hbm_traffic_bench_1e9.csv"
    rows = [r for r in csv.DictReader(open(f)) if bench.BENCH_KERNEL_INSTANCE in r["kernel"]]
    assert rows, "%s holds no row of %s: re-run scratch/profile_bench.sh" % (os.path.basename(f), bench.BENCH_KERNEL_INSTANCE)
    stale = sorted({r.get("kernel_sha") for r in rows} - {sha})
    assert not stale, ("%s was collected from another build of the benched kernel (kernel_sha %s, this tree builds %s): re-run "
                       "scratch/profile_bench.sh on a GPU box and commit its summaries" % (os.path.basename(f), stale, sha))
    assert {r["counter"] for r in rows} >= {"FETCH_SIZE", "WRITE_SIZE"}
    t = bench.measured_traffic(bench.HEADLINE_ROWS, bench.BENCH_KERNEL_INSTANCE)
    # 16.0 GB read + 1.6 GB written algorithmic; wasted re-reads would show as traffic well above that
    assert t is not None and 17.0e9 <= t <= 19.5e9, t
    # the kernel-stats summary and the counter table of the same collection sit next to it
    tag = os.path.basename(f).split("_")[0]
    for name in ("%s_kernel_stats_bench_1e9.csv", "%s_pmc_counters_bench_1e9.txt", "%s_bench_1e9.json"):
        assert os.path.exists(os.path.join(ROOT, "profiles", name % tag)), name % tag
    assert "kernel_sha: %s" % sha in open(os.path.join(ROOT, "profiles", "%s_pmc_counters_bench_1e9.txt" % tag)).read()
    line = json.load(open(os.path.join(ROOT, "profiles", "%s_bench_1e9.json" % tag)))
    assert line["roofline"]["kernel_instance"] == bench.BENCH_KERNEL_INSTANCE and line["roofline"]["kernel_sha"] == sha
    assert line["roofline"]["traffic"] is not None


def test_the_identity_ignores_what_the_kernel_is_not_compiled_from(tmp_path):
    """the hash covers the kernel's machine code + descriptor: every instantiation is listed, two different instantiations differ, and the
    descriptor's entry offset (which moves when ANOTHER kernel of the object changes size) is left out"""
    bench, sha = _tree_sha()
    doc = json.load(open(os.path.join(ROOT, "bow_amd", "libbowgpu.kernel_sha.json")))
    ks = doc["kernels"]
    assert len(ks) >= 64 and all(k.startswith("rolling_simple_kernel<") for k in ks)
    padded = bench.BENCH_KERNEL_INSTANCE.replace("false>", "true>")
    assert padded in ks and ks[padded]["sha"] != sha
    import sys
    sys.path.insert(0, CSRC)
    import kernel_sha
    again = kernel_sha.kernel_identities(os.path.join(CSRC, "build", "rolling_simple.o"), "rolling_simple_kernel")
    assert again[bench.BENCH_KERNEL_INSTANCE]["sha"] == sha


ROUND = "r06"


def test_every_profile_file_of_the_round_is_described_and_every_described_file_exists():
    """profiles/README.md, the section of the current round: the files it names are there, and no file of the round sits in profiles/ without
    a line about it.  The same for what DESIGN.md, README.md, the sources under bow_amd/csrc and include/bowgpu.h quote from ANY round
    (a routing rule in api.cpp cited an A/B file that had never been committed, and nothing noticed)."""
    import re
    readme = open(os.path.join(ROOT, "profiles", "README.md")).read()
    sec = readme[readme.index("## Round 6"):readme.index("## Round 5")]
    named = set(re.findall(r"`(%s_[A-Za-z0-9_.]+)`" % ROUND, sec))
    present_all = set(os.listdir(os.path.join(ROOT, "profiles")))
    present = {f for f in present_all if f.startswith(ROUND + "_")}
    assert named <= present, sorted(named - present)
    assert present <= named, sorted(present - named)
    pat = re.compile(r"(?:profiles/)?(r0[1-9]_[A-Za-z0-9_]+\.(?:txt|csv|json))")
    docs = [os.path.join(ROOT, "DESIGN.md"), os.path.join(ROOT, "README.md"), os.path.join(ROOT, "HISTORY.md"), os.path.join(ROOT, "include", "bowgpu.h")]
    docs += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cpp", ".hip", ".h"))]
    for path in docs:
        if not os.path.exists(path):
            continue
        quoted = set(pat.findall(open(path).read()))
        missing = sorted(q for q in quoted if q not in present_all)
        assert not missing, "%s quotes evidence that is not under profiles/: %s" % (os.path.relpath(path, ROOT), missing)


# which device sources a counter file of the round speaks about (host code - api.cpp, extras.cpp, multi.cpp - routes calls; it does not change
# what a kernel's counters say)
COUNTER_FILES = {
    "r06_pmc_rolling_fused.txt": ["rolling_fused.hip", "rolling_simple.hip", "agg_device.h", "interp_device.h", "bitmap_device.h"],
    "r06_pmc_long_strict.txt": ["long_windows.hip", "agg_device.h"],
    "r06_pmc_band.txt": ["long_windows.hip", "rolling_simple.hip", "agg_device.h"],
    "r06_pmc_callers.txt": ["interp_fill.hip", "agg_device.h", "bitmap_device.h"],
    "r06_pmc_mid_windows.txt": ["rolling_tw.hip", "rolling_twc.hip", "rolling_simple.hip", "agg_device.h"],
    "r06_pmc_interp_wave3_1e8.txt": ["interpolate.hip", "interp_device.h", "agg_device.h"],
    "r06_pmc_long_short_tw.txt": ["long_windows.hip", "agg_device.h"],
}


def test_the_rounds_counter_files_belong_to_the_kernels_this_tree_builds():
    """the sha rule of the benched instantiation extended to the round's other counter files.  They were collected from
    ONE snapshot (profiles/r06_commit.txt, first line; a file collected again later has its own line there); a device source one of them
    speaks about changed since ITS commit means it describes other code - the CPU suite is red until it has been collected again on a GPU
    box and committed.  (Without a git history - the
    GPU box gets a snapshot - there is nothing to compare with.)"""
    for f in COUNTER_FILES:
        assert os.path.exists(os.path.join(ROOT, "profiles", f)), f
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        return
    # profiles/r06_commit.txt: the snapshot the round's collection ran on; further lines "<file> <commit>" name the files collected again later
    lines = [l.split() for l in open(os.path.join(ROOT, "profiles", "%s_commit.txt" % ROUND)).read().splitlines() if l.strip() and not l.startswith("#")]
    default, later = lines[0][0], {l[0]: l[1] for l in lines[1:]}
    assert set(later) <= set(COUNTER_FILES), sorted(set(later) - set(COUNTER_FILES))
    dirty = subprocess.check_output(["git", "diff", "--name-only", "HEAD", "--", "bow_amd/csrc"], cwd=ROOT).decode().split()   # (working-tree edits count too)
    changed_since = {}
    for commit in {default, *later.values()}:
        rc = subprocess.run(["git", "cat-file", "-e", commit + "^{commit}"], cwd=ROOT)
        assert rc.returncode == 0, "profiles/%s_commit.txt names %s, which this history does not hold" % (ROUND, commit)
        names = subprocess.check_output(["git", "diff", "--name-only", commit, "HEAD", "--", "bow_amd/csrc"], cwd=ROOT).decode().split()
        changed_since[commit] = {os.path.basename(f) for f in names + dirty}
    stale = {}
    for f, srcs in COUNTER_FILES.items():
        hit = changed_since[later.get(f, default)] & set(srcs)
        if hit:
            stale[f] = (later.get(f, default)[:10], sorted(hit))
    assert not stale, ("device sources changed since these counter files were collected (file: (commit, sources)): %s - collect them again on a GPU box "
                       "(scratch/profile_all.sh %s, scratch/refresh_fused.sh) and name the commit in profiles/%s_commit.txt" % (stale, ROUND, ROUND))
