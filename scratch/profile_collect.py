"""Boil the rocprofv3 output of scratch/profile_bench.sh down to the files kept under profiles/."""
import csv, glob, hashlib, os, sys

out, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sha, BENCH_KERNEL_INSTANCE   # identity of the machine code of each rolling_simple_kernel instantiation (bow_amd/libbowgpu.kernel_sha.json)


def instance_of(kernel_name):
    """void bowgpu::rolling_simple_kernel<0, false, ...>(bowgpu::SimpleParams, long, long) -> rolling_simple_kernel<0, false, ...>"""
    s = kernel_name.replace("void ", "").replace("bowgpu::", "")
    return s[:s.index(">") + 1] if ">" in s else s.split("(")[0]


sha = kernel_sha(BENCH_KERNEL_INSTANCE)
BENCH_GRID = 100_000_000   # work-items of the benched launch (1e9 rows = 1 953 125 tiles x 64 lanes); smaller launches of the same kernel are other legs

# 1. kernel stats
rows = []
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
with open("%s/%s_kernel_stats_bench_1e9.csv" % (out, tag), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])

# 2. HBM traffic counters of the rolling kernels: full kernel signature + source hash per row
with open("%s/%s_pmc_hbm_traffic_bench_1e9.csv" % (out, tag), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "counter", "value_KB", "dispatch_id", "kernel_sha"])
    for sub in ("fetch", "write"):
        for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "rolling" not in r["Kernel_Name"] or int(r["Grid_Size"]) < BENCH_GRID:
                    continue     # (the bench line's host_pinned leg runs the same kernel over 1e8 rows: not the benched launch)
                w.writerow([r["Kernel_Name"], r["Counter_Name"], r["Counter_Value"], r["Dispatch_Id"], kernel_sha(instance_of(r["Kernel_Name"]))])

# 3. the SQ / L2 / L1 counter table of the benched kernel (averages over its dispatches)
import collections
acc = collections.defaultdict(list)
name = None
for sub in ("sq1", "sq2", "tcc", "tcp", "fetch", "write"):
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rolling_simple_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) >= BENCH_GRID:
                name = r["Kernel_Name"]
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for f in glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rolling_simple_kernel" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= BENCH_GRID:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open("%s/%s_pmc_counters_bench_1e9.txt" % (out, tag), "w") as fh:
    fh.write("kernel: %s\nkernel_sha: %s\n" % (name, kernel_sha(instance_of(name)) if name else sha))
    if dur:
        dur.sort()
        fh.write("kernel-trace duration: n=%d min %.4f ms median %.4f ms max %.4f ms\n" % (len(dur), dur[0], dur[len(dur) // 2], dur[-1]))
    av = {k: sum(v) / len(v) for k, v in acc.items()}
    for k in sorted(av):
        fh.write("%-32s n=%d avg=%.6g\n" % (k, len(acc[k]), av[k]))
    waves = av.get("SQ_WAVES")
    if waves:
        fh.write("\nper wavefront (one 512-row tile): ")
        fh.write(", ".join("%s %.0f" % (k.replace("SQ_INSTS_", "").replace("SQ_", ""), av[k] / waves) for k in
                           ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if k in av) + "\n")
        for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in av:
                fh.write("  %-20s %.0f cycles per wavefront (quad-cycle counter x 4)\n" % (k, 4 * av[k] / waves))
        if "SQ_WAVE_CYCLES" in av and "SQ_BUSY_CYCLES" in av:
            fh.write("  average resident wavefronts per CU: %.1f (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / 2)\n" % (av["SQ_WAVE_CYCLES"] / av["SQ_BUSY_CYCLES"] / 2))
    if "FETCH_SIZE" in av and "WRITE_SIZE" in av:
        fh.write("HBM traffic per launch: read 2 x FETCH_SIZE = %.3f GB, written WRITE_SIZE = %.3f GB\n" % (2 * av["FETCH_SIZE"] * 1024 / 1e9, av["WRITE_SIZE"] * 1024 / 1e9))
    if "TCP_TCC_READ_REQ_LATENCY_sum" in av:
        fh.write("L1 -> L2 read latency %.0f cycles, write (to acknowledgement) %.0f cycles\n" %
                 (av["TCP_TCC_READ_REQ_LATENCY_sum"] / av["TCP_TCC_READ_REQ_sum"], av["TCP_TCC_WRITE_REQ_LATENCY_sum"] / max(av["TCP_TCC_WRITE_REQ_sum"], 1)))
for f in sorted(glob.glob(out + "/%s_*" % tag)):
    print("==", f)
    print(open(f).read()[:3500])
