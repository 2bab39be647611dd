"""Boil the rocprofv3 output of scratch/profile_bench.sh down to the CSVs kept under profiles/."""
import csv, glob, os, re, sys

out, tag = sys.argv[1], sys.argv[2]


def short(name):
    m = re.search(r"(\w+)\s*(<.*>)?\(", name)
    base = name.split("(")[0].split("::")[-1]
    return re.sub(r"<.*", "", base).strip() or name


# 1. kernel stats
stats = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)
rows = []
for f in stats:
    rows += list(csv.DictReader(open(f)))
with open("%s/%s_kernel_stats_bench_1e9.csv" % (out, tag), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])

# 2. HBM traffic counters of the rolling kernels
with open("%s/%s_pmc_hbm_traffic_bench_1e9.csv" % (out, tag), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "counter", "value_KB", "dispatch_id"])
    for sub in ("fetch", "write"):
        for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "rolling" not in r["Kernel_Name"]:
                    continue
                w.writerow([short(r["Kernel_Name"]), r["Counter_Name"], r["Counter_Value"], r["Dispatch_Id"]])
for f in sorted(glob.glob(out + "/*.csv")) + [out + "/bench.json"]:
    print("==", f)
    print(open(f).read()[:3000])
