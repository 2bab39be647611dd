"""markdown table 'before -> after' from two outputs of scratch/midw_sweep.py (fractions of 8 TB/s on the default route)"""
import re, sys, collections
def load(f):
    t = {}
    for l in open(f):
        m = re.match(r"(\w+)\s+(\d+) rows/window (\w+)\s+auto ([\d.]+) ms ([\d.]+) \((\w+)", l)
        if m:
            t[(m.group(1), int(m.group(2)), m.group(3))] = (float(m.group(4)), float(m.group(5)), m.group(6))
    return t
a, b = load(sys.argv[1]), load(sys.argv[2])
sets = ["Mean", "MinMax", "SumMinMax", "FirstLast", "WAvgStep", "TW4"]
print("| rows per window | Mean | Min+Max | Sum+Min+Max | First+Last | WeightedAverageStep | all four integrals |")
print("|---|---|---|---|---|---|---|")
for d, label in (("dense", "regular, no nulls"), ("sparse", "irregular, 30 % nulls")):
    for r in (16, 24, 32, 48, 64, 96, 128, 192, 256):
        if (d, r, "Mean") not in b:
            continue
        cells = []
        for s in sets:
            x, y = a.get((d, r, s)), b.get((d, r, s))
            cells.append("%s → %.2f%s" % ("%.2f" % x[1] if x else "-", y[1], "" if y[2].startswith("r_") else " (s)"))
        print("| %s%d | %s |" % (label + ": " if r == 16 else "", r, " | ".join(cells)))
print("\n(s): served by the streaming form; the others by the tile kernels")
