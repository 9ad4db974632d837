"""Compact per-kernel table from a rocprofv3 --kernel-trace --stats CSV: python tools/kstats.py <kernel_stats.csv> [n]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    name = r["Name"]
    m = re.search(r"unetmi::(\w+<[^(]*>)", name)
    name = m.group(1) if m else name[:60]
    print("{:58s} calls {:5d}  avg {:9.1f} us  total {:8.2f} ms  {:5.1f}%".format(
        name, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
