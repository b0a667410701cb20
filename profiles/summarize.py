"""Condenses a rocprofv3 output tree (kernel trace stats + PMC csv) into a small text summary."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        print("%-40s calls=%-5s total_ns=%-14s avg_ns=%-12s pct=%s" % (row.get("Name", "")[:40], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
print("== PMC (per kernel: mean over dispatches) ==")
acc = defaultdict(lambda: defaultdict(list))
for f in find("*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")[:32]
        acc[k][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
for k in sorted(acc):
    if "k_" not in k:
        continue
    print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(acc[k].items())}, "n=%d" % max(len(v) for v in acc[k].values()))
