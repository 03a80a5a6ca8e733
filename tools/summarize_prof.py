"""Condenses rocprofv3 output directories (kernel-trace stats + PMC csv) into one text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel trace stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:25]:
        print("%-70s calls=%-6s total_ms=%10.3f avg_us=%10.2f pct=%s" % (
            r.get("Name", "")[:70], r.get("Calls"), float(r.get("TotalDurationNs", 0)) / 1e6,
            float(r.get("AverageNs", 0)) / 1e3, r.get("Percentage")))


def pmc(dirname):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for f in find(dirname + "/**/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            acc[k][r.get("Counter_Name")] += float(r.get("Counter_Value", 0))
    return acc


for d in ("pmc_sq", "pmc_fetch", "pmc_write"):
    acc = pmc(d)
    if not acc:
        continue
    print(f"\n== {d} (summed over dispatches) ==")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:14]:
        print("%-60s %s" % (k[:60], " ".join(f"{n}={x:.4g}" for n, x in sorted(v.items()))))
