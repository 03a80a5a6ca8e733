"""Condenses rocprofv3 output (kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes) into a text summary and a
per-kernel JSON {kernel: {calls, avg_us, fetch_bytes_per_launch, write_bytes_per_launch}} (only k_* kernels)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

raw = sys.argv[1]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def find(pattern):
    return sorted(glob.glob(os.path.join(raw, pattern), recursive=True))


def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0]


stats = {}
print("== kernel trace stats (rocprofv3 --kernel-trace --stats), own kernels ==")
for f in find("trace/**/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        nm = r.get("Name", "")
        if not short(nm).startswith("k_"):
            continue
        stats[short(nm)] = dict(calls=int(r["Calls"]), total_ms=float(r["TotalDurationNs"]) / 1e6, avg_us=float(r["AverageNs"]) / 1e3,
                                pct=float(r["Percentage"]))
        print("%-60s calls=%-5s total_ms=%9.3f avg_us=%10.2f pct=%s" % (short(nm)[:60], r["Calls"], stats[short(nm)]["total_ms"],
                                                                          stats[short(nm)]["avg_us"], r["Percentage"]))


def pmc(dirname, counter):
    acc, cnt = defaultdict(float), defaultdict(int)
    for f in find(dirname + "/**/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = short(r.get("Kernel_Name", ""))
            acc[k] += float(r.get("Counter_Value", 0))
            cnt[k] += 1
    return acc, cnt


fa, fc = pmc("pmc_fetch", "FETCH_SIZE")
wa, wc = pmc("pmc_write", "WRITE_SIZE")
out = {}
print("\n== HBM traffic per launch from PMC (FETCH_SIZE / WRITE_SIZE are in KiB-units of the counter: x1024 = bytes) ==")
for k in sorted(set(fa) | set(wa)):
    fb = fa.get(k, 0) * 1024 / max(fc.get(k, 1), 1)
    wb = wa.get(k, 0) * 1024 / max(wc.get(k, 1), 1)
    out[k] = dict(stats.get(k, {}), fetch_bytes_per_launch=fb, write_bytes_per_launch=wb, launches_profiled=fc.get(k, 0))
    print("%-60s fetch/launch=%.4g B write/launch=%.4g B (n=%d)" % (k[:60], fb, wb, fc.get(k, 0)))
json.dump(dict(structures=S, kernels=out), open(os.path.join(raw, "prof_traffic.json"), "w"), indent=1)
