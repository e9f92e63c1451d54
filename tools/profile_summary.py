"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into one text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


def short(name):
    name = name.split("(")[0]
    for pre in ("void sx::", "sx::", "void "):
        if name.startswith(pre):
            name = name[len(pre):]
    return name[:70]


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for r in rows[:40]:
        print(f"{short(r['Name']):70s} {r['Calls']:>7s} {float(r['TotalDurationNs']) / 1e6:10.3f} "
              f"{float(r['AverageNs']) / 1e3:10.2f} {float(r['Percentage']):6.2f}")

for label, pat in (("FETCH_SIZE", "pmc_fetch/**/*counter_collection.csv"), ("WRITE_SIZE", "pmc_write/**/*counter_collection.csv")):
    agg = defaultdict(lambda: [0, 0.0])
    for f in find(pat):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != label:
                continue
            k = short(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    print(f"\n== {label} per kernel (sum over dispatches, raw counter units = KiB) ==")
    for k, (cnt, val) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
        print(f"{k:70s} dispatches {cnt:6d}  sum {val:16.0f} KiB  per-dispatch {val / max(cnt, 1):14.1f} KiB")
