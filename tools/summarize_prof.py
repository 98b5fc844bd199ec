#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small text summary."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
out = []
for f in glob.glob(f"{root}/trace/*/*_kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    out.append("== kernel stats (rocprofv3 --kernel-trace --stats): name | calls | avg us | min us | max us | % of GPU time")
    for r in rows[:25]:
        out.append(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:10.2f} {float(r['MinNs'])/1e3:10.2f} "
                   f"{float(r['MaxNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}")
    mh = [r for r in rows if "mhaq" in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out.append(f"-- mhaq kernels: {sum(float(r['TotalDurationNs']) for r in mh)/1e6:.3f} ms of {tot/1e6:.3f} ms GPU time")
for name in ("fetch", "write"):
    for f in glob.glob(f"{root}/pmc_{name}/*/*_counter_collection.csv"):
        agg = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "mhaq" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
        out.append(f"== PMC pass {name}: kernel | counter | dispatches | mean value")
        for (k, c), v in sorted(agg.items()):
            out.append(f"{k:70s} {c:12s} {len(v):4d} {sum(v)/len(v):16.1f}")
print("\n".join(out))
open(f"{root}/summary.txt", "w").write("\n".join(out) + "\n")
