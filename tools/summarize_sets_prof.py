#!/usr/bin/env python3
"""rocprofv3 --kernel-trace of tools/fq_sweep.py -> per (mhaq kernel, grid) durations: what each streaming launch of the
BASELINE quantizer sets costs ON THE DEVICE (start -> end timestamps, kernel boundaries excluded), with the algorithmic
rate of the streaming kernels (elements recovered from the grid: 1024 per forward block, 2048 per backward block).
usage: tools/summarize_sets_prof.py <rocprof output dir>  > profiles/rNN_sets_rocprofv3_summary.txt"""
import csv
import glob
import re
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(list)
for f in glob.glob(f"{root}/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mhaq" not in r["Kernel_Name"]:
            continue
        m = re.match(r"(?:void )?([^(]+)\(", r["Kernel_Name"])
        name = (m.group(1) if m else r["Kernel_Name"]).strip()
        blocks = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
        agg[(name, blocks, int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("# rocprofv3 --kernel-trace --stats -- python3 tools/fq_sweep.py ...: device-side durations per (kernel, grid)")
print("# kernel | workgroups x threads | launches | avg us | min us | elements | GB/s algorithmic (avg)")
for (name, blocks, wg), v in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    avg = sum(v) / len(v)
    per, byt = (1024, 8) if "pt_fwd_kernel" in name else ((2048, 12) if "pt_bwd_kernel" in name else (0, 0))
    tail = ""
    if per and blocks > 8:
        n = blocks * per
        tail = f"  ~{n / 1e6:7.2f} M  {byt * n / avg:8.1f}"
    print(f"{name[:78]:78s} {blocks:6d} x {wg:4d} {len(v):6d} {avg / 1e3:9.2f} {min(v) / 1e3:9.2f}{tail}")
