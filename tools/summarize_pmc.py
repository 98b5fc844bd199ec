#!/usr/bin/env python3
"""rocprofv3 --pmc ... --output-format csv  ->  per (mhaq kernel, grid) mean counter values and the derived figures the
guide (MI355X_MICROARCH.md) prescribes: VALU wave-instructions per workgroup, VALU-busy share of the wave cycles.
usage: tools/summarize_pmc.py <rocprof output dir> [kernel substring]"""
import csv
import glob
import re
import sys
from collections import defaultdict

root, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "mhaq")
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"{root}/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if filt not in r["Kernel_Name"]:
            continue
        m = re.match(r"(?:void )?([^(]+)\(", r["Kernel_Name"])
        name = (m.group(1) if m else r["Kernel_Name"]).strip()
        key = (name, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1), int(r["Workgroup_Size"]),
               r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?"))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# kernel | workgroups x threads | VGPRs | LDS B | dispatches | mean counters")
for key, cs in sorted(agg.items()):
    name, blocks, wg, vg, lds = key
    n = max(len(v) for v in cs.values())
    mean = {c: sum(v) / len(v) for c, v in cs.items()}
    line = f"{name[:70]:70s} {blocks:6d} x {wg:4d}  vgpr {vg:>3}  lds {lds:>6}  n={n:4d} "
    line += "  ".join(f"{c} {mean[c]:.4g}" for c in sorted(mean))
    if "SQ_INSTS_VALU" in mean:
        line += f"  | VALU wave-instr per workgroup {mean['SQ_INSTS_VALU'] / blocks:.0f}"
    if "SQ_ACTIVE_INST_VALU" in mean and "SQ_WAVE_CYCLES" in mean and mean["SQ_WAVE_CYCLES"]:
        line += f"  VALU-active / wave-cycles {4 * mean['SQ_ACTIVE_INST_VALU'] / mean['SQ_WAVE_CYCLES']:.3f}"
    if "SQ_WAIT_ANY" in mean and "SQ_WAVE_CYCLES" in mean and mean["SQ_WAVE_CYCLES"]:
        line += f"  waiting / wave-cycles {mean['SQ_WAIT_ANY'] / mean['SQ_WAVE_CYCLES']:.3f}"
    print(line)
