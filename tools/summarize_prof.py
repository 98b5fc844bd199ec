#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace + PMC passes) into a small text summary.

The kernel table covers only the LAST `steps` training steps of the traced bench run (located by the
16-per-step per-channel weight-forward launches), so MIOpen's first-run algorithm search and the
warm-up steps do not pollute it."""
import csv
import glob
import sys
from collections import defaultdict

import json
import os
import re

root = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
out = []
traffic = {}                      # kernel -> {"fetch_KiB": mean, "write_KiB": mean, "dispatches": n}
LAYER1 = 50176000                 # [250,64,56,56]


def short(name):
    """'void mhaq::pt_bwd_kernel<0, false, true, false, true, true>(float const*, ...)' -> 'mhaq::pt_bwd_kernel<0, ...>'"""
    m = re.match(r"(?:void )?([^(]+)\(", name)
    return (m.group(1) if m else name).strip()
for f in glob.glob(f"{root}/trace/*/*_kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # --- the roofline leg of bench.py: stand-alone launches on the [250,64,56,56] tensor (grid = 12250, 24500 or
    #     49000 workgroups of 256 threads), timed there with HIP events; these are the rocprof durations
    out.append("== bench.py roofline leg, [250,64,56,56] fp32 (50,176,000 elements): rocprofv3 kernel durations")
    leg = defaultdict(list)
    first_step = next((i for i, r in enumerate(rows) if "pc_fwd" in r["Kernel_Name"]), len(rows))
    for r in rows[:first_step]:          # the leg runs before the model is built: the same instantiations recur in-step
        name = r["Kernel_Name"]
        # the roofline leg launches the instantiations the training step uses: <..., ACT = true> / <..., LOGP = true>
        alone = ("pt_bwd_kernel<0, false, true, false, true, true>" in name) or ("pt_fwd_kernel<false, false, true, true, true, 1>" in name)
        if alone and int(r["Grid_Size_X"]) in (12250 * 256, 24500 * 256, 49000 * 256):
            leg[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in sorted(leg.items()):
        byt = 12 if "bwd" in k else 8
        out.append(f"   {k[:70]:70s} {len(v):5d} launches  avg {sum(v)/len(v)/1e3:8.2f} us  min {min(v)/1e3:8.2f} us"
                   f"  -> {byt * 50176000 / (sum(v)/len(v)) :8.1f} GB/s algorithmic")
    # step boundaries: the model-wide weight forward launch opens every training step (one per step); older builds
    # launched 16 per-layer weight forwards per step instead
    marks = [i for i, r in enumerate(rows) if "pc_fwd_multi" in r["Kernel_Name"]]
    per_step = 1
    if not marks:
        marks = [i for i, r in enumerate(rows) if "pc_fwd" in r["Kernel_Name"]]
        per_step = 16
    start = marks[-steps * per_step] if len(marks) >= steps * per_step else 0
    sel = rows[start:]
    t0, t1 = int(sel[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sel)
    agg = defaultdict(list)
    for r in sel:
        agg[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in agg.values())
    out.append(f"== kernels of the last {steps} training steps (rocprofv3 --kernel-trace): window {1e-6*(t1-t0):.2f} ms, "
               f"GPU busy {1e-6*tot:.2f} ms, {len(sel)} launches ({len(sel)/steps:.0f}/step)")
    out.append("name | calls | avg us | min us | max us | % of GPU time")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:28]:
        out.append(f"{k[:88]:88s} {len(v):6d} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} {max(v)/1e3:10.2f} {100*sum(v)/tot:6.2f}")
    mh = {k: v for k, v in agg.items() if "mhaq" in k}
    out.append(f"-- mhaq kernels: {sum(sum(v) for v in mh.values())/1e6:.3f} ms = {100*sum(sum(v) for v in mh.values())/tot:.2f} % of GPU time; "
               f"{sum(len(v) for v in mh.values())/steps:.0f} launches/step")
    for k, v in sorted(mh.items(), key=lambda kv: -sum(kv[1])):
        out.append(f"   {k[:84]:84s} {len(v):6d} {sum(v)/len(v)/1e3:10.2f} us avg")
    with open(f"{root}/kernel_stats_last_steps.csv", "w", newline="") as fh:     # -> profiles/rNN_bench_kernel_stats.csv
        wr = csv.writer(fh)
        wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            wr.writerow([k, len(v), sum(v), sum(v) / len(v), min(v), max(v)])
    small = [d for v in agg.values() for d in v if d < 10000]
    out.append(f"-- launches shorter than 10 us: {len(small)/steps:.0f}/step, {sum(small)/1e6/steps:.3f} ms/step")
for name in ("fetch", "write"):
    for f in glob.glob(f"{root}/pmc_{name}/*/*_counter_collection.csv"):
        agg = defaultdict(list)
        keep = []
        rd = csv.DictReader(open(f))
        for r in rd:
            if "mhaq" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
                keep.append(r)
                # launches over the layer-1 tensor only (the activation-set leg runs the same kernels on smaller ones)
                grid, wg = int(r["Grid_Size"]), int(r["Workgroup_Size"])
                per_block = {"pt_bwd_kernel": 2048, "pt_fwd_kernel": 1024}
                for key, elems in per_block.items():
                    if key in r["Kernel_Name"] and grid // wg == LAYER1 // elems:
                        t = traffic.setdefault(short(r["Kernel_Name"]), {"fetch_KiB": [], "write_KiB": []})
                        t[f"{name}_KiB"].append(float(r["Counter_Value"]))
        cols = ["Kernel_Name", "Counter_Name", "Counter_Value", "Grid_Size", "Workgroup_Size", "VGPR_Count",
                "SGPR_Count", "LDS_Block_Size"]
        with open(f"{root}/pmc_{name}_mhaq.csv", "w", newline="") as fh:             # -> profiles/rNN_pmc_*_mhaq.csv
            wr = csv.DictWriter(fh, fieldnames=cols, extrasaction="ignore", quoting=csv.QUOTE_MINIMAL)
            wr.writeheader()
            wr.writerows(keep)
        out.append(f"== PMC pass {name} (bench.py --roofline-only): kernel | counter | dispatches | mean value (KiB)")
        for (k, c), v in sorted(agg.items()):
            out.append(f"{k:70s} {c:12s} {len(v):4d} {sum(v)/len(v):16.1f}")
# ---- HBM bytes per launch of the layer-1 streaming kernels -> profiles/rNN_traffic.json (read by bench.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kernels = {}
for k, t in traffic.items():
    if t["fetch_KiB"] and t["write_KiB"]:
        f, w = sum(t["fetch_KiB"]) / len(t["fetch_KiB"]), sum(t["write_KiB"]) / len(t["write_KiB"])
        # MI355X_MICROARCH.md, HBM: FETCH_SIZE / WRITE_SIZE are KiB; gfx950 counts a 16 B/lane streaming read at 1/2
        kernels[k] = {"fetch_KiB": round(f, 1), "write_KiB": round(w, 1), "hbm_bytes": int((2 * f + w) * 1024),
                      "dispatches": [len(t["fetch_KiB"]), len(t["write_KiB"])],
                      "tensor_elements": LAYER1, "rule": "2 x FETCH_SIZE + WRITE_SIZE (KiB) x 1024"}
        out.append(f"== traffic {k}: 2 x {f:.1f} + {w:.1f} KiB = {kernels[k]['hbm_bytes'] / 1e6:.1f} MB per launch")
if kernels:
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_hash", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "tools", "kernel_hash.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(f"{root}/traffic.json", "w") as fh:
        json.dump({"kernel_source_hash": mod.kernel_source_hash(), "kernels": kernels,
                   "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE -- python3 bench.py --roofline-only "
                              "--no-roofline-set --kernel-reps 5 (two separate passes)"}, fh, indent=1)
print("\n".join(out))
open(f"{root}/summary.txt", "w").write("\n".join(out) + "\n")
