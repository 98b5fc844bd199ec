#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace + PMC passes) into a small text summary.

The kernel table covers only the LAST `steps` training steps of the traced bench run (located by the
16-per-step per-channel weight-forward launches), so MIOpen's first-run algorithm search and the
warm-up steps do not pollute it."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
out = []
for f in glob.glob(f"{root}/trace/*/*_kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # --- the roofline leg of bench.py: stand-alone launches on the [250,64,56,56] tensor (grid = 12250, 24500 or
    #     49000 workgroups of 256 threads), timed there with HIP events; these are the rocprof durations
    out.append("== bench.py roofline leg, [250,64,56,56] fp32 (50,176,000 elements): rocprofv3 kernel durations")
    leg = defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"]
        # the stand-alone entry points use the <..., LOGP/ACT = false> instantiations; the training step uses <true>
        alone = ("pt_bwd_kernel<0, false, true, false, false>" in name) or ("pt_fwd_kernel<false, false, true, false>" in name)
        if alone and int(r["Grid_Size_X"]) in (12250 * 256, 24500 * 256, 49000 * 256):
            leg[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in sorted(leg.items()):
        byt = 12 if "bwd" in k else 8
        out.append(f"   {k[:70]:70s} {len(v):5d} launches  avg {sum(v)/len(v)/1e3:8.2f} us  min {min(v)/1e3:8.2f} us"
                   f"  -> {byt * 50176000 / (sum(v)/len(v)) :8.1f} GB/s algorithmic")
    marks = [i for i, r in enumerate(rows) if "pc_fwd_kernel" in r["Kernel_Name"]]
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
        cols = ["Kernel_Name", "Counter_Name", "Counter_Value", "Grid_Size", "Workgroup_Size", "VGPR_Count",
                "SGPR_Count", "LDS_Block_Size"]
        with open(f"{root}/pmc_{name}_mhaq.csv", "w", newline="") as fh:             # -> profiles/rNN_pmc_*_mhaq.csv
            wr = csv.DictWriter(fh, fieldnames=cols, extrasaction="ignore", quoting=csv.QUOTE_MINIMAL)
            wr.writeheader()
            wr.writerows(keep)
        out.append(f"== PMC pass {name} (bench.py --roofline-only): kernel | counter | dispatches | mean value (KiB)")
        for (k, c), v in sorted(agg.items()):
            out.append(f"{k:70s} {c:12s} {len(v):4d} {sum(v)/len(v):16.1f}")
print("\n".join(out))
open(f"{root}/summary.txt", "w").write("\n".join(out) + "\n")
