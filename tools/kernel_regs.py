#!/usr/bin/env python3
"""Register / scratch / LDS use per kernel, from the code-object metadata hipcc writes with --save-temps.
usage: tools/kernel_regs.py fq_pc.hip [substring ...] [-Dflags ...]   (compiles with the product Makefile's flags)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flags = [a for a in sys.argv[2:] if a.startswith("-")]
    subs = [a for a in sys.argv[2:] if not a.startswith("-")]
    base = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "mhaq_amd", "csrc"), "print-flags"],
                          capture_output=True, text=True, check=True).stdout.split()
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc", *base, *flags, "--save-temps", "-c",
                        os.path.join(ROOT, "mhaq_amd", "csrc", src), "-o", "x.o"], cwd=tmp, check=True,
                       stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
        text = open(os.path.join(tmp, asm)).read()
    rows = []
    for blk in re.split(r"\n  - \.agpr_count:", text)[1:]:
        f = {k: v for k, v in re.findall(r"\n\s+\.(name|vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size"
                                         r"|max_flat_workgroup_size):\s+(\S+)", blk)}
        name = subprocess.run(["c++filt", f["name"]], capture_output=True, text=True).stdout.strip()
        name = name.split("(")[0].replace("void ", "")
        if subs and not any(s in name for s in subs):
            continue
        rows.append((name, int(f["vgpr_count"]), int(f["sgpr_count"]), int(f["private_segment_fixed_size"]),
                     int(f["group_segment_fixed_size"]), int(f["max_flat_workgroup_size"])))
    for name, vg, sg, scr, lds, wg in sorted(rows):
        waves = min(8, 512 // max(vg, 1))
        print(f"{name:75s} vgpr {vg:3d} (<= {waves} waves/SIMD)  sgpr {sg:3d}  scratch {scr:4d}  lds {lds:6d}  wg {wg}")


if __name__ == "__main__":
    main()
