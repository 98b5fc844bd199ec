#!/usr/bin/env python3
"""Things a kernel-level duration does not show, read off the gfx950 assembly of a source file (hipcc --save-temps with the
product Makefile's flags):
  * flat_load / flat_store: the compiler could not prove a pointer global (a pointer read out of memory): such accesses count
    on vmcnt AND lgkmcnt, so every LDS read waits for them;
  * a global_load followed within three instructions by s_waitcnt vmcnt(0|1): loads going out one round trip after the other
    (a load under a divergent `if` ends in a register copy at the join);
  * scratch: spilled registers.
usage: tools/isa_smells.py fq_pc.hip [-Dflags ...]        exit status 1 if a flat access or scratch is found"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flags = [a for a in sys.argv[2:] if a.startswith("-")]
    base = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "mhaq_amd", "csrc"), "print-flags"],
                          capture_output=True, text=True, check=True).stdout.split()
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc", *base, *flags, "--save-temps", "-c",
                        os.path.join(ROOT, "mhaq_amd", "csrc", src), "-o", "x.o"], cwd=tmp, check=True,
                       stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
        lines = open(os.path.join(tmp, asm)).read().split("\n")
    starts = [i for i, l in enumerate(lines) if l.startswith("_ZN4mhaq") and ":" in l and "@" in l]
    bad = 0
    for s in starts:
        name = lines[s].split(":")[0]
        try:
            e = next(i for i in range(s, len(lines)) if ".amdhsa_kernel" in lines[i])
        except StopIteration:
            continue
        body = [l.strip() for l in lines[s + 1:e] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        flat = sum(1 for l in body if l.startswith(("flat_load", "flat_store")))
        scratch = sum(1 for l in body if l.startswith(("scratch_", "buffer_store_dword v", "buffer_load_dword v")) and "offen" not in l and "off" in l)
        waits = nload = 0
        for i, l in enumerate(body):
            if l.startswith("global_load"):
                nload += 1
                for l2 in body[i + 1:i + 4]:
                    if l2.startswith("global_load"):
                        break
                    m = re.match(r"s_waitcnt vmcnt\((\d+)\)", l2)
                    if m and int(m.group(1)) <= 1:
                        waits += 1
                        break
        if flat or scratch or waits:
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
            print(f"flat {flat:3d}  scratch {scratch:3d}  load-then-wait {waits:3d} of {nload:3d} loads   {dem}")
            bad += flat + scratch
    print(f"{len(starts)} kernels; {bad} flat / scratch instructions")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
