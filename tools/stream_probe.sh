#!/bin/bash
# does the teacher's side stream slow the student's critical path?  Three orderings of the
# same ResNet-18 AEWGS distillation step, each timed plain (ms/step) and under rocprofv3 --kernel-trace (in-step
# per-kernel averages of the activation fake-quant kernels).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/stream_probe
mkdir -p $O
for v in overlap "overlap_hp --student-high-priority" "serial --no-teacher-overlap"; do
  set -- $v; name=$1; shift
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-set "$@" > $O/$name.json 2> $O/$name.err
  echo "$name rc=$? $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['ms_per_step'], d['value'])")"
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_$name -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline-set "$@" > $O/trace_$name.log 2>&1
  python3 - "$O/trace_$name" "$name" <<'PY'
import csv, glob, sys
from collections import defaultdict
root, name = sys.argv[1], sys.argv[2]
for f in glob.glob(f"{root}/*/*_kernel_trace.csv"):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "pc_fwd_multi" in r["Kernel_Name"]]      # one per training step
    sel = rows[marks[-5]:] if len(marks) >= 5 else rows
    agg = defaultdict(list)
    for r in sel:
        if "mhaq" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][-60:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"   {name:10s} {k:60s} {len(v):4d} launches  avg {sum(v)/len(v)/1e3:8.2f} us  max {max(v)/1e3:8.2f} us  sum {sum(v)/5e6:7.3f} ms/step")
PY
done
