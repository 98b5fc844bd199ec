#!/bin/bash
# rocprofv3 kernel-trace of kbench (library kernels only) for each variant; prints avg kernel durations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in tools/variants/*/; do
  v=$(basename $d)
  LD_LIBRARY_PATH=$d rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$v -- ./tools/kbench ${1:-50176000} 20 lib > gpurun_out/prof_$v.log 2>&1
  echo "=== $v"; grep -E "mhaq|fused" gpurun_out/prof_$v.log
  python3 - "$v" <<'PY'
import csv,glob,sys
f=glob.glob(f"gpurun_out/prof_{sys.argv[1]}/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "mhaq" in r["Name"]: print("   ", r["Name"][:52], r["Calls"], "avg %.2f us  min %.2f" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
