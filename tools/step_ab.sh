#!/bin/bash
# end-to-end A/B of library variants (tools/variants.sh): ms per training step of the bench workload
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline-set 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2; do
  echo "product $(run)"
  for d in tools/variants/*/; do echo "$(basename $d) $(MHAQ_FQ_LIB=$PWD/${d}libmhaq_fq.so run)"; done
done
