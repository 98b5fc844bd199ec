#!/bin/bash
# rocprofv3 passes over bench.py; tools/summarize_prof.py condenses them, the summaries are copied to profiles/.
#   pass 1: --kernel-trace --stats over a short full bench run
#   pass 2/3: --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes: TCC has 4 slots, FETCH_SIZE takes 3)
# usage (on the GPU box): bash tools/profile_bench.sh r02      -> gpurun_out/prof_r02/{summary.txt,traffic.json,...}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r03}
O=gpurun_out/prof_$R
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline-set --no-configs > $O/bench_trace.log 2>&1
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --roofline-only --no-roofline-set --no-configs --kernel-reps 5 > $O/bench_pmc_fetch.log 2>&1
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --roofline-only --no-roofline-set --no-configs --kernel-reps 5 > $O/bench_pmc_write.log 2>&1
echo "write rc=$?"
python3 tools/summarize_prof.py $O
