set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/_dbg_hub.py > gpurun_out/r03b_dbg.txt 2>&1 || true
tail -12 gpurun_out/r03b_dbg.txt
timeout -k 10 900 python -m pytest tests -q -m gpu --maxfail=40 > gpurun_out/r03b_gpu.log 2>&1 || true
tail -30 gpurun_out/r03b_gpu.log
python tools/fq_sweep.py --no-eager --configs resnet20:128,resnet20:1000,rfdn:24 > gpurun_out/r03b_fq_sweep.jsonl 2> gpurun_out/r03b_fq_sweep.err || true
python tools/step_host_profile.py 128 > gpurun_out/r03b_host_profile_r20b128.txt 2>&1 || true
python tools/host_profile.py > gpurun_out/r03b_host_profile_act.txt 2>&1 || true
python tools/host_overhead.py > gpurun_out/r03b_host_overhead.txt 2>&1 || true
head -8 gpurun_out/r03b_host_profile_r20b128.txt
