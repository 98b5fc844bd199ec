set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -q -m gpu --maxfail=60 > gpurun_out/r03e_gpu.log 2>&1 || true
tail -60 gpurun_out/r03e_gpu.log
python tools/pc_bench.py 2 4096x4096,8192x8192,1024x16384,50257x768 > gpurun_out/r03e_pc_aewgs.txt 2>&1
python tools/pc_bench.py 3 4096x4096,8192x8192,1024x16384,50257x768 > gpurun_out/r03e_pc_lsq.txt 2>&1
cat gpurun_out/r03e_pc_aewgs.txt gpurun_out/r03e_pc_lsq.txt
