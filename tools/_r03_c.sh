set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -q -m gpu --maxfail=40 > gpurun_out/r03c_gpu.log 2>&1 || true
tail -30 gpurun_out/r03c_gpu.log
python tools/pc_bench.py 2 4096x4096,8192x8192,1024x16384,50257x768 > gpurun_out/r03c_pc_aewgs.txt 2>&1
python tools/pc_bench.py 3 > gpurun_out/r03c_pc_lsq.txt 2>&1
python tools/pc_bench.py 0 4096x4096,8192x8192,1024x16384,50257x768 > gpurun_out/r03c_pc_ste.txt 2>&1
cat gpurun_out/r03c_pc_aewgs.txt gpurun_out/r03c_pc_lsq.txt gpurun_out/r03c_pc_ste.txt
python tools/host_profile.py > gpurun_out/r03c_host_profile_act.txt 2>&1 || true
head -8 gpurun_out/r03c_host_profile_act.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM -d gpurun_out/r03c_pmc_aewgs -o pmc -- python tools/pc_bench.py 2 8192x8192 > gpurun_out/r03c_pmc_aewgs.log 2>&1
