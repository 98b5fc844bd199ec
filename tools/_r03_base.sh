set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r03a_build.log 2>&1
python tools/fq_sweep.py --no-eager --configs resnet20:128,resnet20:1000,rfdn:24 > gpurun_out/r03a_fq_sweep_before.jsonl 2> gpurun_out/r03a_fq_sweep_before.err
python tools/step_host_profile.py 128 > gpurun_out/r03a_host_profile_r20b128.txt 2>&1
python tools/host_profile.py > gpurun_out/r03a_host_profile_act.txt 2>&1
python tools/pc_bench.py 2 4096x4096,8192x8192,1024x16384,50257x768 > gpurun_out/r03a_pc_aewgs_before.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM -d gpurun_out/r03a_pmc_aewgs -o pmc -- python tools/pc_bench.py 2 8192x8192 > gpurun_out/r03a_pmc_aewgs.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM -d gpurun_out/r03a_pmc_lsq -o pmc -- python tools/pc_bench.py 3 8192x8192 > gpurun_out/r03a_pmc_lsq.log 2>&1
ls -la gpurun_out/r03a_pmc_aewgs gpurun_out/r03a_pmc_lsq
