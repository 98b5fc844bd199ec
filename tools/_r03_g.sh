set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 200 ./tools/kbench 50176000 20 > gpurun_out/r03g_kbench.txt 2>&1 || true
timeout -k 10 200 ./tools/glds_triad 50176000 > gpurun_out/r03g_triad.txt 2>&1 || true
timeout -k 10 100 ./tools/glds_triad 6272000 >> gpurun_out/r03g_triad.txt 2>&1 || true
tail -12 gpurun_out/r03g_triad.txt
MHAQ_STEP_GRAPH=0 python tools/step_host_time.py resnet20 128 > gpurun_out/r03g_step_r20b128.txt 2>&1 || true
MHAQ_STEP_GRAPH=auto python tools/step_host_time.py resnet20 128 >> gpurun_out/r03g_step_r20b128.txt 2>&1 || true
MHAQ_STEP_GRAPH=0 python tools/step_host_time.py resnet20 1000 >> gpurun_out/r03g_step_r20b128.txt 2>&1 || true
MHAQ_STEP_GRAPH=0 MHAQ_STEP_QSCHEME=0 python tools/step_host_time.py resnet18 >> gpurun_out/r03g_step_r20b128.txt 2>&1 || true
cat gpurun_out/r03g_step_r20b128.txt | grep -v amdgpu
