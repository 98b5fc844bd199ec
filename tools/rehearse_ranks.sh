#!/bin/bash
# One data-parallel rehearsal of `bench.py --gpus N` in the driver's own launch form (torch.distributed.run, one process per
# rank) on a ONE-GPU box: every rank on cuda:0, collectives over gloo (RCCL refuses two ranks on one device), a tiny batch.
# N = 5: the launcher (torch.distributed.run) opens the GPU too, and the box allows 6 processes on one card (VERDICT r5 asked for 8; a 6-rank attempt was killed by that guard: 7 processes
# on the GPU).  What it shows: N ranks start, SyncBatchNorm + DDP + the packed AEWGS exchanges run, the
# parameters stay in sync, every rank draws its own sign stream, rank 0 prints ONE JSON line with the multi-GPU keys.
#   bash tools/rehearse_ranks.sh [N] > gpurun_out/rehearse_N.json
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
N=${1:-5}
export MHAQ_BENCH_BACKEND=gloo MHAQ_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
exec timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus $N --steps 3 --warmup 2 --batch 4 --image 64 --kernel-reps 1 --no-roofline-set --no-configs --no-cpu-baseline
