"""GPU (-m gpu): `python bench.py --gpus 2` -- bench.py starts its own two ranks (torch.distributed.run
children, nothing exec'd after a GPU call) and relays rank 0's line; the driver's torchrun form is covered by
the second test.  Rehearsed with two ranks on the one GPU of the test box over gloo: ResNet-18 in
channels_last, SyncBatchNorm, the Sym-KL teacher on its side stream, DDP with bucket views, the packed AEWGS
statistics all-reduce, barrier + MAX-over-ranks timing and the single JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _check(out):
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 2
    assert out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and out["config"]["final_loss"] == out["config"]["final_loss"]
    assert abs(out["value"] - 16 * 2 / (out["ms_per_step"] * 2e-3)) <= 0.01 * out["value"]
    assert out["roofline"]["bound"] == "hbm" and out["cpu_baseline"] is None
    assert 0 < out["aewgs_allreduce_share"] < 1          # 8(d) config 4: the exchange's share of a step
    # the rehearsal runs over gloo: RCCL saw no rank (a real N-GPU run reports rccl_ranks == N)
    assert out["collective_backend"] == "gloo" and out["rccl_ranks"] == 0
    # replicated weights + all-reduced gradients: the same parameters on both ranks; one sign stream per rank
    assert out["data_parallel_check"] == {"params_in_sync": True, "param_abs_sum": out["data_parallel_check"]["param_abs_sum"],
                                          "distinct_sign_streams": 2, "ranks": 2}


ARGS = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--image", "64", "--kernel-reps", "1", "--no-roofline-set",
        "--no-cpu-baseline"]


def test_bench_gpus_2_starts_its_own_ranks():
    env = dict(os.environ, MHAQ_BENCH_BACKEND="gloo", MHAQ_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *ARGS], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "[bench launcher] starting 2 ranks" in r.stderr
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout[-2000:]            # exactly the one JSON line on stdout
    _check(json.loads(lines[0]))


def test_bench_two_ranks_under_torchrun_one_json_line():
    env = dict(os.environ, MHAQ_BENCH_BACKEND="gloo", MHAQ_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), *ARGS]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    _check(json.loads(lines[0]))
