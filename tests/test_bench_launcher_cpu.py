"""CPU: `python bench.py --gpus N` (N > 1, no WORLD_SIZE) must start N ranks itself through
torch.distributed.run before anything touches the GPU (the reference gets its ranks from Lightning's DDP
launcher, training/trainer.py:92-97).  Without a GPU every rank stops at the "needs an MI355X" check: what
is tested here is that N fresh rank processes were started, each saw its RANK / WORLD_SIZE, the parent
relayed their output and returned a failing status (no result line)."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_starts_two_ranks_and_reports_failure_without_gpu():
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: covered by tests/test_gpu_bench_two_ranks.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0
    assert "[bench launcher] starting 2 ranks" in r.stderr
    # every rank that got as far as main() passed the `--gpus == WORLD_SIZE` check and stopped at the GPU check; the
    # elastic agent ends the remaining rank as soon as the first one has failed, so the second message may be cut off
    n = r.stderr.count("bench.py needs an MI355X")
    assert 1 <= n <= 2 and "but WORLD_SIZE=" not in r.stderr, r.stderr[-2000:]
    assert r.stdout.strip() == ""                      # no JSON line from a failed launch


def test_bench_refuses_mismatched_world_size():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
