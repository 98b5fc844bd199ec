#!/usr/bin/env python3
"""Stability soak of the compiled binding: N eager QAT steps of ResNet-20 (batch 32, AEWGS weights + STE activations,
distillation) with varying batch sizes, then N captured steps; device memory, host RSS and the hub / plan retention
counters must stay flat after the first steps, losses finite."""
import os
import resource
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mhaq_amd as M
from mhaq_amd import nets, ops
from mhaq_amd._ext import ext
from mhaq_amd.qat import QATConfig, QATTrainer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda:0")
torch.manual_seed(0)
ops.manual_seed(0)
cfg = QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.AEWGS, act_bit=4, weight_bit=4,
                excluded_layers=("features.init_block.conv", "output"))
calib = torch.randn(32, 3, 32, 32, device=dev)


def rss_mb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0


for mode in (False, True):
    tr = QATTrainer(nets.resnet20_cifar(10), cfg, dev, calib_batches=[calib], capture_graph=mode)
    sizes = [32, 32, 32, 32, 24, 32, 32, 16, 32, 32] if mode else [32, 8, 24, 16, 32, 12, 28, 20, 4, 32]
    data = {n: (torch.randn(n, 3, 32, 32, device=dev), torch.randint(0, 10, (n,), device=dev)) for n in set(sizes)}
    marks = []
    for i in range(N):
        n = sizes[i % len(sizes)]
        loss = tr.train_step(*data[n])
        if i in (N // 3, N - 1):
            torch.cuda.synchronize()
            assert bool(torch.isfinite(loss)), (mode, i)
            marks.append((torch.cuda.memory_allocated(), rss_mb(), tr.act_hub.state(), ext().plan_state(tr.weight_forward.plan_id)))
    (m1, r1, h1, p1), (m2, r2, h2, p2) = marks
    print(f"capture_graph={mode}: device MB {m1 / 2**20:.1f} -> {m2 / 2**20:.1f}, peak host RSS MB {r1:.0f} -> {r2:.0f}, "
          f"hub {h1} -> {h2}, plan {p1} -> {p2}, final loss {float(loss):.4f}", flush=True)
    assert m2 <= m1 + (1 << 20), "device memory grows"
    assert r2 <= r1 + 64, "host memory grows"
    assert h2["tables"] == h1["tables"] and h2["retired"] == h1["retired"]
    del tr
print("SOAK_OK")
