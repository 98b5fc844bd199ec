#!/usr/bin/env python3
"""cProfile of the HOST side of one eager QAT step (default: ResNet-20, batch 128 -- BASELINE configs[0]'s shape on
the GPU, the host-bound case): where do the milliseconds between two launches go?  Prints wall time per step and the
top functions by own time and by cumulative time."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402,F401  (seeds the MIOpen user db before torch loads MIOpen)
import torch  # noqa: E402

from mhaq_amd import nets, ops  # noqa: E402
from mhaq_amd.enums import QNMethod, QScheme  # noqa: E402
from mhaq_amd.qat import QATConfig, QATTrainer  # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
ops.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = QATConfig(qscheme=QScheme.PER_CHANNEL, qnmethod=QNMethod.STE, act_bit=4, weight_bit=4,
                excluded_layers=("features.init_block.conv", "output"))
net = nets.resnet20_cifar(100).to(memory_format=torch.channels_last)
x = torch.randn(B, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
y = torch.randint(0, 100, (B,), device=dev)
tr = QATTrainer(net, cfg, dev, calib_batches=[x[:64]], capture_graph=False)
for _ in range(10):
    tr.train_step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    tr.train_step(x, y)
torch.cuda.synchronize()
print(f"eager step: {(time.perf_counter() - t0) / 50 * 1e3:.2f} ms", flush=True)
# phases, host time only (sync between them)
def phase(name, fn, n=30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"  {name}: host {(t1 - t0) / n * 1e3:.2f} ms, with sync {(time.perf_counter() - t0) / n * 1e3:.2f} ms", flush=True)
    return out
tr.module.train()
with torch.no_grad():
    phase("teacher forward", lambda: tr.teacher(x))
out = phase("student forward (+regulariser inputs)", lambda: tr.module(x))
fp = tr.teacher(x)
def fb():
    o = tr.module(x)
    loss = tr.loss(o, fp)
    tr.optimizer.zero_grad(set_to_none=True)
    loss.backward()
    return loss
phase("student forward + loss + backward", fb)
phase("optimizer.step", lambda: tr.optimizer.step())
pr = cProfile.Profile()
pr.enable()
for _ in range(40):
    tr.train_step(x, y)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumtime").print_stats(45)
