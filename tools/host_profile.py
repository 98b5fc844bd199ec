#!/usr/bin/env python3
"""cProfile of the host side of a NoisyAct forward + backward through the product path (tiny tensors: the GPU
is never the limit), to see where the ~40-50 us per op go."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mhaq_amd as M
from mhaq_amd.act_hub import ActGradHub

dev = "cuda:0"
acts = torch.nn.ModuleList([M.NoisyAct(init_s=-4, init_q=2) for _ in range(16)]).to(dev).train()
xs = [torch.randn(4, 8, 8, 8, device=dev) for _ in range(16)]
gs = [torch.randn(4, 8, 8, 8, device=dev) for _ in range(16)]
hub = ActGradHub(acts)


def step(use_hub=True):
    for p in acts.parameters():
        p.grad = None
    if use_hub:
        hub.begin()
    outs = [a(x.detach().requires_grad_(True)) for a, x in zip(acts, xs)]
    if use_hub:
        hub.end()
    torch.autograd.backward(outs, gs)


def fwd_only():
    with torch.enable_grad():
        return [a(x.detach().requires_grad_(True)) for a, x in zip(acts, xs)]


for _ in range(50):
    step()
torch.cuda.synchronize()
for name, fn in (("fwd+bwd with hub", step), ("fwd+bwd without hub", lambda: step(False)), ("fwd only", fwd_only)):
    t0 = time.perf_counter()
    for _ in range(300):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 300 / 16 * 1e6:.1f} us per quantizer")
from mhaq_amd._ext import ext
ext().host_timers(True)
for _ in range(300):
    step()
torch.cuda.synchronize()
print("compiled nodes, mean host us per call (calls):",
      {k: (round(v[1], 2), v[0]) for k, v in ext().host_timers(True).items()})
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
