#!/usr/bin/env python3
"""Host-side cost of one op call (tiny tensors, so the GPU is never the limit)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhaq_amd import ops, _lib
dev = "cuda:0"
x = torch.randn(4, 8, 8, 8, device=dev, requires_grad=True)
g = torch.randn_like(x)
ls = torch.tensor([-4.0], device=dev, requires_grad=True); lq = torch.tensor([2.0], device=dev, requires_grad=True)
b = torch.tensor([-2.0], device=dev, requires_grad=True)
w = torch.randn(16, 8, 3, 3, device=dev, requires_grad=True); lws = torch.full((16, 1, 1, 1), -6.0, device=dev, requires_grad=True)
G = torch.randn_like(w)
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
y = torch.empty_like(x); params = torch.empty(5, device=dev)
print("raw ctypes act_fwd call      : %.1f us" % t(lambda: L.mhaq_fq_act_fwd(x.data_ptr(), y.data_ptr(), x.numel(), ls.data_ptr(), lq.data_ptr(), b.data_ptr(), params.data_ptr(), None, None, None, 0, st)))
print("torch.empty_like             : %.1f us" % t(lambda: torch.empty_like(x)))
print("current_stream().cuda_stream : %.1f us" % t(lambda: torch.cuda.current_stream().cuda_stream))
with torch.no_grad():
    print("act layer fwd (no grad)      : %.1f us" % t(lambda: ops.fake_quant_act_layer(x, ls, lq, b, "STE")))
print("act layer fwd (grad)         : %.1f us" % t(lambda: ops.fake_quant_act_layer(x, ls, lq, b, "STE")))
def fb():
    yy, _ = ops.fake_quant_act_layer(x, ls, lq, b, "STE"); yy.backward(g)
print("act layer fwd+bwd            : %.1f us" % t(fb))
def wfb():
    wq, zp, s, lwq = ops.fake_quant_weight_layer(w, lws, "LSQ"); wq.backward(G)
print("weight layer fwd+bwd         : %.1f us" % t(wfb))
def eager():
    s_ = torch.exp2(ls); q_ = torch.exp2(lq)
    v = (torch.clamp(x, b, b + q_ - s_) - b) / s_
    yy = (v + (torch.round(v) - v).detach()) * s_ + b; yy.backward(g)
print("torch eager chain fwd+bwd    : %.1f us" % t(eager, 500))
