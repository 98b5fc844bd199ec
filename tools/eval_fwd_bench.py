#!/usr/bin/env python3
"""Eval-mode NoisyAct forward (mhaq_fq_act_fwd with q min/max + integrity flags: gdnsq.py:211-217, gdnsq_act.py:51-54)
against the training forward on the four ResNet-18 activation sizes: HIP events, rotated buffers."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mhaq_amd import _lib

L = _lib.lib()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
ls = torch.tensor([math.log2(0.2371)], device=dev)
lq = ls + 4
b = torch.tensor([-1.9], device=dev)
params = torch.empty(5, device=dev)
qstats = torch.empty(2, device=dev)
flags = torch.empty(1, dtype=torch.int32, device=dev)
for n, nbuf in ((50176000, 3), (25088000, 4), (12544000, 6), (6272000, 10)):
    xs = [torch.randn(n, device=dev) * 2 for _ in range(nbuf)]
    ys = [torch.empty(n, device=dev) for _ in range(nbuf)]
    nb = L.mhaq_fq_pt_fwd_workspace_bytes(n)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)

    def train(i):
        k = i % nbuf
        return L.mhaq_fq_act_fwd(xs[k].data_ptr(), ys[k].data_ptr(), n, ls.data_ptr(), lq.data_ptr(), b.data_ptr(),
                                 params.data_ptr(), None, None, None, 0, st)

    def evalf(i):
        k = i % nbuf
        return L.mhaq_fq_act_fwd(xs[k].data_ptr(), ys[k].data_ptr(), n, ls.data_ptr(), lq.data_ptr(), b.data_ptr(),
                                 params.data_ptr(), qstats.data_ptr(), flags.data_ptr(), ws.data_ptr(), nb, st)

    def timed(fn, reps=30):
        for i in range(10):
            assert fn(i) == 0
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(reps):
            fn(i)
        e.record()
        torch.cuda.synchronize()
        return a.elapsed_time(e) / reps
    tt = sorted(timed(train) for _ in range(5))[2]
    te = sorted(timed(evalf) for _ in range(5))[2]
    print(f"{n/1e6:5.1f}M  train fwd {tt*1e3:6.1f} us {8*n/tt/1e6:6.0f} GB/s | eval fwd (+q range, flags, finalize) "
          f"{te*1e3:6.1f} us {8*n/te/1e6:6.0f} GB/s", flush=True)
    del xs, ys
    torch.cuda.empty_cache()
