#!/usr/bin/env python3
"""A/B timing of the streaming kernels of whichever library MHAQ_FQ_LIB points at (tools/variants.sh builds):
mhaq_fq_act_fwd / mhaq_fq_act_bwd_partials on the four ResNet-18 activation sizes, HIP events, rotated buffers.
One line per size: us per launch and algorithmic GB/s."""
import ctypes
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mhaq_amd import _lib

L = _lib.lib()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.path.dirname(_lib.LIB_PATH))
METHOD = int(os.environ.get("MHAQ_AB_METHOD", "0"))      # 0 = STE (in-kernel Philox signs), 3 = LSQ (no random term)
ls = torch.tensor([math.log2(0.2371)], device=dev)
lq = ls + 4
b = torch.tensor([-1.9], device=dev)
params = torch.empty(5, device=dev)
nparts = ctypes.c_int32(0)
out = []
for n, nbuf in ((50176000, 3), (25088000, 4), (12544000, 6), (6272000, 10)):
    xs = [torch.randn(n, device=dev) * 2 for _ in range(nbuf)]
    gs = [torch.randn(n, device=dev) for _ in range(nbuf)]
    ys = [torch.empty(n, device=dev) for _ in range(nbuf)]
    nb = L.mhaq_fq_act_bwd_workspace_bytes(n)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)

    def fwd(i):
        k = i % nbuf
        return L.mhaq_fq_act_fwd(xs[k].data_ptr(), ys[k].data_ptr(), n, ls.data_ptr(), lq.data_ptr(), b.data_ptr(),
                                 params.data_ptr(), None, None, None, 0, st)

    def bwd(i):
        k = i % nbuf
        return L.mhaq_fq_act_bwd_partials(xs[k].data_ptr(), gs[k].data_ptr(), ys[k].data_ptr(), n, params.data_ptr(),
                                          METHOD, None, 1234, i + 1, None, ws.data_ptr(), nb, ctypes.byref(nparts), st)

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
    tf = sorted(timed(fwd) for _ in range(5))[2]
    tb = sorted(timed(bwd) for _ in range(5))[2]
    out.append(f"{n/1e6:5.1f}M fwd {tf*1e3:6.1f} us {8*n/tf/1e6:6.0f} GB/s | bwd {tb*1e3:6.1f} us {12*n/tb/1e6:6.0f} GB/s")
    del xs, gs, ys
    torch.cuda.empty_cache()
print(f"{tag:22s} " + "  ||  ".join(out), flush=True)
