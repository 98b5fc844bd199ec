#!/usr/bin/env python3
"""PotentialLoss hinge kernels on ResNet-18's vectors (3840 weight channels, 16 activation quantizers): HIP-event time per
forward / backward launch through the C ABI (mhaq_fq_potential_loss_fwd / _bwd)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mhaq_amd import _lib  # noqa: E402

L = _lib.lib()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
nw, na = 3840, 16
lws, lwq = torch.randn(nw, device=dev) - 6, torch.randn(nw, device=dev) - 2
las, laq = torch.randn(na, device=dev) - 4, torch.randn(na, device=dev)
base = torch.tensor([1.7], device=dev)
state = torch.tensor([3.3, 3.0, 0.35], device=dev)
out = torch.empty(12, device=dev)
g = torch.ones(1, device=dev)
gb, gas, gaq, gws, gwq = (torch.empty_like(t) for t in (base, las, laq, lws, lwq))


def fwd():
    assert L.mhaq_fq_potential_loss_fwd(base.data_ptr(), las.data_ptr(), laq.data_ptr(), na, lws.data_ptr(), lwq.data_ptr(), nw,
                                        4.0, 4.0, 1.0, 0, state.data_ptr(), 0, out.data_ptr(), st) == 0


def bwd():
    assert L.mhaq_fq_potential_loss_bwd(g.data_ptr(), out.data_ptr(), las.data_ptr(), laq.data_ptr(), na, lws.data_ptr(),
                                        lwq.data_ptr(), nw, 4.0, 4.0, 1.0, gb.data_ptr(), gas.data_ptr(), gaq.data_ptr(),
                                        gws.data_ptr(), gwq.data_ptr(), st) == 0


for name, fn in (("forward", fwd), ("backward", bwd)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 100 * 1e3)
    print(f"potential_loss {name}: {sorted(ts)[3]:.2f} us per launch (back to back)")
