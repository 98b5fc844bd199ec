#!/usr/bin/env python3
"""Per-channel weight kernels (mhaq_fq_wlayer_fwd / _bwd) on their own: HIP-event timing per launch over
rotated buffers, GB/s of algorithmic bytes (fwd 8 B/elem, bwd 12 B/elem)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mhaq_amd import _lib

SHAPES = [(64, 576), (128, 1152), (256, 2304), (512, 4608), (512, 2304), (1000, 512), (4096, 4096), (8192, 8192),
          (1024, 16384), (50257, 768)]


def timed(fn, reps=9, batch=20):
    """median over `reps` of the per-launch time of `batch` back-to-back launches (host overhead hidden)"""
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    ts = []
    for r in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(batch):
            fn(r * batch + i)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / batch)
    return sorted(ts)[len(ts) // 2] * 1e3


def main():
    L = _lib.lib()
    dev = "cuda:0"
    method = int(sys.argv[1]) if len(sys.argv) > 1 else 3        # 3 = LSQ, 2 = AEWGS, 0 = STE (include/mhaq_fq.h)
    shapes = SHAPES
    if len(sys.argv) > 2:                                        # e.g. 8192x8192,50257x768
        shapes = [tuple(int(v) for v in t.split("x")) for t in sys.argv[2].split(",")]
    for co, row in shapes:
        n = co * row
        nb = max(3, min(16, int(1.5e9 // (n * 16)) or 3))
        W = [torch.randn(co, row, device=dev) * 0.05 for _ in range(nb)]
        G = [torch.randn(co, row, device=dev) for _ in range(nb)]
        out = [torch.empty(co, row, device=dev) for _ in range(nb)]
        ls = torch.full((co,), -6.0, device=dev)
        aux = torch.empty(4, co, device=dev)
        gls = torch.empty(co, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def fwd(i):
            k = i % nb
            _lib.check(L.mhaq_fq_wlayer_fwd(W[k].data_ptr(), out[k].data_ptr(), ls.data_ptr(), co, row,
                                            aux[0].data_ptr(), aux[1].data_ptr(), aux[2].data_ptr(),
                                            aux[3].data_ptr(), st), "fwd")

        def bwd(i):
            k = i % nb
            _lib.check(L.mhaq_fq_wlayer_bwd(W[k].data_ptr(), G[k].data_ptr(), out[k].data_ptr(), gls.data_ptr(),
                                            aux[0].data_ptr(), aux[1].data_ptr(), aux[2].data_ptr(), None, co, row,
                                            method, None, None, None, 7, i + 1, None, st), "bwd")

        tf, tb = timed(fwd), timed(bwd)
        print(f"[{co:6d} x {row:6d}] {n*4/1e6:8.1f} MB  fwd {tf:8.1f} us {8*n/tf/1e3:7.0f} GB/s   "
              f"bwd {tb:8.1f} us {12*n/tb/1e3:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
