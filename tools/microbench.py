#!/usr/bin/env python3
"""Kernel-level timing of the fake-quant path on synthetic BASELINE tensors (HIP events on the
launch stream, buffer rotation to defeat the 256 MB Infinity Cache).  Prints GB/s of ALGORITHMIC
bytes: fwd 8 B/elem, bwd 12 B/elem."""
import argparse
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mhaq_amd import _lib


def time_calls(fn, reps, warmup=3):
    for _ in range(warmup):
        fn(0)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps):
        evs[i][0].record()
        fn(i)
        evs[i][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="250,64,56,56")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--nbuf", type=int, default=3)
    args = ap.parse_args()
    shape = tuple(int(v) for v in args.shape.split(","))
    dev = "cuda:0"
    L = _lib.lib()
    n = 1
    for v in shape:
        n *= v
    xs = [torch.randn(shape, device=dev) * 2 for _ in range(args.nbuf)]
    gs = [torch.randn(shape, device=dev) for _ in range(args.nbuf)]
    ys = [torch.empty(shape, device=dev) for _ in range(args.nbuf)]
    s = torch.tensor([0.2371], device=dev)
    b = torch.tensor([-1.9], device=dev)
    hi = b + 16 * s - s
    grads = torch.empty(5, device=dev)
    nb = L.mhaq_fq_pt_bwd_workspace_bytes(n)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def fwd(i):
        k = i % args.nbuf
        L.mhaq_fq_pt_fwd(xs[k].data_ptr(), ys[k].data_ptr(), n, s.data_ptr(), b.data_ptr(), b.data_ptr(),
                         hi.data_ptr(), None, None, None, None, 0, st)

    def bwd(method):
        def f(i):
            k = i % args.nbuf
            L.mhaq_fq_pt_bwd(xs[k].data_ptr(), gs[k].data_ptr(), ys[k].data_ptr(), n, s.data_ptr(), b.data_ptr(),
                             b.data_ptr(), hi.data_ptr(), method, None, 0, None, 1234, i + 1, None, 0, grads.data_ptr(),
                             ws.data_ptr(), nb, st)
        return f

    def copy(i):
        k = i % args.nbuf
        ys[k].copy_(xs[k])

    def eager_fwd(i):
        k = i % args.nbuf
        v = (torch.clamp(xs[k], b, hi) - b) / s
        ys[k] = (v + (torch.round(v) - v)) * s + b

    med, best = time_calls(copy, args.reps)
    print(f"torch copy_        : {med:8.3f} ms (best {best:.3f})  {8*n/med/1e6:8.1f} GB/s")
    med, best = time_calls(fwd, args.reps)
    print(f"pt_fwd             : {med:8.3f} ms (best {best:.3f})  {8*n/med/1e6:8.1f} GB/s")
    for name, m in (("STE", 0), ("LSQ", 3)):
        med, best = time_calls(bwd(m), args.reps)
        print(f"pt_bwd {name} (+final): {med:8.3f} ms (best {best:.3f})  {12*n/med/1e6:8.1f} GB/s")
    med, best = time_calls(eager_fwd, max(3, args.reps // 4))
    print(f"torch eager fwd    : {med:8.3f} ms (best {best:.3f})  {8*n/med/1e6:8.1f} GB/s (algorithmic)")


if __name__ == "__main__":
    main()
