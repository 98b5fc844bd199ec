#!/usr/bin/env python3
"""How much slower are the kernels SyncBatchNorm uses for N>1 (torch native batch_norm_stats / _elemt /
_backward_reduce / _backward_elemt) than the MIOpen BatchNorm the N=1 step runs?  Times both on the ResNet-18
activation shapes, channels_last and NCHW, without any process group (the collectives themselves move 3*C floats)."""
import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")
shapes = [(250, 64, 112, 112), (250, 64, 56, 56), (250, 128, 28, 28), (250, 256, 14, 14), (250, 512, 7, 7)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for cl in (True, False):
    for shp in shapes:
        C = shp[1]
        x = torch.randn(shp, device=dev)
        dy = torch.randn(shp, device=dev)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
            dy = dy.contiguous(memory_format=torch.channels_last)
        w = torch.ones(C, device=dev, requires_grad=True)
        b = torch.zeros(C, device=dev, requires_grad=True)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        count = torch.full((1,), x.numel() // C, device=dev, dtype=torch.float32)

        def miopen():
            xx = x.detach().requires_grad_(True)
            y = F.batch_norm(xx, rm, rv, w, b, True, 0.1, 1e-5)
            y.backward(dy)

        def native_sync():
            mean, invstd = torch.batch_norm_stats(x, 1e-5)
            mean2, invstd2 = torch.batch_norm_gather_stats_with_counts(
                x, mean.unsqueeze(0), invstd.unsqueeze(0), rm, rv, 0.1, 1e-5, count)
            y = torch.batch_norm_elemt(x, w, b, mean2, invstd2, 1e-5)
            sum_dy, sum_dy_xmu, gw, gb = torch.batch_norm_backward_reduce(dy, x, mean2, invstd2, w, True, True, True)
            gi = torch.batch_norm_backward_elemt(dy, x, mean2, invstd2, w, sum_dy, sum_dy_xmu, count.int())
            return y, gi

        t_m, t_n = timeit(miopen), timeit(native_sync)
        gb_ = x.numel() * 4 / 1e9
        print(f"{'NHWC' if cl else 'NCHW'} {shp}: MIOpen fwd+bwd {t_m:8.1f} us   SyncBN kernels {t_n:8.1f} us   "
              f"(tensor {gb_*1e3:.0f} MB; 8 passes at 6 TB/s = {8*gb_/6e3*1e6:.0f} us)", flush=True)
