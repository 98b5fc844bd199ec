#!/usr/bin/env python3
"""Fake-quant path only: time the HIP ops over the activation / weight tensor sets of the BASELINE configs
 with HIP events, against the same op chain in torch eager on the GPU.
Reports algorithmic GB/s (20 B/elem fused fwd+bwd) per config as one JSON line per config."""
import argparse
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mhaq_amd import ops


class _EagerNoise(torch.autograd.Function):
    """Comparison leg only: the reference's eager op chain written out with torch ops on the GPU
    (round noise with a straight-through input gradient and the random scale gradient)."""

    @staticmethod
    def forward(ctx, v, s):
        ctx.save_for_backward(v)
        return torch.round(v) - v

    @staticmethod
    def backward(ctx, g):
        (v,) = ctx.saved_tensors
        r = torch.randint_like(v, 2).sub_(0.5)
        return g * 0, (3.0 ** -0.5) * g * r


def eager_act(x, ls, lq, b):
    s, q = torch.exp2(ls), torch.exp2(lq)
    v = (torch.clamp(x, min=b, max=b + q - s) - b) / s
    return (v + _EagerNoise.apply(v, s)) * s + b


def eager_weight(w, ls):
    s = torch.exp2(ls)
    zp = w.amin((1, 2, 3), keepdim=True)
    v = (w - zp) / s
    return (v + _EagerNoise.apply(v, s)) * s + zp


def act_shapes(cfg, B):
    if cfg == "resnet20":
        return [(B, 16, 32, 32)] * 7 + [(B, 32, 16, 16)] * 6 + [(B, 64, 8, 8)] * 5
    if cfg == "resnet18":
        return [(B, 64, 56, 56)] * 5 + [(B, 128, 28, 28)] * 4 + [(B, 256, 14, 14)] * 4 + [(B, 512, 7, 7)] * 3
    if cfg == "rfdn":       # reference training shape h = w = 24
        h = 24
        per = [(B, 50, h, h)] * 4 + [(B, 12, h, h)] + [(B, 12, 2, 2)] * 3
        return per * 4 + [(B, 50, h, h)]
    if cfg == "rfdn_stress":
        per = [(B, 50, 180, 320)] * 4 + [(B, 12, 180, 320)] + [(B, 12, 15, 26)] * 3
        return per * 4 + [(B, 50, 180, 320)]
    raise ValueError(cfg)


def weight_shapes(cfg):
    if cfg == "resnet20":
        return [(16, 16, 3, 3)] * 6 + [(32, 16, 3, 3)] + [(32, 32, 3, 3)] * 5 + [(64, 32, 3, 3)] + [(64, 64, 3, 3)] * 5
    if cfg == "resnet18":
        return [(64, 64, 3, 3)] * 4 + [(128, 64, 3, 3)] + [(128, 128, 3, 3)] * 3 + [(256, 128, 3, 3)] + \
               [(256, 256, 3, 3)] * 3 + [(512, 256, 3, 3)] + [(512, 512, 3, 3)] * 3
    per = [(50, 50, 3, 3)] * 3 + [(25, 50, 3, 3)] + [(12, 12, 3, 3)] * 4
    return per * 4 + [(50, 50, 3, 3)]


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="resnet20:128,resnet20:1000,resnet18:250,rfdn:24,rfdn_stress:24")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--no-eager", action="store_true")
    args = ap.parse_args()
    dev = "cuda:0"
    for item in args.configs.split(","):
        cfg, B = item.split(":")
        B = int(B)
        shapes = act_shapes(cfg, B)
        uniq = sorted(set(shapes), key=lambda s: -math.prod(s))
        data = {}
        for shp in uniq:
            x = torch.randn(shp, device=dev) * 2
            g = torch.randn(shp, device=dev)
            rng_ = float(x.max() - x.min())
            ls = torch.tensor([math.log2(rng_ / 1023)], device=dev, requires_grad=True)
            lq = torch.tensor([math.log2(rng_ / 1023) + 10], device=dev, requires_grad=True)
            b = torch.tensor([float(x.min())], device=dev, requires_grad=True)
            data[shp] = (x.requires_grad_(True), g, ls, lq, b)

        def hip_pass():       # all forwards, then ONE backward over every quantizer, like a training step
            ys, gs = [], []
            for shp in shapes:
                x, g, ls, lq, b = data[shp]
                xi = x.detach().requires_grad_(True)      # own leaf per quantizer (shared storage): no grad accumulation
                ys.append(ops.fake_quant_act_layer(xi, ls, lq, b, "STE")[0])
                gs.append(g)
            torch.autograd.backward(ys, gs)

        def eager_pass():
            ys, gs = [], []
            for shp in shapes:
                x, g, ls, lq, b = data[shp]
                xi = x.detach().requires_grad_(True)
                ys.append(eager_act(xi, ls, lq, b))
                gs.append(g)
            torch.autograd.backward(ys, gs)

        n_act = sum(math.prod(s) for s in shapes)
        t_hip = timeit(hip_pass, args.reps)
        t_eager = None if args.no_eager or n_act > 2.5e9 else timeit(eager_pass, max(2, args.reps // 3))
        # weights
        wsh = weight_shapes("rfdn" if cfg.startswith("rfdn") else cfg)
        wdata = []
        for shp in wsh:
            w = (torch.randn(shp, device=dev) * math.sqrt(2.0 / (shp[1] * 9))).requires_grad_(True)
            span = (w.detach().amax((1, 2, 3)) - w.detach().amin((1, 2, 3)))
            ls = torch.maximum(torch.full((shp[0],), -12.0, device=dev), torch.log2(span / 1023)).reshape(-1, 1, 1, 1)
            wdata.append((w, ls.requires_grad_(True), torch.randn(shp, device=dev)))
        method = "AEWGS" if cfg == "resnet18" else ("LSQ" if cfg.startswith("rfdn") else "STE")

        def hip_w():
            outs, grads = [], []
            for w, ls, G in wdata:
                wq, zp, s, lwq = ops.fake_quant_weight_layer(w, ls, method)
                outs += [wq, lwq]
                grads += [G, torch.ones_like(lwq)]
            torch.autograd.backward(outs, grads)
            for w, ls, G in wdata:
                w.grad = None

        def eager_w():
            outs, grads = [], []
            for w, ls, G in wdata:
                wq = eager_weight(w, ls)      # STE-style estimator for every method: same op count
                lwq = torch.log2(w.amax((1, 2, 3)) - w.amin((1, 2, 3)) + torch.exp2(ls.ravel()))
                outs += [wq, lwq]
                grads += [G, torch.ones_like(lwq)]
            torch.autograd.backward(outs, grads)
            for w, ls, G in wdata:
                w.grad = None

        n_w = sum(math.prod(s) for s in wsh)
        t_hw = timeit(hip_w, args.reps)
        t_ew = None if args.no_eager else timeit(eager_w, max(2, args.reps // 3))
        out = {"config": cfg, "batch": B, "act_tensors": len(shapes), "act_elements": n_act,
               "act_hip_ms": round(t_hip, 4), "act_hip_GBps": round(20 * n_act / t_hip / 1e6, 1),
               "act_eager_gpu_ms": None if t_eager is None else round(t_eager, 3),
               "act_speedup_vs_eager_gpu": None if t_eager is None else round(t_eager / t_hip, 1),
               "weight_tensors": len(wsh), "weight_elements": n_w, "weight_method": method,
               "weight_hip_ms": round(t_hw, 4), "weight_eager_gpu_ms": None if t_ew is None else round(t_ew, 3),
               "weight_speedup_vs_eager_gpu": None if t_ew is None else round(t_ew / t_hw, 1),
               "note": "all forwards then one backward over every quantizer of the config, through the autograd "
                       "ops (includes Python and launch overhead); 20 B/elem algorithmic"}
        print(json.dumps(out), flush=True)
        del data, wdata
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
