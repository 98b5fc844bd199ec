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

import mhaq_amd as M
from mhaq_amd import ops
from mhaq_amd.act_hub import ActGradHub


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
    """median of three event-timed rounds of `reps` calls (3 warm-up calls first: clocks ramp over the first ms)"""
    for _ in range(3):
        fn()
    rounds = []
    for _ in range(3):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        rounds.append(a.elapsed_time(b) / reps)
    return sorted(rounds)[1]


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
        # one tensor pair and one quantizer (own parameters, post-calibration state) per NoisyAct of the config, as
        # in a training step; tensors of one shape share storage only when the set would not fit otherwise
        share = sum(math.prod(s) for s in shapes) * 16 > 64e9
        data, pool = [], {}
        acts = torch.nn.ModuleList([M.NoisyAct() for _ in shapes]).to(dev).train()
        for a, shp in zip(acts, shapes):
            if share and shp in pool:
                x, g = pool[shp]
            else:
                x, g = torch.randn(shp, device=dev) * 2, torch.randn(shp, device=dev)
                pool[shp] = (x, g)
            rng_ = float(x.max() - x.min())
            with torch.no_grad():
                a.log_act_s.fill_(math.log2(rng_ / 1023))
                a.log_act_q.fill_(math.log2(rng_ / 1023) + 10)
                a.act_b.fill_(float(x.min()))
            data.append((x, g))
        hub = ActGradHub(acts)
        eager_par = [(a.log_act_s.detach().clone().requires_grad_(True), a.log_act_q.detach().clone().requires_grad_(True),
                      a.act_b.detach().clone().requires_grad_(True)) for a in acts]

        def hip_pass():       # all forwards, then ONE backward over every quantizer, like a training step
            for p in acts.parameters():
                p.grad = None
            hub.begin()
            ys = [a(x.detach().requires_grad_(True)) for a, (x, _) in zip(acts, data)]
            hub.end()
            torch.autograd.backward(ys, [g for _, g in data])

        def eager_pass():
            ys = []
            for (x, _), (ls, lq, b) in zip(data, eager_par):
                ls.grad = lq.grad = b.grad = None
                ys.append(eager_act(x.detach().requires_grad_(True), ls, lq, b))
            torch.autograd.backward(ys, [g for _, g in data])

        def graph_time():     # the same pass captured once and replayed (device-side rate of the product path)
            base = torch.zeros(1, dtype=torch.int64, device=dev)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), ops.rng.device_offset(base):
                hip_pass()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side), ops.rng.device_offset(base):
                drawn = ops.rng.drawn()
                hip_pass()
                base.add_(ops.rng.drawn() - drawn)
            return timeit(graph.replay, args.reps)

        n_act = sum(math.prod(s) for s in shapes)
        t_hip = timeit(hip_pass, args.reps)
        t_graph = graph_time()
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

        # ... and as the trainer runs them: NoisyConv2d modules, ONE model-wide forward launch, the backward in groups
        # of consecutive layers (multi.py; small models: one group)
        from mhaq_amd.multi import MultiTensorWeightQuant
        convs = torch.nn.ModuleList([M.NoisyConv2d(shp[1], shp[0], 3, bias=False, qscheme=M.QScheme.PER_CHANNEL,
                                                   qnmethod=M.QNMethod[method]) for shp in wsh]).to(dev)
        with torch.no_grad():
            for c, (w, ls, _) in zip(convs, wdata):
                c.weight.copy_(w)
                c.log_wght_s.copy_(ls)
        plan = MultiTensorWeightQuant(convs, joint_backward=False, backward_group_elems=4 << 20)

        def hip_w_grouped():
            plan.run()
            outs, grads = [], []
            for c, (_, _, G) in zip(convs, wdata):
                wq, _, _ = c._quantized_weight()
                lwq = c.regulariser_input()
                outs += [wq, lwq]
                grads += [G, torch.ones_like(lwq)]
            torch.autograd.backward(outs, grads)
            for c in convs:
                c.weight.grad = None

        n_w = sum(math.prod(s) for s in wsh)
        t_hw = timeit(hip_w, args.reps)
        t_hwg = timeit(hip_w_grouped, args.reps)
        t_ew = None if args.no_eager else timeit(eager_w, max(2, args.reps // 3))
        out = {"config": cfg, "batch": B, "act_tensors": len(shapes), "act_elements": n_act,
               "act_hip_ms": round(t_hip, 4), "act_hip_GBps": round(20 * n_act / t_hip / 1e6, 1),
               "act_hip_graph_ms": round(t_graph, 4), "act_hip_graph_GBps": round(20 * n_act / t_graph / 1e6, 1),
               "act_eager_gpu_ms": None if t_eager is None else round(t_eager, 3),
               "act_speedup_vs_eager_gpu": None if t_eager is None else round(t_eager / t_hip, 1),
               "weight_tensors": len(wsh), "weight_elements": n_w, "weight_method": method,
               "weight_hip_ms": round(t_hw, 4), "weight_hip_grouped_ms": round(t_hwg, 4),
               "weight_backward_groups": len(plan.groups), "weight_eager_gpu_ms": None if t_ew is None else round(t_ew, 3),
               "weight_speedup_vs_eager_gpu": None if t_ew is None else round(t_ew / t_hw, 1),
               "note": "all forwards then one backward over every quantizer of the config (own parameters and tensors "
                       "per quantizer, joint finalize), through the NoisyAct modules and autograd ops: act_hip_* "
                       "eager (includes ~40-70 us of Python + autograd per op), act_hip_graph_* the same pass "
                       "replayed as a hipGraph; 20 B/elem algorithmic"}
        print(json.dumps(out), flush=True)
        del data, wdata, pool
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
