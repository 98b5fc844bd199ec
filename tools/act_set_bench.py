#!/usr/bin/env python3
"""Where the ResNet-18 W4A4 activation set (16 tensors, 420 M elements) loses time
against the layer-1 rate: per-size kernel rates through the raw C ABI (no Python between launches beyond one
ctypes call), then the 16-tensor forward + backward sequence three ways -- raw C ABI calls, the autograd ops,
and the autograd ops captured in one hipGraph.  One JSON line."""
import ctypes
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mhaq_amd import _lib, ops

SHAPES = [(250, 64, 56, 56)] * 5 + [(250, 128, 28, 28)] * 4 + [(250, 256, 14, 14)] * 4 + [(250, 512, 7, 7)] * 3


def ev_time(fn, reps, warm=3):
    for _ in range(warm):
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
    dev = torch.device("cuda:0")
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    out = {"per_size": {}}
    gen = torch.Generator(device=dev).manual_seed(0)
    # ---- one tensor per quantizer (distinct memory, like a step)
    xs = [torch.randn(s, device=dev, generator=gen) * 2 for s in SHAPES]
    gs = [torch.randn(s, device=dev, generator=gen) for s in SHAPES]
    ys = [torch.empty_like(x) for x in xs]
    gxs = [torch.empty_like(x) for x in xs]
    ls = torch.tensor([math.log2(16.0 / 1023)], device=dev)
    lq = ls + 10
    b = torch.tensor([-8.0], device=dev)
    params = [torch.empty(5, device=dev) for _ in SHAPES]
    grads = [torch.empty(3, device=dev) for _ in SHAPES]
    wss = []
    for x in xs:
        nb = L.mhaq_fq_act_bwd_workspace_bytes(x.numel())
        wss.append((torch.empty(nb, dtype=torch.uint8, device=dev), nb))

    def fwd(i):
        x = xs[i]
        rc = L.mhaq_fq_act_fwd(x.data_ptr(), ys[i].data_ptr(), x.numel(), ls.data_ptr(), lq.data_ptr(),
                               b.data_ptr(), params[i].data_ptr(), None, None, None, 0, st)
        assert rc == 0

    def bwd(i, off=1):
        x = xs[i]
        rc = L.mhaq_fq_act_bwd(x.data_ptr(), gs[i].data_ptr(), gxs[i].data_ptr(), x.numel(), params[i].data_ptr(),
                               0, None, 1234, off, None, grads[i].data_ptr(), wss[i][0].data_ptr(), wss[i][1], st)
        assert rc == 0

    for i in range(len(SHAPES)):
        fwd(i)
    torch.cuda.synchronize()
    # ---- per size: rotate over the tensors of that size (5 / 4 / 4 / 3 buffers)
    groups = {}
    for i, s in enumerate(SHAPES):
        groups.setdefault(s, []).append(i)
    for s, idx in groups.items():
        n = math.prod(s)
        reps = 30
        cnt = [0]

        def f():
            fwd(idx[cnt[0] % len(idx)]); cnt[0] += 1

        def bk():
            bwd(idx[cnt[0] % len(idx)]); cnt[0] += 1
        tf = sorted(ev_time(f, reps) for _ in range(3))[1]
        tb = sorted(ev_time(bk, reps) for _ in range(3))[1]
        out["per_size"]["x".join(map(str, s))] = {
            "elements": n, "rotated_buffers": len(idx), "fwd_us": round(tf * 1e3, 2), "bwd_us": round(tb * 1e3, 2),
            "fwd_GBps": round(8 * n / tf / 1e6, 1), "bwd_GBps": round(12 * n / tb / 1e6, 1),
            "fused_GBps": round(20 * n / (tf + tb) / 1e6, 1)}
    ntot = sum(math.prod(s) for s in SHAPES)
    ideal = sum(v["fwd_us"] + v["bwd_us"] for s, v in out["per_size"].items() for _ in groups[tuple(map(int, s.split("x")))])
    out["sum_of_per_size_us"] = round(ideal, 1)

    # ---- the 16-tensor sequence: raw C ABI
    def seq_capi():
        for i in range(len(SHAPES)):
            fwd(i)
        for i in reversed(range(len(SHAPES))):
            bwd(i)
    t = sorted(ev_time(seq_capi, 10) for _ in range(3))[1]
    out["set_capi_ms"] = round(t, 4)
    out["set_capi_GBps"] = round(20 * ntot / t / 1e6, 1)

    # ---- through the autograd ops
    lsp = ls.clone().requires_grad_(True)
    lqp = lq.clone().requires_grad_(True)
    bp = b.clone().requires_grad_(True)

    def seq_autograd():
        outs = []
        for x in xs:
            xi = x.detach().requires_grad_(True)
            outs.append(ops.fake_quant_act_layer(xi, lsp, lqp, bp, "STE")[0])
        torch.autograd.backward(outs, gs)
    t = sorted(ev_time(seq_autograd, 10) for _ in range(3))[1]
    out["set_autograd_ms"] = round(t, 4)
    out["set_autograd_GBps"] = round(20 * ntot / t / 1e6, 1)

    # ---- the same pass as one hipGraph
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                seq_autograd()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        lsp.grad = lqp.grad = bp.grad = None
        with torch.cuda.graph(graph, stream=side):
            seq_autograd()
        t = sorted(ev_time(graph.replay, 10) for _ in range(3))[1]
        out["set_graph_ms"] = round(t, 4)
        out["set_graph_GBps"] = round(20 * ntot / t / 1e6, 1)
    except Exception as e:  # noqa: BLE001
        out["set_graph_error"] = repr(e)[:300]
    out["elements"] = ntot
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
