#!/usr/bin/env python3
"""Generate tests/golden/big_cases.npz from the REAL reference (CPU, build container only) -- TEST INFRASTRUCTURE ONLY.

The fixtures of gen_golden.py are small tensors (<= 9216 elements): they reach the small instantiations of the kernels.
The BIG form of the streaming backward (>= 20 Mi elements), the non-temporal forward (> 16 Mi), the headline tensor
[250,64,56,56] itself and the per-channel kernels' streaming policy (>= 32 MB) were pinned through the eager oracle only.
Here the reference's own modules (imported unchanged, like gen_golden.py) run on FULL-SIZE seeded inputs, and what is
committed is not the 80-200 MB tensors but (a) the recipe that regenerates the inputs bit for bit anywhere (numpy's
default_rng for x / g / w / G, torch's CPU generator for the reference's randint_like draw), (b) three 64-bit checksums of the
bit patterns of every elementwise output (sum, xor, position-weighted sum: tests/golden_util.py::bits_checksum), (c) the
reduced gradients as values with their sum|terms| yardsticks, and (d) a 256-element window of each output for diagnosis.

Usage:  python oracle/gen_golden_big.py  [--out tests/golden]        (~2 min, ~6 GB of host memory)
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import import_reference, npf  # noqa: E402
from tests.golden_util import big_inputs, bits_checksum  # noqa: E402


def act_big(R, name, n, shape, log_s, log_q, b, method, seed):
    x, g = big_inputs(seed, n, 2.0)
    x, g = torch.from_numpy(x).reshape(shape), torch.from_numpy(g).reshape(shape)
    m = R.NoisyAct(signed=True, qnmethod=R.QNMethod[method])
    with torch.no_grad():
        m.log_act_s.fill_(log_s)
        m.log_act_q.fill_(log_q)
        m.act_b.fill_(b)
    m.train()
    xr = x.clone().requires_grad_(True)
    y = m(xr)
    torch.manual_seed(seed)                       # the reference draws randint_like(v, 2) from torch's CPU generator here
    y.backward(g)
    s, qr = 2.0 ** np.float32(log_s), 2.0 ** np.float32(log_q)
    with torch.no_grad():                         # yardsticks: sum of the magnitudes the reduced gradients sum over
        v = (torch.clamp(x, b, float(np.float32(b) + np.float32(qr) - np.float32(s))) - b) / float(s)
        abs_g = float(g.abs().double().sum()) * 2
        abs_s = float((g * torch.round(v)).abs().double().sum()) + float((g * v).abs().double().sum()) + abs_g * 0.3
    mid = n // 2
    out = dict(n=np.int64(n), shape=np.array(shape, dtype=np.int64), seed=np.int64(seed), scale=np.float32(2.0),
               log_act_s=np.float32(log_s), log_act_q=np.float32(log_q), act_b=np.float32(b),
               method=np.int8(R.QNMethod[method].value),
               y_sum=bits_checksum(npf(y)), gx_sum=bits_checksum(npf(xr.grad) + np.float32(0.0)),
               y_win=npf(y).reshape(-1)[mid:mid + 256], gx_win=npf(xr.grad).reshape(-1)[mid:mid + 256],
               g_log_act_s=npf(m.log_act_s.grad), g_log_act_q=npf(m.log_act_q.grad), g_act_b=npf(m.act_b.grad),
               abs_s=np.float64(abs_s), abs_g=np.float64(abs_g))
    return {f"{name}__{k}": v for k, v in out.items()}


def weight_big(R, name, co, ci, method, seed):
    n = co * ci * 9
    w, G = big_inputs(seed, n, 0.05)
    w, G = torch.from_numpy(w).reshape(co, ci, 3, 3), torch.from_numpy(G).reshape(co, ci, 3, 3)
    m = R.NoisyConv2d(ci, co, 3, bias=False, qscheme=R.QScheme.PER_CHANNEL, qnmethod=R.QNMethod[method])
    span = w.amax((1, 2, 3)) - w.amin((1, 2, 3))
    log_s = torch.round(torch.log2(span / 15.0))            # integer log-scales: the device's exp2 gives the host's bits
    with torch.no_grad():
        m.weight.copy_(w)
        m.log_wght_s.copy_(log_s.view_as(m.log_wght_s))
    m.train()
    m._conv_forward = lambda inp, weight, b: weight
    wq = m(torch.zeros(1, ci, 8, 8))
    torch.manual_seed(seed)
    wq.backward(G)
    gw = npf(m.weight.grad)
    w2 = npf(w).reshape(co, -1)
    off = (w2 != w2.min(axis=1, keepdims=True)).reshape(-1)          # gW off the row minima is elementwise: exact
    s = torch.exp2(log_s).reshape(co, 1)
    with torch.no_grad():
        wf, Gf = w.reshape(co, -1), G.reshape(co, -1)
        v = (wf - wf.amin(1, keepdim=True)) / s
        abs_s = ((Gf * torch.round(v)).abs().double().sum(1) + (Gf * v).abs().double().sum(1)
                 + Gf.abs().double().sum(1) * 0.6) * s.reshape(-1).double() * np.log(2.0) * 2
    out = dict(co=np.int64(co), ci=np.int64(ci), seed=np.int64(seed), scale=np.float32(0.05),
               method=np.int8(R.QNMethod[method].value), log_wght_s=npf(log_s),
               wq_sum=bits_checksum(npf(wq)), zp=npf(m.Q.zero_point).reshape(-1),
               gw_off_sum=bits_checksum(np.where(off, gw.reshape(-1), np.float32(0.0)) + np.float32(0.0)),
               gw_win=gw.reshape(-1)[n // 2:n // 2 + 256], g_log_wght_s=npf(m.log_wght_s.grad).reshape(-1),
               abs_s=abs_s.numpy())
    return {f"{name}__{k}": v for k, v in out.items()}


def weight_pt_big(R, name, co, ci, method, seed):
    """A PER_TENSOR layer beyond one workgroup (> 64 K weights: the streaming per-tensor layer path, mhaq_fq_wlayer_ptl_*)."""
    n = co * ci * 9
    w, G = big_inputs(seed, n, 0.05)
    w, G = torch.from_numpy(w).reshape(co, ci, 3, 3), torch.from_numpy(G).reshape(co, ci, 3, 3)
    m = R.NoisyConv2d(ci, co, 3, bias=False, qscheme=R.QScheme.PER_TENSOR, qnmethod=R.QNMethod[method])
    log_s = float(torch.round(torch.log2((w.max() - w.min()) / 15.0)))
    with torch.no_grad():
        m.weight.copy_(w)
        m.log_wght_s.fill_(log_s)
    m.train()
    m._conv_forward = lambda inp, weight, b: weight
    wq = m(torch.zeros(1, ci, 8, 8))
    torch.manual_seed(seed)
    wq.backward(G)
    gw = npf(m.weight.grad).reshape(-1)
    off = npf(w).reshape(-1) != float(w.min())
    s = 2.0 ** log_s
    with torch.no_grad():
        v = (w - w.min()) / s
        abs_s = (float((G * torch.round(v)).abs().double().sum()) + float((G * v).abs().double().sum())
                 + float(G.abs().double().sum()) * 0.6) * s * np.log(2.0) * 2
    out = dict(co=np.int64(co), ci=np.int64(ci), seed=np.int64(seed), scale=np.float32(0.05),
               method=np.int8(R.QNMethod[method].value), log_wght_s=np.float32(log_s),
               wq_sum=bits_checksum(npf(wq)), zp=npf(m.Q.zero_point).reshape(-1),
               gw_off_sum=bits_checksum(np.where(off, gw, np.float32(0.0)) + np.float32(0.0)),
               gw_win=gw[n // 2:n // 2 + 256], g_log_wght_s=npf(m.log_wght_s.grad).reshape(-1), abs_s=np.float64(abs_s))
    return {f"{name}__{k}": v for k, v in out.items()}


def weight_aewgs_big(R, name, co, ci, seed):
    """Per-channel AEWGS at full row length: the estimator's gradient is elementwise GIVEN its three group means, so the
    means are recorded too (recomputed here with the reference's own ops -- gdnsq.py:117-124, reduce_to_shape -- in the same
    process, on the same tensors: the bits QNAEWGS.backward used, which the consistency check below proves) and the test hands
    them to the kernels through their `stats` argument, the data-parallel path."""
    n = co * ci * 9
    w, G = big_inputs(seed, n, 0.05)
    w, G = torch.from_numpy(w).reshape(co, ci, 3, 3), torch.from_numpy(G).reshape(co, ci, 3, 3)
    m = R.NoisyConv2d(ci, co, 3, bias=False, qscheme=R.QScheme.PER_CHANNEL, qnmethod=R.QNMethod.AEWGS)
    span = w.amax((1, 2, 3)) - w.amin((1, 2, 3))
    log_s = torch.round(torch.log2(span / 15.0))
    with torch.no_grad():
        m.weight.copy_(w)
        m.log_wght_s.copy_(log_s.view_as(m.log_wght_s))
    m.train()
    m._conv_forward = lambda inp, weight, b: weight
    wq = m(torch.zeros(1, ci, 8, 8))
    torch.manual_seed(seed)
    wq.backward(G)
    gw = npf(m.weight.grad)
    with torch.no_grad():
        s = torch.exp2(log_s).reshape(co, 1, 1, 1)
        zp = w.amin((1, 2, 3), keepdim=True)
        v = (w - zp) / s
        gq = G * s
        e = torch.round(v) - v
        num_full = gq.sign() * e
        stats = torch.stack([torch.mean(t, dim=(1, 2, 3), keepdim=True) for t in (num_full, e.square(), e)])
        # consistency: the reference's elementwise arithmetic on these statistics gives the reference's gW off the extremes
        den = (stats[1] - stats[2].square()).clamp_min(1e-3)
        g_scale = ((stats[0] / den) * num_full).clamp_max(0.99)
        gv = gq + (-gq * g_scale)
        mine = npf(gv / s)
    w2 = npf(w).reshape(co, -1)
    off = ((w2 != w2.min(axis=1, keepdims=True)) & (w2 != w2.max(axis=1, keepdims=True))).reshape(-1)
    assert np.array_equal(mine.reshape(-1)[off], gw.reshape(-1)[off]), "recorded statistics are not the reference's"
    out = dict(co=np.int64(co), ci=np.int64(ci), seed=np.int64(seed), scale=np.float32(0.05),
               method=np.int8(R.QNMethod.AEWGS.value), log_wght_s=npf(log_s), stats=npf(stats).reshape(3, co),
               wq_sum=bits_checksum(npf(wq)), zp=npf(m.Q.zero_point).reshape(-1),
               gw_off_sum=bits_checksum(np.where(off, gw.reshape(-1), np.float32(0.0)) + np.float32(0.0)),
               gw_win=gw.reshape(-1)[n // 2:n // 2 + 256])
    return {f"{name}__{k}": v for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    args = ap.parse_args()
    torch.set_num_threads(8)
    R = import_reference()
    data = {}
    big = (20 << 20) + 4099                                  # just above the BIG threshold of the streaming backward, odd tail
    data.update(act_big(R, "act_big_ste", big, (big,), -3.0, 2.0, -2.0, "STE", 501))
    data.update(act_big(R, "act_big_lsq", big, (big,), -3.0, 2.0, -2.0, "LSQ", 502))
    data.update(act_big(R, "act_headline_ste", 250 * 64 * 56 * 56, (250, 64, 56, 56), -3.0, 2.0, -2.0, "STE", 503))
    data.update(weight_big(R, "w_2048x4608_lsq", 2048, 512, "LSQ", 511))       # 37.7 MB: the streaming policy of fq_pc.hip
    data.update(weight_big(R, "w_2048x4608_ste", 2048, 512, "STE", 512))
    # configs[4]'s stress shape: one 69.1 M-element RFDN activation, LSQ at 2 bits
    data.update(act_big(R, "act_rfdn_stress_lsq", 24 * 50 * 180 * 320, (24, 50, 180, 320), -1.0, 1.0, -1.0, "LSQ", 504))
    data.update(weight_pt_big(R, "wpt_512x4608_ste", 512, 512, "STE", 521))     # PER_TENSOR, 2.36 M weights: the streaming layer path
    data.update(weight_pt_big(R, "wpt_512x4608_lsq", 512, 512, "LSQ", 522))
    data.update(weight_aewgs_big(R, "waewgs_2048x4608", 2048, 512, 531))         # AEWGS at full row length, statistics recorded
    os.makedirs(args.out, exist_ok=True)
    path = os.path.join(args.out, "big_cases.npz")
    np.savez_compressed(path, **data)
    names = sorted({k.split("__")[0] for k in data})
    print(f"big_cases.npz: {len(names)} cases, {os.path.getsize(path) / 1024:.1f} KiB: {', '.join(names)}")


if __name__ == "__main__":
    main()
