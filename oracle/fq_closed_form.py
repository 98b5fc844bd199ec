"""Closed forms of the fake-quant backward, fp32 elementwise terms with
fp64 sums  --  TEST INFRASTRUCTURE ONLY (see oracle/fq_eager.py for the import rule).

fq_eager.py restates the reference op-for-op and lets autograd derive the gradients (that is
what is pinned against the golden vectors).  This file states the same gradients in closed
form, derived from that autograd graph (gdnsq.py:189-229 + QN*.backward), and is itself
checked against fq_eager.py in tests/test_oracle_golden.py.  Its two uses:
  * a tolerance yardstick: `abs_*` = sum of |terms| of each reduced gradient, so tests can
    state "|kernel - reference| <= 1e-6 * sum|terms|" (fp32 reductions of up to 5e7 terms
    cannot agree to 1e-6 of a cancelling sum);
  * an accurate (fp64-accumulated) value of each reduction.
"""
from __future__ import annotations

import torch

INV_SQRT3_F32 = torch.tensor(3.0 ** -0.5, dtype=torch.float32)


def _f64sum(t, dims=None):
    t = t.to(torch.float64)
    return t.sum() if dims is None else t.sum(dim=dims, keepdim=True)


def _noise_grad_v(gq, e, method, delta=None):
    if method in ("STE", "LSQ"):
        return gq * 0
    if method == "EWGS":
        return -torch.abs(gq) * e * 1e-2
    if method == "AEWGS":
        return -gq * (1.0 * delta * (gq.sign() * e)).clamp_max(0.99)
    raise AttributeError(f"Unknown method {method}!")


def per_tensor(x, g, r, s, zp, lo, hi, method="STE", delta=None):
    """x, g fp32 tensors; r +-0.5 tensor (ignored for LSQ); s, zp, lo, hi fp32 0-dim/1-elem tensors.
    `delta` (AEWGS only) broadcastable per-element num/den."""
    s, zp, lo, hi = (torch.as_tensor(t, dtype=torch.float32).reshape(()) for t in (s, zp, lo, hi))
    v0 = torch.clamp(x, min=lo, max=hi)
    v1 = v0 - zp
    v = v1 / s
    n = torch.round(v) - v
    q = v + n
    y = q * s + zp
    gq = g * s
    gv = gq + _noise_grad_v(gq, n, method, delta)
    g1 = gv / s
    inside = (x >= lo) & (x <= hi)
    gx = torch.where(inside, g1, torch.zeros_like(g1))
    noise = gq * n if method == "LSQ" else (INV_SQRT3_F32 * gq) * r
    t_s = (g * q + (-gv) * (v / s)) + noise
    lo_lt_hi, hi_lt_lo = bool(lo < hi), bool(hi < lo)
    m_lo = (x < lo) & lo_lt_hi
    m_hi = (x > hi) | hi_lt_lo
    zero = torch.zeros_like(g1)
    return dict(
        y=y, q=q, gx=gx,
        g_s=_f64sum(t_s), g_zp=_f64sum(g - g1),
        g_lo=_f64sum(torch.where(m_lo, g1, zero)), g_hi=_f64sum(torch.where(m_hi, g1, zero)),
        count_zp=int((x == zp).sum()),
        # yardsticks: the reference reduces g*q, gv*(v/s) and the noise term separately
        abs_s=_f64sum((g * q).abs()) + _f64sum((gv * (v / s)).abs()) + _f64sum(noise.abs()),
        abs_g=_f64sum(g.abs()) + _f64sum(g1.abs()),
    )


def per_channel(w, G, r, s, method="STE", stats=None):
    """w, G: [co, ...]; s: [co] (or [co,1,..]).  stats: optional (num, e2, me) each [co] (AEWGS,
    e.g. after a cross-rank average); None = this tensor's own statistics."""
    co = w.shape[0]
    shp = [co] + [1] * (w.dim() - 1)
    dims = tuple(range(1, w.dim()))
    s = s.reshape(shp).to(torch.float32)
    zp = w.amin(dims, keepdim=True)
    v1 = w - zp
    v = v1 / s
    n = torch.round(v) - v
    q = v + n
    wq = q * s + zp
    gq = G * s
    delta = None
    if method == "AEWGS":
        if stats is None:
            num = (gq.sign() * n).mean(dims, keepdim=True)
            e2 = n.square().mean(dims, keepdim=True)
            me = n.mean(dims, keepdim=True)
        else:
            num, e2, me = (t.reshape(shp) for t in stats)
        delta = num / (e2 - me.square()).clamp_min(1e-3)
    gv = gq + _noise_grad_v(gq, n, method, delta)
    gvs = gv / s
    noise = gq * n if method == "LSQ" else (INV_SQRT3_F32 * gq) * r
    t_s = (G * q + (-gv) * (v / s)) + noise
    g_zp = _f64sum(G - gvs, dims)
    mask = (w == zp)
    cnt = mask.sum(dims, keepdim=True)
    gw = gvs + torch.where(mask, (g_zp.to(torch.float32) * 1.0) / cnt, torch.zeros_like(gvs))
    return dict(
        wq=wq, q=q, zp=zp, gw=gw, g_s=_f64sum(t_s, dims).reshape(co), g_zp=g_zp.reshape(co),
        count_zp=cnt.reshape(co),
        abs_s=(_f64sum((G * q).abs(), dims) + _f64sum((gv * (v / s)).abs(), dims)
               + _f64sum(noise.abs(), dims)).reshape(co),
        abs_g=(_f64sum(G.abs(), dims) + _f64sum(gvs.abs(), dims)).reshape(co),
    )
