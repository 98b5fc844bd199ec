#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (CPU, build container only).

TEST INFRASTRUCTURE ONLY.  Imports the reference's hot-path modules unchanged
from /root/reference (SURVEY.md Appendix B recipe: only gdnsq.py, the three
layer wrappers, model_helper.py, gdnsq_loss.py and calib/minmaxobserver.py are loaded; the package
__init__ that needs Lightning is bypassed), runs them on seeded inputs and
stores inputs + outputs as small fp32 arrays.  Nothing of the reference's
source travels: the fixtures are data.  /root/reference does not exist on the
GPU box, so this script only ever runs here; its outputs are committed.

Capturing the random +-0.5 tensor `r` of the STE/AEWGS scale gradient
(gdnsq.py:54,144): torch.manual_seed(k) immediately before backward(), then
re-seed and re-draw torch.randint_like(v, 2) - 0.5 (one draw per quantizer).

Usage:  python oracle/gen_golden.py  [--out tests/golden]
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


def import_reference():
    sys.dont_write_bytecode = True            # /root/reference is read-only for this build: no __pycache__ there
    sys.path.insert(0, REF)
    import src  # noqa: F401  (empty package)
    pkg = types.ModuleType("src.quantization")
    pkg.__path__ = [os.path.join(REF, "src/quantization")]
    sys.modules["src.quantization"] = pkg
    from src.quantization.gdnsq.gdnsq_utils import QNMethod
    from src.quantization.gdnsq.layers.gdnsq_act import NoisyAct
    from src.quantization.gdnsq.layers.gdnsq_conv2d import NoisyConv2d
    from src.quantization.gdnsq.layers.gdnsq_linear import NoisyLinear
    from src.quantization.gdnsq.utils.model_helper import ModelHelper
    from src.quantization.gdnsq.gdnsq_loss import PotentialLoss, PotentialLossNoPred
    from src.aux.types import QScheme
    # the calibration module logs through src.loggers.default_logger, which needs Lightning (absent here): give it
    # a plain logging.Logger under that name (no arithmetic involved) and import the module unchanged
    import logging
    lg = types.ModuleType("src.loggers")
    lg.__path__ = []
    dl = types.ModuleType("src.loggers.default_logger")
    dl.logger = logging.getLogger("mhaq.reference")
    lg.default_logger = dl
    sys.modules["src.loggers"], sys.modules["src.loggers.default_logger"] = lg, dl
    from src.quantization.gdnsq.calib import minmaxobserver as calib
    from src.quantization.gdnsq.utils import model_stats
    return types.SimpleNamespace(**locals())


def npf(t):
    return t.detach().to(torch.float32).cpu().numpy().copy()


def draw_r(seed, like):
    torch.manual_seed(seed)
    return torch.randint_like(like, 2).sub_(0.5)


# --------------------------------------------------------------------------- K1
def act_case(R, name, x, g, log_s, log_q, b, signed, seed, method="STE"):
    """One NoisyAct fixture; `method` = NoisyAct(qnmethod=...) (gdnsq_act.py:17).  AEWGS on a [1]-shaped scale
    reduces its statistics over dim 0 only (gdnsq.py:150-152)."""
    out = {}
    m = R.NoisyAct(signed=signed, qnmethod=R.QNMethod[method])
    with torch.no_grad():
        m.log_act_s.fill_(log_s)
        m.log_act_q.fill_(log_q)
        m.act_b.fill_(b)
    m.train()
    xr = x.clone().requires_grad_(True)
    y = m(xr)
    torch.manual_seed(seed)
    y.backward(g)
    r = draw_r(seed, x)
    out.update(x=npf(x), g=npf(g), r=(npf(r) * 2).astype(np.int8),
               log_act_s=np.float32(log_s), log_act_q=np.float32(log_q), act_b=np.float32(b),
               signed=np.int8(signed), method=np.int8(R.QNMethod[method].value), y=npf(y), gx=npf(xr.grad),
               g_log_act_s=npf(m.log_act_s.grad), g_log_act_q=npf(m.log_act_q.grad),
               g_act_b=npf(m.act_b.grad) if m.act_b.grad is not None else np.zeros(1, np.float32))
    # eval mode: q statistics + bit width (gdnsq_act.py:51-54); may raise the integrity asserts
    m.eval()
    with torch.no_grad():
        try:
            ye = m(x)
            out.update(y_eval=npf(ye), bw=npf(m.bw), eval_raises=np.int8(0))
        except AssertionError:
            out.update(eval_raises=np.int8(1))
    return {f"{name}__{k}": v for k, v in out.items()}


def gen_act(R):
    cases = {}
    gen = torch.Generator().manual_seed(1234)
    shape = (2, 8, 6, 6)

    def rn(*s):
        return torch.randn(*s, generator=gen)

    # 1. everything inside the clamp range, power-of-two scale
    x = rn(*shape) * 2
    cases.update(act_case(R, "in_range", x, rn(*shape), -6.0, 4.0, -8.0, True, 11))
    # 2. heavy clipping on both sides
    x = rn(*shape) * 4
    cases.update(act_case(R, "clip_both", x, rn(*shape), -3.0, 1.0, -1.0, True, 12))
    # 3. values exactly on lo / hi (inclusive bounds keep the gradient on x)
    log_s, log_q, b = -2.0, 2.0, -2.0
    lo, hi = b, b + 2.0 ** log_q - 2.0 ** log_s
    x = rn(*shape) * 3
    xf = x.flatten()
    xf[0:40] = lo
    xf[40:80] = hi
    xf[80:90] = math.nextafter(lo, -math.inf)
    xf[90:100] = math.nextafter(hi, math.inf)
    cases.update(act_case(R, "on_bounds", xf.view(shape), rn(*shape), log_s, log_q, b, True, 13))
    # 4. v exactly at .5 ties -> round-half-even
    log_s, log_q, b = -1.0, 4.0, -8.0
    k = torch.randint(0, 31, shape, generator=gen).float()
    x = b + (k + 0.5) * 2.0 ** log_s
    x.flatten()[::7] += 0.123
    cases.update(act_case(R, "ties", x, rn(*shape), log_s, log_q, b, True, 14))
    # 5. non-power-of-two scale: post-calibration state (calib/minmaxobserver.py:56-61)
    x = rn(*shape) * 2
    rng_ = (x.max() - x.min()).item()
    log_s = math.log2(rng_ / 1023.0)
    cases.update(act_case(R, "calibrated10", x, rn(*shape), log_s, log_s + 10, x.min().item(), True, 15))
    # 5b. 4-bit state with clipping and non-pow2 scale
    log_s = math.log2(rng_ * 0.6 / 15.0)
    cases.update(act_case(R, "w4_nonpow2", x, rn(*shape), log_s, log_s + 4, x.min().item() * 0.6, True, 16))
    # 6. unsigned (post-ReLU) quantizer: act_b frozen at 0
    x = torch.relu(rn(*shape) * 2)
    cases.update(act_case(R, "unsigned", x, rn(*shape), -4.0, 2.5, 0.0, False, 17))
    # 7. inverted range qr < s  (lo > hi): everything collapses onto hi
    x = rn(*shape)
    cases.update(act_case(R, "inverted", x, rn(*shape), 1.0, 0.0, -0.5, True, 18))
    # 8. larger ragged tensor (numel not a multiple of 4) with default init (-10, 10)
    shape2 = (3, 5, 7, 9)
    x = rn(*shape2) * 100
    cases.update(act_case(R, "default_init_ragged", x, rn(*shape2), -10.0, 10.0, -512.0, True, 19))
    # 9. a realistic larger tensor
    shape3 = (4, 16, 12, 12)
    x = rn(*shape3) * 2
    rng_ = (x.max() - x.min()).item()
    log_s = math.log2(rng_ * 0.8 / 15.0)
    cases.update(act_case(R, "big_w4", x, rn(*shape3), log_s, log_s + 4, x.min().item() * 0.8, True, 20))
    # 10-15. the other estimators NoisyAct(qnmethod=...) accepts (gdnsq_act.py:17): LSQ (deterministic scale
    # gradient) and AEWGS (statistics over dim 0 for the [1]-shaped scale), drawn after the cases above so
    # that those keep their values
    x = rn(*shape) * 4
    cases.update(act_case(R, "lsq_clip_both_pow2", x, rn(*shape), -3.0, 1.0, -1.0, True, 21, "LSQ"))
    x = rn(*shape3) * 2
    rng_ = (x.max() - x.min()).item()
    log_s = math.log2(rng_ * 0.8 / 15.0)
    cases.update(act_case(R, "lsq_big_w4", x, rn(*shape3), log_s, log_s + 4, x.min().item() * 0.8, True, 22, "LSQ"))
    x = torch.relu(rn(*shape) * 2)
    cases.update(act_case(R, "lsq_unsigned", x, rn(*shape), -4.0, 2.5, 0.0, False, 23, "LSQ"))
    x = rn(*shape3) * 2
    cases.update(act_case(R, "aewgs_big_w4", x, rn(*shape3), log_s, log_s + 4, x.min().item() * 0.8, True, 24, "AEWGS"))
    x = rn(*shape) * 4
    cases.update(act_case(R, "aewgs_clip_both_pow2", x, rn(*shape), -3.0, 1.0, -1.0, True, 25, "AEWGS"))
    x = torch.relu(rn(*shape) * 2)
    cases.update(act_case(R, "aewgs_unsigned", x, rn(*shape), -4.0, 2.5, 0.0, False, 26, "AEWGS"))
    # 16. AEWGS with a batch of ONE: the statistics of a [1]-shaped scale run over dim 0 only, so every position is
    # its own group, e2 - me^2 == 0, the denominator clamps to 1e-3 and g_scale saturates at 0.99 wherever
    # sign(g) * e > 0 (gdnsq.py:131-141): both clamps active in the activation kernel
    x = rn(1, 8, 6, 6) * 2
    cases.update(act_case(R, "aewgs_batch_of_one", x, rn(1, 8, 6, 6), -3.0, 2.0, -2.0, True, 27, "AEWGS"))
    return cases


# --------------------------------------------------------------------------- K2
def weight_case(R, name, w, G, log_s, per_channel, method, seed, linear=False,
                bias=None, Gb=None):
    out = {}
    qs = R.QScheme.PER_CHANNEL if per_channel else R.QScheme.PER_TENSOR
    qn = R.QNMethod[method]
    if linear:
        m = R.NoisyLinear(w.shape[1], w.shape[0], bias=False, qscheme=qs, qnmethod=qn)
    else:
        m = R.NoisyConv2d(w.shape[1], w.shape[0], tuple(w.shape[2:]), bias=bias is not None,
                          qscheme=qs, qnmethod=qn, quant_bias=bias is not None)
    with torch.no_grad():
        m.weight.copy_(w)
        if bias is not None:
            m.bias.copy_(bias)
        if torch.is_tensor(log_s):
            m.log_wght_s.copy_(log_s.view_as(m.log_wght_s))
        else:
            m.log_wght_s.fill_(log_s)
    m.train()
    captured = {}
    if linear:
        import torch.nn.functional as F
        orig = F.linear

        def fake_linear(inp, weight, b=None):
            captured["w"] = weight
            return weight
        F.linear = fake_linear
        try:
            wq = m(torch.zeros(1, w.shape[1]))
        finally:
            F.linear = orig
    else:
        def conv_forward(inp, weight, b):
            captured["w"], captured["b"] = weight, b
            return weight
        m._conv_forward = conv_forward
        wq = m(torch.zeros(1, w.shape[1], 8, 8))
    torch.manual_seed(seed)
    if bias is not None:
        torch.autograd.backward([captured["w"], captured["b"]], [G, Gb])
    else:
        wq.backward(G)
    torch.manual_seed(seed)
    r = torch.randint_like(w, 2).sub_(0.5)
    out.update(w=npf(w), G=npf(G), r=(npf(r) * 2).astype(np.int8),
               log_wght_s=npf(m.log_wght_s), per_channel=np.int8(per_channel),
               method=np.int8(qn.value), wq=npf(wq), zp=npf(m.Q.zero_point),
               gw=npf(m.weight.grad), g_log_wght_s=npf(m.log_wght_s.grad))
    if bias is not None:
        rb = torch.randint_like(bias, 2).sub_(0.5)  # second draw: the bias quantizer's backward
        out.update(bias=npf(bias), Gb=npf(Gb), rb=(npf(rb) * 2).astype(np.int8),
                   bq=npf(captured["b"]), gbias=npf(m.bias.grad))
    return {f"{name}__{k}": v for k, v in out.items()}


def gen_weight(R):
    cases = {}
    gen = torch.Generator().manual_seed(4321)

    def rn(*s):
        return torch.randn(*s, generator=gen)

    shape = (8, 4, 3, 3)
    fan_in = 4 * 9
    for method in ("STE", "LSQ", "AEWGS"):
        for pc in (False, True):
            w = rn(*shape) * math.sqrt(2.0 / fan_in)
            G = rn(*shape)
            if pc:
                mx, mn = w.amax((1, 2, 3)), w.amin((1, 2, 3))
                log_s = torch.log2((mx - mn) / 15.0) + 0.1 * rn(shape[0])
            else:
                log_s = math.log2((w.max() - w.min()).item() / 15.0) + 0.05
            tag = f"{method.lower()}_{'pc' if pc else 'pt'}"
            cases.update(weight_case(R, tag, w, G, log_s, pc, method, 100 + len(cases)))
    # default init log_s=-12 (fine grid, |q| large) per-channel STE
    w = rn(*shape) * 0.2
    cases.update(weight_case(R, "ste_pc_init12", w, rn(*shape), -12.0, True, "STE", 201))
    # tied minima: channel 0 has 3 equal minima, channel 1 has 2; per-tensor has 2 global
    w = rn(*shape) * 0.3
    w[0].flatten()[[1, 7, 20]] = w[0].min() - 0.05
    w[1].flatten()[[0, 35]] = w[1].min() - 0.01
    for method in ("STE", "LSQ", "AEWGS"):
        cases.update(weight_case(R, f"tied_{method.lower()}_pc", w, rn(*shape), -3.0, True, method, 210))
    wt = w.clone()
    wt.flatten()[[5, 100]] = wt.min() - 0.1
    cases.update(weight_case(R, "tied_aewgs_pt", wt, rn(*shape), -3.0, False, "AEWGS", 211))
    cases.update(weight_case(R, "tied_lsq_pt", wt, rn(*shape), -3.0, False, "LSQ", 212))
    # AEWGS clamps: channel 0 -> e constant (den clamp 1e-3, g_scale clamp .99);
    # channel 1 -> exactly on the grid (e == 0)
    s = 2.0 ** -3
    w = rn(*shape) * 0.3
    k = torch.randint(0, 12, (36,), generator=gen).float()
    k[0] = 0
    w[0] = (k * s + 0.3 * s).view(4, 3, 3)
    w[0].flatten()[0] = 0.0
    w[1] = (k * s).view(4, 3, 3) - 0.5
    G = rn(*shape)
    G[0] = G[0].abs() + 0.1
    cases.update(weight_case(R, "aewgs_clamps_pc", w, G, -3.0, True, "AEWGS", 220))
    # a wider layer shape (row = 16*9 = 144, not a multiple of 64) per-channel, all methods
    shape2 = (32, 16, 3, 3)
    w = rn(*shape2) * math.sqrt(2.0 / 144)
    mx, mn = w.amax((1, 2, 3)), w.amin((1, 2, 3))
    log_s = torch.maximum(torch.full((32,), -12.0), torch.log2((mx - mn) / 1023.0))
    for method in ("STE", "LSQ", "AEWGS"):
        cases.update(weight_case(R, f"wide_{method.lower()}_pc", w, rn(*shape2), log_s, True, method, 230))
    cases.update(weight_case(R, "wide_aewgs_pt", w, rn(*shape2), -6.3, False, "AEWGS", 231))
    # quant_bias=True (per-channel only): bias reuses s.ravel() / zp.ravel()
    w = rn(*shape) * 0.3
    b = rn(shape[0]) * 0.1 + 0.2
    cases.update(weight_case(R, "qbias_lsq_pc", w, rn(*shape), -4.0, True, "LSQ", 240,
                             bias=b, Gb=rn(shape[0])))
    cases.update(weight_case(R, "qbias_ste_pc", w, rn(*shape), -4.0, True, "STE", 241,
                             bias=b, Gb=rn(shape[0])))
    # AEWGS bias: scale.shape == bias.shape, so reduce_to_shape averages over the whole vector
    cases.update(weight_case(R, "qbias_aewgs_pc", w, rn(*shape), -4.0, True, "AEWGS", 242,
                             bias=b, Gb=rn(shape[0])))
    # Linear, per-tensor (per-channel Linear raises IndexError in the reference)
    wl = rn(10, 64) * 0.2
    for method in ("STE", "LSQ", "AEWGS"):
        cases.update(weight_case(R, f"linear_{method.lower()}_pt", wl, rn(10, 64), -5.0, False,
                                 method, 250, linear=True))
    return cases


# ------------------------------------------------------------------ model level
def gen_model(R):
    """get_model_values + PotentialLoss(NoPred) on a 2-layer toy net (model_helper.py:13-76,
    gdnsq_loss.py:32-86,114-168)."""
    cases = {}
    gen = torch.Generator().manual_seed(777)
    for pc in (False, True):
        qs = R.QScheme.PER_CHANNEL if pc else R.QScheme.PER_TENSOR
        tag = "pc" if pc else "pt"
        torch.manual_seed(5)
        net = torch.nn.Sequential(
            R.NoisyAct(signed=True), R.NoisyConv2d(3, 6, 3, padding=1, qscheme=qs, qnmethod=R.QNMethod.LSQ),
            torch.nn.ReLU(),
            R.NoisyAct(signed=False), R.NoisyConv2d(6, 4, 3, padding=1, qscheme=qs, qnmethod=R.QNMethod.LSQ),
        )
        with torch.no_grad():
            for i, m in enumerate(net):
                if isinstance(m, R.NoisyAct):
                    m.log_act_s.fill_(-4.0 - i)
                    m.log_act_q.fill_(1.5 + i)
                    if m.signed:
                        m.act_b.fill_(-2.0)
                if isinstance(m, R.NoisyConv2d):
                    m.log_wght_s.fill_(-6.0)
                    m.log_wght_s.add_(0.3 * torch.randn(m.log_wght_s.shape, generator=gen))
        las, laq, lws, lwq = R.ModelHelper.get_model_values(net, qs)
        base = torch.tensor(1.7, requires_grad=True)
        for nopred in (True, False):
            net.zero_grad()
            if nopred:
                L = R.PotentialLossNoPred(criterion=None, p=1, a=4, w=4)
                L.t, L.loss_sum, L.cnt = 0.35, torch.tensor(3.3), 3
                las, laq, lws, lwq = R.ModelHelper.get_model_values(net, qs)
                ploss = L((base * 1.0, las, laq, lws, lwq))
            else:
                L = R.PotentialLoss(criterion=torch.nn.MSELoss(), p=1, a=4, w=4)
                L.t, L.loss_sum, L.cnt = 0.35, torch.tensor(3.3), 3
                las, laq, lws, lwq = R.ModelHelper.get_model_values(net, qs)
                prd = torch.linspace(-1, 1, 12).view(3, 4).requires_grad_(True)
                tgt = torch.linspace(1, -1, 12).view(3, 4) * 0.5
                ploss = L((prd, las, laq, lws, lwq), tgt)
            ploss.backward()
            k = f"model_{tag}_{'nopred' if nopred else 'pred'}"
            convs = [m for m in net if isinstance(m, R.NoisyConv2d)]
            acts = [m for m in net if isinstance(m, R.NoisyAct)]
            d = dict(las=npf(las), laq=npf(laq), lws=npf(lws), lwq=npf(lwq), ploss=npf(ploss),
                     t=np.float32(0.35), loss_sum=np.float32(3.3), cnt=np.int32(3),
                     a_bits=np.int32(4), w_bits=np.int32(4), per_channel=np.int8(pc),
                     base=np.float32(1.7))
            for i, c in enumerate(convs):
                d[f"w{i}"] = npf(c.weight)
                d[f"log_wght_s{i}"] = npf(c.log_wght_s)
                d[f"gw{i}"] = npf(c.weight.grad)
                d[f"g_log_wght_s{i}"] = npf(c.log_wght_s.grad)
            for i, a in enumerate(acts):
                d[f"log_act_s{i}"] = npf(a.log_act_s)
                d[f"log_act_q{i}"] = npf(a.log_act_q)
                d[f"g_log_act_s{i}"] = npf(a.log_act_s.grad)
                d[f"g_log_act_q{i}"] = npf(a.log_act_q.grad)
            cases.update({f"{k}__{kk}": v for kk, v in d.items()})
    return cases


# ------------------------------------------------------------------ calibration
def gen_calib(R):
    """MinMaxObserver._hook + apply_mean_stats_activations + apply_quantile_weights_s
    (gdnsq/calib/minmaxobserver.py:26-88) on a flat list of reference layers.  The observer's constructor allocates
    on "cuda" (:22-23) and is bypassed; its hook -- the part that records -- runs unchanged."""
    C = R.calib
    cases = {}
    gen = torch.Generator().manual_seed(2468)

    def rn(*s):
        return torch.randn(*s, generator=gen)

    for tag, abits, wbits in (("a10w10", 10, 10), ("a4w4", 4, 4)):
        acts = torch.nn.ModuleList([R.NoisyAct(signed=True), R.NoisyAct(signed=False), R.NoisyAct(signed=True),
                                    R.NoisyAct(signed=True), R.NoisyAct(signed=True), R.NoisyAct(signed=False)])
        # 2: constant input -> zero width -> "pruned"; 3: frozen quantizer -> max_bits, and (reference quirk) so are
        # the quantizers after it
        acts[3].log_act_s.requires_grad_(False)
        acts[3].log_act_q.requires_grad_(False)
        shapes = [(2, 3, 9, 9), (2, 6, 9, 9), (2, 4, 5, 5), (3, 5, 7), (2, 8, 6, 6), (2, 6, 4, 4)]
        obs = C.MinMaxObserver.__new__(C.MinMaxObserver)
        d = {}
        for i, (a, shp) in enumerate(zip(acts, shapes)):
            for b in range(3):                                  # three observed batches per quantizer
                x = rn(*shp) * (1.0 + i) + 0.3 * b
                if i in (1, 5):
                    x = torch.relu(x)
                if i == 2:
                    x = torch.full(shp, -0.75)
                d[f"act{i}_x{b}"] = npf(x)
                obs._hook(a, (x,), None)
            d[f"act{i}_grad_in"] = np.array([a.log_act_s.requires_grad, a.log_act_q.requires_grad,
                                             a.act_b.requires_grad], dtype=np.int8)
        C.apply_mean_stats_activations(acts, abits=abits)
        for i, a in enumerate(acts):
            d[f"act{i}_log_act_s"] = npf(a.log_act_s.float())
            d[f"act{i}_log_act_q"] = npf(a.log_act_q.float())
            d[f"act{i}_act_b"] = npf(a.act_b.float())
            d[f"act{i}_grad_out"] = np.array([a.log_act_s.requires_grad, a.log_act_q.requires_grad,
                                              a.act_b.requires_grad], dtype=np.int8)
        convs = torch.nn.ModuleList([
            R.NoisyConv2d(3, 6, 3, qscheme=R.QScheme.PER_CHANNEL), R.NoisyConv2d(6, 4, 3, qscheme=R.QScheme.PER_CHANNEL),
            R.NoisyConv2d(4, 5, 1, qscheme=R.QScheme.PER_CHANNEL), R.NoisyConv2d(5, 8, 3, qscheme=R.QScheme.PER_CHANNEL)])
        with torch.no_grad():
            for i, c in enumerate(convs):
                c.weight.copy_(rn(*c.weight.shape) * (0.05 + 0.1 * i))
            convs[0].weight[2] = 0.125                           # a constant channel: log2(0) = -inf keeps -12
            convs[1].log_wght_s.fill_(-3.0)                      # already coarser than the range asks for
            convs[1].weight[0] *= 30                             # ... except in channel 0
        convs[2].log_wght_s.requires_grad_(False)                # frozen -> max_bits, sticks for convs[3] too
        for i, c in enumerate(convs):
            d[f"conv{i}_w"] = npf(c.weight)
            d[f"conv{i}_log_wght_s_in"] = npf(c.log_wght_s)
            d[f"conv{i}_grad_in"] = np.int8(c.log_wght_s.requires_grad)
        C.apply_quantile_weights_s(convs, wbits=wbits)
        for i, c in enumerate(convs):
            d[f"conv{i}_log_wght_s"] = npf(c.log_wght_s)
            d[f"conv{i}_grad_out"] = np.int8(c.log_wght_s.requires_grad)
        d["abits"], d["wbits"] = np.int32(abits), np.int32(wbits)
        cases.update({f"calib_{tag}__{k}": v for k, v in d.items()})
    return cases


# ------------------------------------------------------------------ bit-width statistics
def gen_stats(R):
    """utils/model_stats.py:116-262 on a flat list of reference layers after one eval-mode forward each (the
    functions read Q.scale / Q.zero_point as the last forward left them, and NoisyAct.bw)."""
    MS = R.model_stats
    cases = {}
    gen = torch.Generator().manual_seed(1357)

    def rn(*s):
        return torch.randn(*s, generator=gen)

    for tag, qs in (("pt", R.QScheme.PER_TENSOR), ("pc", R.QScheme.PER_CHANNEL)):
        d = {}
        convs = torch.nn.ModuleList([R.NoisyConv2d(3, 6, 3, qscheme=qs), R.NoisyConv2d(6, 4, 3, qscheme=qs),
                                     R.NoisyConv2d(4, 8, 1, qscheme=qs)])
        acts = torch.nn.ModuleList([R.NoisyAct(signed=True), R.NoisyAct(signed=False), R.NoisyAct(signed=True)])
        with torch.no_grad():
            for i, c in enumerate(convs):
                c.weight.copy_(rn(*c.weight.shape) * (0.1 + 0.2 * i))
                c.log_wght_s.copy_(-6.0 + 1.7 * i + 0.4 * rn(*c.log_wght_s.shape))       # not powers of two
            convs[0].weight[1] = 0.25                                                  # a constant channel
            for i, a in enumerate(acts):
                a.log_act_s.fill_(-3.3 - 0.6 * i)
                a.log_act_q.fill_(1.2 + 0.9 * i)
                a.act_b.fill_(-1.1 if a.signed else 0.0)
        model = torch.nn.ModuleDict({"convs": convs, "acts": acts}).eval()
        xin = [rn(2, 3, 7, 7), rn(2, 6, 7, 7), rn(2, 4, 7, 7)]
        xact = [rn(2, 5, 6, 6) * 2, torch.relu(rn(2, 5, 6, 6) * 2), rn(3, 7, 5) * 3]
        with torch.no_grad():
            for c, x in zip(convs, xin):
                c(x)
            for a, x in zip(acts, xact):
                a(x)
        for i, c in enumerate(convs):
            d[f"conv{i}_w"], d[f"conv{i}_log_wght_s"] = npf(c.weight), npf(c.log_wght_s)
            d[f"conv{i}_bw_max"] = np.float64(MS.get_true_layer_bit_width(c, max=True))
            d[f"conv{i}_bw_mean"] = np.float64(MS.get_true_layer_bit_width(c, max=False))
            d[f"conv{i}_wnb"] = npf(MS.get_layer_wnb_bit_width(c.weight.detach(), c.log_wght_s.detach(), c.qscheme))
        for i, (a, x) in enumerate(zip(acts, xact)):
            d[f"act{i}_x"] = npf(x)
            d[f"act{i}_params"] = np.array([float(a.log_act_s), float(a.log_act_q), float(a.act_b)], np.float32)
            d[f"act{i}_signed"] = np.int8(a.signed)
            d[f"act{i}_bw"] = npf(a.bw)
        d["true_weights_width_max"] = np.float64(MS.get_true_weights_width(model, max=True))
        d["true_weights_width_mean"] = np.float64(MS.get_true_weights_width(model, max=False))
        d["weights_bit_width_mean"] = npf(MS.get_weights_bit_width_mean(model))
        d["activations_bit_width_mean"] = npf(MS.get_activations_bit_width_mean(model))
        d["true_activations_width_max"] = np.float64(MS.get_true_activations_width(model, max=True))
        d["true_activations_width_mean"] = np.float64(MS.get_true_activations_width(model, max=False))
        d["per_channel"] = np.int8(qs == R.QScheme.PER_CHANNEL)
        cases.update({f"stats_{tag}__{k}": v for k, v in d.items()})
    return cases


# ------------------------------------------------------------------ EWGS
class ewgs_enabled:
    """QNEWGS.backward (gdnsq.py:90-107) reads `ctx.need_input_grad` at :102 -- a misspelling of `needs_input_grad` --
    and therefore raises AttributeError whenever EWGS is used; no shipped config selects it.  Everything ELSE in that
    method is well-formed, so the reference's own lines can still be run, unmodified, by handing them a context that HAS
    the misspelled attribute: a subclass whose setup_context (gdnsq.py:21-23, called unchanged) also sets
    `ctx.need_input_grad = ctx.needs_input_grad`, installed under the name `QNEWGS` in the reference module's namespace
    (Quantizer._get_rnoise looks the name up at call time, gdnsq.py:234-235) for the duration of the block.  The
    vectors recorded under it are what the reference computes with that one attribute spelled either way."""

    def __init__(self, R):
        import src.quantization.gdnsq.gdnsq as ref
        self.ref = ref

    def __enter__(self):
        ref = self.ref

        class QNEWGS(ref.QNEWGS):                 # same name: autograd node names stay `QNEWGSBackward`
            @staticmethod
            def setup_context(ctx, inputs, output):
                ref.QNoise.setup_context(ctx, inputs, output)
                ctx.need_input_grad = ctx.needs_input_grad

        self.orig, ref.QNEWGS = ref.QNEWGS, QNEWGS
        return self

    def __exit__(self, *exc):
        self.ref.QNEWGS = self.orig
        return False


def gen_ewgs(R):
    """NoisyAct / NoisyConv2d / NoisyLinear with qnmethod=EWGS through the reference's own forward and backward lines
    (see ewgs_enabled); same field layout as act_cases.npz / weight_cases.npz.  First: the shipped reference raises."""
    m = R.NoisyAct(qnmethod=R.QNMethod.EWGS)
    try:
        m(torch.ones(1, 1, 2, 2, requires_grad=True)).sum().backward()
        raise SystemExit("the reference's QNEWGS.backward no longer raises: regenerate with the plain classes")
    except AttributeError as e:
        assert "need_input_grad" in str(e)
    acts, wgts = {}, {}
    gen = torch.Generator().manual_seed(9753)

    def rn(*s):
        return torch.randn(*s, generator=gen)

    with ewgs_enabled(R):
        shape, shape3 = (2, 8, 6, 6), (4, 16, 12, 12)
        x = rn(*shape) * 2
        acts.update(act_case(R, "ewgs_in_range_pow2", x, rn(*shape), -6.0, 4.0, -8.0, True, 31, "EWGS"))
        x = rn(*shape) * 4
        acts.update(act_case(R, "ewgs_clip_both_pow2", x, rn(*shape), -3.0, 1.0, -1.0, True, 32, "EWGS"))
        x = rn(*shape3) * 2
        rng_ = (x.max() - x.min()).item()
        log_s = math.log2(rng_ * 0.8 / 15.0)
        acts.update(act_case(R, "ewgs_big_w4", x, rn(*shape3), log_s, log_s + 4, x.min().item() * 0.8, True, 33, "EWGS"))
        x = torch.relu(rn(*shape) * 2)
        acts.update(act_case(R, "ewgs_unsigned", x, rn(*shape), -4.0, 2.5, 0.0, False, 34, "EWGS"))
        log_s = math.log2(rng_ / 1023.0)
        x = rn(3, 5, 7, 9) * 2
        acts.update(act_case(R, "ewgs_calibrated10_ragged", x, rn(3, 5, 7, 9), log_s, log_s + 10, x.min().item(), True, 35,
                             "EWGS"))
        # weights
        wshape = (8, 4, 3, 3)
        for pc in (False, True):
            w = rn(*wshape) * math.sqrt(2.0 / 36)
            if pc:
                mx, mn = w.amax((1, 2, 3)), w.amin((1, 2, 3))
                log_s = torch.log2((mx - mn) / 15.0) + 0.1 * rn(wshape[0])
            else:
                log_s = math.log2((w.max() - w.min()).item() / 15.0) + 0.05
            wgts.update(weight_case(R, f"ewgs_{'pc' if pc else 'pt'}", w, rn(*wshape), log_s, pc, "EWGS", 300 + int(pc)))
        w = rn(*wshape) * 0.3                      # tied minima per channel / per tensor
        w[0].flatten()[[1, 7, 20]] = w[0].min() - 0.05
        w[1].flatten()[[0, 35]] = w[1].min() - 0.01
        wgts.update(weight_case(R, "tied_ewgs_pc", w, rn(*wshape), -3.0, True, "EWGS", 310))
        wt = w.clone()
        wt.flatten()[[5, 100]] = wt.min() - 0.1
        wgts.update(weight_case(R, "tied_ewgs_pt", wt, rn(*wshape), -3.0, False, "EWGS", 311))
        wshape2 = (32, 16, 3, 3)
        w = rn(*wshape2) * math.sqrt(2.0 / 144)
        mx, mn = w.amax((1, 2, 3)), w.amin((1, 2, 3))
        log_s = torch.maximum(torch.full((32,), -12.0), torch.log2((mx - mn) / 1023.0))
        wgts.update(weight_case(R, "wide_ewgs_pc", w, rn(*wshape2), log_s, True, "EWGS", 320))
        wgts.update(weight_case(R, "wide_ewgs_pt", w, rn(*wshape2), -6.3, False, "EWGS", 321))
        w = rn(*wshape) * 0.3
        b = rn(wshape[0]) * 0.1 + 0.2
        wgts.update(weight_case(R, "qbias_ewgs_pc", w, rn(*wshape), -4.0, True, "EWGS", 330, bias=b, Gb=rn(wshape[0])))
        wl = rn(10, 64) * 0.2
        wgts.update(weight_case(R, "linear_ewgs_pt", wl, rn(10, 64), -5.0, False, "EWGS", 340, linear=True))
    return acts, wgts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    args = ap.parse_args()
    torch.set_num_threads(1)
    R = import_reference()
    os.makedirs(args.out, exist_ok=True)
    for fname, fn in (("act_cases.npz", gen_act), ("weight_cases.npz", gen_weight),
                      ("model_cases.npz", gen_model), ("calib_cases.npz", gen_calib), ("stats_cases.npz", gen_stats)):
        data = fn(R)
        path = os.path.join(args.out, fname)
        np.savez_compressed(path, **data)
        names = sorted({k.split("__")[0] for k in data})
        print(f"{fname}: {len(names)} cases, {os.path.getsize(path)/1024:.1f} KiB: {', '.join(names)}")
    acts, wgts = gen_ewgs(R)
    for fname, data in (("ewgs_act_cases.npz", acts), ("ewgs_weight_cases.npz", wgts)):
        path = os.path.join(args.out, fname)
        np.savez_compressed(path, **data)
        names = sorted({k.split("__")[0] for k in data})
        print(f"{fname}: {len(names)} cases, {os.path.getsize(path)/1024:.1f} KiB: {', '.join(names)}")


if __name__ == "__main__":
    main()
