"""CPU oracle for the MHAQ fake-quant hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch PyTorch-eager *restatement* of the reference's
algorithm, op for op, so that elementwise results are bit-identical to the
reference's own eager path and the autograd engine produces the same
parameter gradients.  It is pinned against golden vectors generated from the
real reference (tests/golden/, produced by oracle/gen_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product path (mhaq_amd/) never does: it runs the HIP
kernels behind include/mhaq_fq.h or fails loudly.

Reference lines restated here (all under /root/reference/src/quantization/gdnsq/):
  gdnsq.py:11-29    QNoise      noise = round(v) - v
  gdnsq.py:32-57    QNSTE       grad_v = 0*g ; grad_scale = 3^-1/2 * g * r
  gdnsq.py:60-84    QNLSQ       grad_v = 0*g ; grad_scale = g * (round(v)-v)
  gdnsq.py:87-107   QNEWGS      (reference raises AttributeError at :102, a misspelled `ctx.need_input_grad`; restated as
                                 intended and pinned by tests/golden/ewgs_*.npz: the reference's own lines run with that
                                 attribute supplied, oracle/gen_golden.py `ewgs_enabled`)
  gdnsq.py:110-147  QNAEWGS     adaptive element-wise gradient scaling
  gdnsq.py:150-152  reduce_to_shape
  gdnsq.py:189-229  Quantizer.quantize / dequantize
  layers/gdnsq_act.py:39-55      NoisyAct.forward
  layers/gdnsq_conv2d.py:71-100  NoisyConv2d.forward
  layers/gdnsq_linear.py:61-78   NoisyLinear.forward
"""
from __future__ import annotations

import math

import torch
import torch.distributed as dist

INV_SQRT3 = 3.0 ** -0.5  # python double, multiplied into an fp32 tensor like the reference does

METHODS = ("STE", "EWGS", "AEWGS", "LSQ")  # index == QNMethod value (gdnsq_utils.py:9-13)


def _method_name(method) -> str:
    if isinstance(method, str):
        name = method
    elif isinstance(method, int):
        name = METHODS[method]
    else:  # Enum
        name = method.name
    if name not in METHODS:
        raise AttributeError(f"Unknown method {method}!")  # gdnsq.py:241
    return name


def _allreduce_avg_(t: torch.Tensor) -> None:
    """dist.all_reduce(op=AVG) (gdnsq.py:127-129); gloo has no AVG so SUM/world there."""
    if dist.get_backend() == "gloo":
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t.div_(dist.get_world_size())
    else:
        dist.all_reduce(t, op=dist.ReduceOp.AVG)


def group_mean(t: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """Mean over every dim where `like` has extent 1 (gdnsq.py:150-152).

    Note the reference quirk: a per-tensor scale of shape [1] against a 4-D
    weight reduces dim 0 only, giving statistics of shape [1, Ci, kh, kw].
    """
    dims = tuple(i for i, n in enumerate(like.shape) if n == 1)
    return torch.mean(t, dim=dims, keepdim=True)


class RoundNoise(torch.autograd.Function):
    """noise(v) = rne(v) - v with the estimator selected by `method`.

    `r` is the +-0.5 tensor of the stochastic scale gradient; None draws it with
    torch.randint_like exactly as the reference does (gdnsq.py:54).
    """

    @staticmethod
    def forward(v, scale, method, r):
        return torch.round(v) - v

    @staticmethod
    def setup_context(ctx, inputs, output):
        v, scale, method, r = inputs
        ctx.save_for_backward(v, scale)
        ctx.method = _method_name(method)
        ctx.r = r

    @staticmethod
    def _random_scale_grad(ctx, v, g):
        r = ctx.r if ctx.r is not None else torch.randint_like(v, 2).sub_(0.5)
        return INV_SQRT3 * g * r

    @staticmethod
    def backward(ctx, g):
        v, scale = ctx.saved_tensors
        gv = gs = None
        m = ctx.method
        if m == "STE":
            if ctx.needs_input_grad[0]:
                gv = g * 0
            if ctx.needs_input_grad[1]:
                gs = RoundNoise._random_scale_grad(ctx, v, g)
        elif m == "LSQ":
            if ctx.needs_input_grad[0]:
                gv = g * 0
            if ctx.needs_input_grad[1]:
                gs = g * (torch.round(v) - v)
        elif m == "EWGS":
            if ctx.needs_input_grad[0]:
                e = torch.round(v) - v
                gv = -torch.abs(g) * e * 1e-2
            if ctx.needs_input_grad[1]:
                gs = RoundNoise._random_scale_grad(ctx, v, g)
        elif m == "AEWGS":
            if ctx.needs_input_grad[0]:
                e = torch.round(v) - v
                num_full = g.sign() * e
                num = group_mean(num_full, scale).detach()
                e2 = group_mean(e.square(), scale).detach()
                me = group_mean(e, scale).detach()
                if dist.is_available() and dist.is_initialized():
                    _allreduce_avg_(num)
                    _allreduce_avg_(e2)
                    _allreduce_avg_(me)
                den = (e2 - me.square()).clamp_min(1e-3)
                delta = num / den
                g_scale = (1.0 * delta * num_full).clamp_max(1 - 0.01)
                gv = -g * g_scale
            if ctx.needs_input_grad[1]:
                gs = RoundNoise._random_scale_grad(ctx, v, g)
        return gv, gs, None, None


def quantize(x, scale, zero_point, min_val, max_val, method="STE", r=None):
    """gdnsq.py:189-219 (train-mode arithmetic; the eval asserts are `check_integrity`)."""
    v = torch.clamp(x, min=min_val, max=max_val)
    v = v - zero_point
    v = v / scale
    return v + RoundNoise.apply(v, scale, method, r)


def dequantize(q, scale, zero_point):
    """gdnsq.py:221-229."""
    return q * scale + zero_point


def check_integrity(q, scale, zero_point, min_val, max_val) -> None:
    """Eval-mode assertions of gdnsq.py:211-217."""
    if torch.any(q < torch.floor((min_val - zero_point) / scale)):
        raise AssertionError("Not all elements in the tensor above min val")
    if torch.any(q > torch.ceil((max_val - zero_point) / scale)):
        raise AssertionError("Not all elements in the tensor below max val")
    if not torch.all((q == q.floor()) | (q == q.ceil())):
        raise AssertionError("Not all elements in the tensor have integer values.")


def act_fake_quant(x, log_act_s, log_act_q, act_b, r=None, method="STE"):
    """NoisyAct.forward (gdnsq_act.py:39-55).  Returns (y, q)."""
    s = torch.exp2(log_act_s)
    qr = torch.exp2(log_act_q)
    q = quantize(x, s, act_b, act_b, act_b + qr - s, method, r)
    return dequantize(q, s, act_b), q


def act_bit_width(q):
    """Eval-mode `bw` of gdnsq_act.py:51-54."""
    mm = q.aminmax()
    return torch.log2(mm.max - mm.min + 1)


def weight_zero_point(w, per_channel: bool):
    """gdnsq_conv2d.py:80-83 / gdnsq_linear.py:69-72 (amin is differentiable)."""
    if per_channel:
        return w.amin(tuple(range(1, w.dim())), keepdim=True)
    return w.amin()


def weight_fake_quant(w, log_wght_s, per_channel: bool, method="AEWGS", r=None):
    """NoisyConv2d / NoisyLinear weight path.  Returns (wq, q, zp)."""
    s = torch.exp2(log_wght_s)
    zp = weight_zero_point(w, per_channel)
    q = quantize(w, s, zp, -math.inf, math.inf, method, r)
    return dequantize(q, s, zp), q, zp


def bias_fake_quant(bias, w, log_wght_s, method="AEWGS", r=None):
    """quant_bias=True branch of gdnsq_conv2d.py:86-94 (per-channel only)."""
    s = torch.exp2(log_wght_s).ravel()
    zp = weight_zero_point(w, True).ravel()
    q = quantize(bias, s, zp, -math.inf, math.inf, method, r)
    return dequantize(q, s, zp)


def regulariser_inputs(weights, log_wght_s_list, per_channel: bool):
    """Weight half of ModelHelper.get_model_values (utils/model_helper.py:18-45):
    returns (cat log_wght_s, cat log2(max - min + 2^log_s))."""
    lws, lwq = [], []
    for w, ls in zip(weights, log_wght_s_list):
        if per_channel:
            mn, mx = w.amin((1, 2, 3)), w.amax((1, 2, 3))
            lws.append(ls.ravel())
        else:
            mn, mx = w.amin(), w.amax()
            lws.append(ls)
        lwq.append(torch.log2(mx - mn + torch.exp2(ls.ravel())))
    if per_channel:
        return torch.cat(lws), torch.cat(lwq)
    return torch.stack(lws).ravel(), torch.stack(lwq).ravel()


def potential_loss(base_loss, las, laq, lws, lwq, a_bits, w_bits, t, loss_sum, cnt,
                   lossless=False, p=1):
    """PotentialLoss / PotentialLossNoPred arithmetic (gdnsq_loss.py:47-71,129-153).

    Returns (ploss, rloss); the caller owns the running `loss_sum`/`cnt` state.
    """
    l_eps = torch.tensor(1e-3)
    z = torch.tensor(0)
    pw = torch.tensor(p)
    wloss0 = torch.max(z, (lwq - lws) - (w_bits - l_eps)).pow(pw)
    wloss = wloss0.mean()
    wact = (wloss0 > 0).sum()
    aloss0 = torch.max(z, (laq - las) - (a_bits - l_eps)).pow(pw)
    aloss = aloss0.mean()
    aact = (aloss0 > 0).sum()
    rloss = base_loss.pow(pw)
    calib_mul = loss_sum / cnt
    wmul = (wact + l_eps) / (wact + aact + l_eps)
    amul = (aact + l_eps) / (wact + aact + l_eps)
    l1, l2 = (1.0, t) if lossless else (t, 1.0)
    return calib_mul * l1 * (wmul * wloss + amul * aloss) + l2 * rloss, rloss
