"""The unfused facade: the reference's custom autograd Functions and the two-method
Quantizer API, for callers that do not go through the fused layer path.

  QNoise, QNSTE, QNLSQ, QNEWGS, QNAEWGS   /root/reference/src/quantization/gdnsq/gdnsq.py:11-147
  quantize / dequantize / round_noise     gdnsq.py:189-241

The Functions' forward and backward are HIP kernels (mhaq_fq_noise_fwd / mhaq_fq_noise_bwd).
quantize / dequantize run on the fused kernels wherever one serves the quantizer's parameter
shapes (per-tensor: mhaq_fq_pt_fwd / _bwd as ONE pair node; per-channel without gradients:
mhaq_fq_pc_quantize), with the eval-mode asserts as a device flag word; the remaining shapes take
torch's broadcasting arithmetic around the noise Functions (see the route table below).  The step
loop never comes through here: the layers call the fused ops in mhaq_amd/ops.py.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import _lib, ops
from .enums import QNMethod


def _groups_of(v: torch.Tensor, scale: torch.Tensor):
    """(groups, len, period) describing how grad_scale sums to scale.shape (autograd sum_to) and how
    AEWGS' reduce_to_shape (gdnsq.py:150-152) groups its statistics; None = per-element scale."""
    if scale.numel() == 1:
        unit = {i for i, n in enumerate(scale.shape) if n == 1}      # dims reduce_to_shape averages over
        alld = set(range(v.dim()))
        if not unit or unit >= alld:      # 0-dim scale (mean over no dims = all dims) or all-ones shape
            period = 1
        elif unit == {0}:                 # the [1]-shaped-scale quirk: statistics per position i % period
            period = max(1, v.numel() // max(1, v.shape[0]))
        else:
            raise NotImplementedError(f"AEWGS grouping for scale shape {tuple(scale.shape)}")
        return 1, v.numel(), period
    if (v.dim() > 1 and scale.numel() == v.shape[0] and scale.dim() == v.dim()
            and all(n == 1 for n in scale.shape[1:])):
        return v.shape[0], v.numel() // v.shape[0], 0      # [C,1,..]: reduce_to_shape averages dims 1..
    if scale.shape == v.shape:
        return None      # per-element scale, no unit dims: reduce_to_shape averages over everything
    raise NotImplementedError(
        f"scale shape {tuple(scale.shape)} against input {tuple(v.shape)}: only per-tensor, per-output-channel "
        "and per-element scales exist in the reference's layers")


@ops._on_device
def _noise_forward(v):
    out = torch.empty_like(v)
    _lib.check(_lib.lib().mhaq_fq_noise_fwd(v.data_ptr(), out.data_ptr(), v.numel(), ops._stream()),
               "mhaq_fq_noise_fwd")
    return out


@ops._on_device
def _noise_backward(v, scale, g, method: int, r_sign=None):
    L = _lib.lib()
    dev = v.device
    g = g.contiguous()
    layout = _groups_of(v, scale)
    one = ops._scalar(1.0, dev, "one")
    zero = ops._scalar(0.0, dev, "zero")
    if layout is None:  # per-element scale (the quantized bias): groups of length 1
        groups, length, period = v.numel(), 1, 0
    else:
        groups, length, period = layout
    stats = None
    if method == QNMethod.AEWGS.value:
        if layout is None:
            stats = torch.empty(3, dtype=torch.float32, device=dev)
            ones = torch.ones_like(v)
            _lib.check(L.mhaq_fq_vec_aewgs_stats(v.data_ptr(), g.data_ptr(), ones.data_ptr(),
                                                 torch.zeros_like(v).data_ptr(), v.numel(), stats.data_ptr(),
                                                 ops._stream()), "mhaq_fq_vec_aewgs_stats")
            period, stats = 1, stats.reshape(3, 1)
        elif groups == 1:
            co = v.shape[0] if (v.dim() > 0 and period > 1) else v.numel()
            row = period
            stats = torch.empty(3, row, dtype=torch.float32, device=dev)
            nbc = L.mhaq_fq_pt_aewgs_colstats_workspace_bytes(co, row)
            wsc = ops._workspace(nbc, dev) if nbc else None
            _lib.check(L.mhaq_fq_pt_aewgs_colstats(v.data_ptr(), g.data_ptr(), co, row, one.data_ptr(),
                                                   zero.data_ptr(), None, None, stats.data_ptr(),
                                                   wsc.data_ptr() if wsc is not None else None, nbc,
                                                   ops._stream()),
                       "mhaq_fq_pt_aewgs_colstats")
        else:
            stats = torch.empty(3, groups, dtype=torch.float32, device=dev)
            ones = torch.ones(groups, dtype=torch.float32, device=dev)
            zeros = torch.zeros(groups, dtype=torch.float32, device=dev)
            _lib.check(L.mhaq_fq_pc_aewgs_stats(v.data_ptr(), g.data_ptr(), ones.data_ptr(), zeros.data_ptr(),
                                                groups, length, stats.data_ptr(), ops._stream()),
                       "mhaq_fq_pc_aewgs_stats")
        ops._allreduce_avg_(stats)
    gv = torch.empty_like(v)
    gs = torch.empty(groups, dtype=torch.float32, device=dev)
    nb = L.mhaq_fq_noise_bwd_workspace_bytes(groups, length)
    ws = ops._workspace(nb, dev)
    r_sign, seed, offset, odev = ops._signs(r_sign, method, v)
    _lib.check(L.mhaq_fq_noise_bwd(v.data_ptr(), g.data_ptr(), gv.data_ptr(), gs.data_ptr(), groups, length, method,
                                   stats.data_ptr() if stats is not None else None, period,
                                   r_sign.data_ptr() if r_sign is not None else None, seed, offset, odev,
                                   ws.data_ptr(), nb, ops._stream()), "mhaq_fq_noise_bwd")
    return gv, gs.reshape(scale.shape)


class QNoise(Function):
    """noise = round(input) - input (gdnsq.py:11-29).  Use a derived class."""
    _method = None
    r_sign = None  # class-level test hook: int8 +-1 tensor replacing the in-kernel Philox stream

    @staticmethod
    def forward(input, scale):
        return _noise_forward(ops._require_cuda_f32(input, "input"))

    @staticmethod
    def setup_context(ctx, inputs, output):
        input, scale = inputs
        ctx.save_for_backward(input.contiguous(), scale)

    @staticmethod
    def backward(ctx, grad_output):
        raise AttributeError("You can't use QNoise directly. Use derivative classes instead")


def _make(name, method):
    def backward(ctx, grad_output):
        v, scale = ctx.saved_tensors
        gv, gs = _noise_backward(v, scale, grad_output, method.value, r_sign=cls.r_sign)
        return (gv if ctx.needs_input_grad[0] else None), (gs if ctx.needs_input_grad[1] else None)
    cls = type(name, (QNoise,), {"backward": staticmethod(backward), "_method": method,
                                 "__doc__": f"{name}: QNoise with the {method.name} estimator (gdnsq.py)."})
    return cls


QNSTE = _make("QNSTE", QNMethod.STE)
QNLSQ = _make("QNLSQ", QNMethod.LSQ)
QNEWGS = _make("QNEWGS", QNMethod.EWGS)      # the reference raises AttributeError in backward (gdnsq.py:102)
QNAEWGS = _make("QNAEWGS", QNMethod.AEWGS)
_BY_METHOD = {QNMethod.STE: QNSTE, QNMethod.EWGS: QNEWGS, QNMethod.AEWGS: QNAEWGS, QNMethod.LSQ: QNLSQ}


def scaled_noise(x, s):
    return QNoise.apply(x, s)


def round_noise(value, scale, qnmethod):
    """Quantizer._get_rnoise (gdnsq.py:231-241)."""
    try:
        cls = _BY_METHOD[QNMethod(ops._method_value(qnmethod))]
    except (KeyError, ValueError):
        raise AttributeError(f"Unknown method {qnmethod}!")
    if not torch.is_tensor(scale):
        scale = torch.as_tensor(scale, dtype=torch.float32, device=value.device)
    return cls.apply(value, scale.to(value.device))


# ----------------------------------------------------------------------------- the two-method facade
# Quantizer.quantize / dequantize (gdnsq.py:189-229) for callers outside the step loop.  Routes, by what Q holds:
#   * per-tensor parameters (one-element scale, zero point and bounds: every NoisyAct, PER_TENSOR weights): ONE launch of
#     the fused forward kernel (mhaq_fq_pt_fwd) writes q -- and y = q * s + zp next to it, which a following
#     Q.dequantize(q) hands out instead of a second pass --; with gradients the pair is one autograd node whose backward is
#     the fused backward kernel (mhaq_fq_pt_bwd): SURVEY.md 8b's "both on top of one fused op".  In eval mode the three
#     asserts of gdnsq.py:211-217 are the launch's flag word (Q.last_flags, raised by Q.check_integrity()): no host sync.
#   * [C,1,..] scale and zero point with infinite bounds, no gradient asked for (model_stats.py:118,123 on detached
#     weights): one launch of mhaq_fq_pc_quantize with the zero point Q holds.
#   * everything else -- a per-channel quantizer someone differentiates through (the caller owns the zero point's amin
#     graph; the layers use the fused per-channel op, which folds it into gW), per-element parameters (the quantized bias),
#     per-channel parameters with finite bounds, a non-positive scale -- takes `_chain`: torch's broadcasting arithmetic
#     around the HIP noise kernels (QN*), whose autograd graph is the reference's for any shape it accepts.  Its eval-mode
#     checks are device-side too.
class _Pair:
    """What quantize() leaves on the q it returns so that dequantize(q) can hand out the y of the same launch."""
    __slots__ = ("Q", "y", "scale", "zero_point")

    def __init__(self, Q, y):
        self.Q, self.y, self.scale, self.zero_point = Q, y, Q.scale, Q.zero_point


def _one(v):
    return (not torch.is_tensor(v)) or v.numel() == 1


def _wants_grad(*ts):
    return torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in ts)


def _chain(value, scale, zero_point, min_val, max_val, qnmethod):
    """The affine glue in torch ops around the HIP noise Function: q for parameters of any broadcastable shape."""
    v = (torch.clamp(value, min=min_val, max=max_val) - zero_point) / scale
    return v + round_noise(v, scale, qnmethod)


def _eval_flags(q, Q):
    """gdnsq.py:211-217 as a device word (bit 1 below min, 2 above max, 4 not integer), no host sync."""
    lo = torch.floor((Q.min_val - Q.zero_point) / Q.scale)
    hi = torch.ceil((Q.max_val - Q.zero_point) / Q.scale)
    word = (torch.any(q < lo).to(torch.int32) + 2 * torch.any(q > hi).to(torch.int32)
            + 4 * (~torch.all((q == q.floor()) | (q == q.ceil()))).to(torch.int32))
    return word.reshape(1)


class QuantizePairPT(Function):
    """(q, y = q * s + zp, flags) of a per-tensor quantizer from ONE forward launch; backward = the fused backward kernel
    when only y carries a gradient (the pair used as a pair), the torch chain when q itself was differentiated."""

    @staticmethod
    def forward(ctx, x, s, zp, lo, hi, method, r_sign, want_flags):
        y, q, _, flags = ops._pt_forward(x, s, zp, lo, hi, want_q=True, want_stats=want_flags)
        ctx.save_for_backward(x, s, zp, lo, hi)
        ctx.method, ctx.r_sign = method, r_sign
        ctx.set_materialize_grads(False)
        if flags is None:
            flags = torch.empty(0, dtype=torch.int32, device=x.device)
        ctx.mark_non_differentiable(flags)
        return q, y, flags

    @staticmethod
    def backward(ctx, gq, gy, _gflags):
        x, s, zp, lo, hi = ctx.saved_tensors
        if gq is None and gy is None:
            return (None,) * 8
        if gq is None:
            gy = gy.contiguous()
            col_stats, period = None, 0
            if ctx.method == QNMethod.AEWGS.value:
                col_stats, period = ops._col_stats(x, gy, s, zp, lo, hi)
            gx, grads = ops._pt_backward(x, gy, s, zp, lo, hi, ctx.method, col_stats, period, ctx.r_sign)
            return (gx, grads[0].reshape(s.shape), grads[1].reshape(zp.shape), grads[2].reshape(lo.shape),
                    grads[3].reshape(hi.shape), None, None, None)
        # q was consumed by something other than dequantize: the reference's graph, rebuilt on the saved inputs
        with torch.enable_grad():
            ins = [t.detach().requires_grad_(True) for t in (x, s, zp, lo, hi)]
            cls = _BY_METHOD[QNMethod(ctx.method)]
            saved, cls.r_sign = cls.r_sign, (ctx.r_sign if ctx.r_sign is not None else cls.r_sign)
            try:
                qd = _chain(ins[0], ins[1], ins[2], ins[3], ins[4], QNMethod(ctx.method))
                outs, gouts = [qd], [gq]
                if gy is not None:
                    outs.append(qd * ins[1] + ins[2])
                    gouts.append(gy)
                grads = torch.autograd.grad(outs, ins, gouts, allow_unused=True)
            finally:
                cls.r_sign = saved
        return (*grads, None, None, None)


@ops._on_device
def _quantize_pt(Q, value, method):
    x = ops._require_cuda_f32(value, "value", any_dense_layout=True)
    dev = x.device
    s, zp = ops._scalar(Q.scale, dev, "scale"), ops._scalar(Q.zero_point, dev, "zero_point")
    lo, hi = ops._scalar(Q.min_val, dev, "min_val"), ops._scalar(Q.max_val, dev, "max_val")
    evalm = not Q.module.training
    if _wants_grad(x, s, zp, lo, hi):
        r = _BY_METHOD[QNMethod(method)].r_sign
        q, y, flags = QuantizePairPT.apply(x, s, zp, lo, hi, method, ops._r_ptr(r, x), evalm)
    else:
        y, q, _, flags = ops._pt_forward(x, s, zp, lo, hi, want_q=True, want_stats=evalm)
    if evalm:
        Q.last_flags = flags
    q._mhaq_pair = _Pair(Q, y)
    return q


@ops._on_device
def _quantize_pc(Q, value):
    x = ops._require_cuda_f32(value, "value")
    co = x.shape[0]
    s = ops._require_cuda_f32(Q.scale.detach(), "scale").reshape(co)
    zp = ops._require_cuda_f32(Q.zero_point.detach(), "zero_point").reshape(co)
    q, y = torch.empty_like(x), torch.empty_like(x)
    flags = torch.zeros(1, dtype=torch.int32, device=x.device) if not Q.module.training else None
    _lib.check(_lib.lib().mhaq_fq_pc_quantize(x.data_ptr(), q.data_ptr(), y.data_ptr(), s.data_ptr(), zp.data_ptr(), co,
                                              x.numel() // max(co, 1), flags.data_ptr() if flags is not None else None,
                                              ops._stream()), "mhaq_fq_pc_quantize")
    if flags is not None:
        Q.last_flags = flags
    q._mhaq_pair = _Pair(Q, y)
    return q


def _is_inf(v, sign):
    return (not torch.is_tensor(v)) and float(v) == sign * float("inf")


def _per_channel(Q, value):
    s, z = Q.scale, Q.zero_point
    if not (torch.is_tensor(s) and torch.is_tensor(z) and value.dim() > 1 and s.numel() == value.shape[0] > 1):
        return False
    want = (value.shape[0],) + (1,) * (value.dim() - 1)
    return tuple(s.shape) == want and tuple(z.shape) == want and _is_inf(Q.min_val, -1) and _is_inf(Q.max_val, 1)


def quantize(Q, value):
    """Quantizer.quantize (gdnsq.py:189-219): the rounding indices q of `value` (integer-valued fp32)."""
    ops._require_cuda_f32(value, "value")
    if not Q.positive_scale:            # gdnsq.py:201-202: a quantizer built with a non-positive scale only clamps and shifts
        return torch.clamp(value, min=Q.min_val, max=Q.max_val) - Q.zero_point
    method = ops._method_value(Q.qnmethod)          # AttributeError for an unknown estimator, as _get_rnoise raises it
    if _one(Q.scale) and _one(Q.zero_point) and _one(Q.min_val) and _one(Q.max_val) and (
            method != QNMethod.AEWGS.value or (torch.is_tensor(Q.scale) and Q.scale.dim() == 1)):
        # (AEWGS groups its statistics by the SHAPE of the scale, gdnsq.py:150-152: the fused backward implements the
        # [1]-shaped scale of the reference's layers -- means over dim 0 --; a 0-dim or [1,1,1,1] scale takes the chain)
        return _quantize_pt(Q, value, method)
    if _per_channel(Q, value) and not _wants_grad(value, Q.scale, Q.zero_point):
        return _quantize_pc(Q, value)
    q = _chain(value, Q.scale, Q.zero_point, Q.min_val, Q.max_val, Q.qnmethod)
    if not Q.module.training:
        Q.last_flags = _eval_flags(q.detach(), Q)
    return q


def dequantize(Q, quantized_value):
    """Quantizer.dequantize (gdnsq.py:221-229): q * scale + zero_point -- for a q that this Quantizer's quantize() has just
    produced with the parameters it still holds, the y the same launch wrote."""
    if not Q.positive_scale:
        return quantized_value + Q.zero_point
    pair = getattr(quantized_value, "_mhaq_pair", None)
    if pair is not None and pair.Q is Q and pair.scale is Q.scale and pair.zero_point is Q.zero_point:
        return pair.y
    return quantized_value * Q.scale + Q.zero_point
