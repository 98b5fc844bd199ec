"""The unfused facade: the reference's custom autograd Functions and the two-method
Quantizer API, for callers that do not go through the fused layer path.

  QNoise, QNSTE, QNLSQ, QNEWGS, QNAEWGS   /root/reference/src/quantization/gdnsq/gdnsq.py:11-147
  quantize / dequantize / round_noise     gdnsq.py:189-241

The Functions' forward and backward are HIP kernels (mhaq_fq_noise_fwd / mhaq_fq_noise_bwd);
the affine glue around them (clamp, -zp, /s, *s, +zp) is the same aten chain the reference
runs, on the GPU, so autograd gives the reference's gradients for every scale / zero-point /
bound shape the reference accepts.  The step loop never comes through here: the layers call
the fused ops in mhaq_amd/ops.py.
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


def quantize(Q, value):
    """Quantizer.quantize (gdnsq.py:189-219), op for op."""
    ops._require_cuda_f32(value, "value")
    value = torch.clamp(value, min=Q.min_val, max=Q.max_val)
    value = value - Q.zero_point
    if not Q.positive_scale:
        return value
    value = value / Q.scale
    noise = Q._get_rnoise(value, Q.scale)
    value = value + noise
    if not Q.module.training:
        if torch.any(value < torch.floor((Q.min_val - Q.zero_point) / Q.scale)):
            raise AssertionError("Not all elements in the tensor above min val")
        if torch.any(value > torch.ceil((Q.max_val - Q.zero_point) / Q.scale)):
            raise AssertionError("Not all elements in the tensor below max val")
        if not torch.all((value == value.floor()) | (value == value.ceil())):
            raise AssertionError("Not all elements in the tensor have integer values.")
    return value


def dequantize(Q, quantized_value):
    """Quantizer.dequantize (gdnsq.py:221-229)."""
    if not Q.positive_scale:
        return quantized_value + Q.zero_point
    return quantized_value * Q.scale + Q.zero_point
