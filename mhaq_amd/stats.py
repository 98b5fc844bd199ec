"""Bit-width statistics of a quantized model, same function names and
results as /root/reference/src/quantization/gdnsq/utils/model_stats.py:116-262, without the reference's
per-channel Python loop of `.item()` host syncs: the level count of a weight group follows from its min and max
(one read-only HIP sweep, see _layer_bit_widths; `_weight_indices` keeps the literal q for checkers), and each
public function syncs once.
"""
from __future__ import annotations

import contextlib
import math

import torch

from . import _lib, ops
from .enums import QScheme
from .layers import NoisyAct, NoisyConv2d, NoisyLinear


@ops._on_device
@torch.no_grad()
def _weight_indices(module) -> torch.Tensor:
    """q = Quantizer.quantize(weight) (gdnsq.py:189-219) for a weight layer, from one kernel launch."""
    w = ops._require_cuda_f32(module.weight.detach(), "weight")
    s = torch.exp2(module.log_wght_s.detach())
    L = _lib.lib()
    q = torch.empty_like(w)
    if module.qscheme == QScheme.PER_CHANNEL:
        co = w.shape[0]
        wq = torch.empty_like(w)
        zp = torch.empty(co, dtype=torch.float32, device=w.device)
        _lib.check(L.mhaq_fq_pc_fwd(w.data_ptr(), wq.data_ptr(), zp.data_ptr(), q.data_ptr(),
                                    s.contiguous().data_ptr(), co, w.numel() // co, ops._stream()), "mhaq_fq_pc_fwd")
        return q
    zp = ops.minmax(w)[0:1]
    _, q, _, _ = ops._pt_forward(w, ops._scalar(s, w.device, "scale"), zp,
                                 ops._scalar(-math.inf, w.device, "lo"), ops._scalar(math.inf, w.device, "hi"),
                                 want_q=True)
    return q


def val_count(q) -> float:
    mm = q.aminmax()
    return (mm.max - mm.min + 1).item()


@torch.no_grad()
def _layer_bit_widths(module) -> torch.Tensor:
    """log2(#levels) per channel (PER_CHANNEL) or a 1-element tensor (PER_TENSOR), on the device.

    The reference quantizes the whole weight and takes max(q) - min(q) + 1 per channel (model_stats.py:116-132).
    Every step of q = rne((w - zp) / s) is monotone in w (fp32 subtraction, division by s > 0 and rounding all are),
    and zp is the group minimum, so min(q) = q(zp) = 0 and max(q) = q(max w): the level count follows from the
    group's min and max alone -- ONE read-only HIP sweep (mhaq_fq_row_minmax / mhaq_fq_minmax, 4 B/elem) instead of a
    forward kernel that materialises q plus two reductions over it (20 B/elem).  The [Co]-sized tail below is the
    same fp32 op chain the quantizer applies to that one element."""
    if _memo is None:
        return _layer_bit_widths_uncached(module)
    bw = _memo.get(id(module))
    if bw is None:
        bw = _memo[id(module)] = _layer_bit_widths_uncached(module)
    return bw


_memo = None


@contextlib.contextmanager
def memoised():
    """The reference recomputes every layer's width three times per validation batch (mean, max, is_converged:
    gdnsq_quant.py:258-290).  Inside this context each layer is swept once; nothing is kept beyond it, so a weight
    update between two validation steps can never meet a stale value."""
    global _memo
    prev, _memo = _memo, {}
    try:
        yield
    finally:
        _memo = prev


@torch.no_grad()
def _layer_bit_widths_uncached(module) -> torch.Tensor:
    w = ops._require_cuda_f32(module.weight.detach(), "weight", any_dense_layout=True)
    s = torch.exp2(module.log_wght_s.detach())
    if module.qscheme == QScheme.PER_CHANNEL:
        mn, mx = ops.row_minmax(w)
        s = s.reshape(-1)
    else:
        mm = ops.minmax(w)
        mn, mx = mm[0:1], mm[1:2]
        s = s.reshape(1)
    v = (mx - mn) / s
    qmax = v + (torch.round(v) - v)
    return torch.log2(qmax + 1)


def get_true_layer_bit_width(module, max=True):
    bw = _layer_bit_widths(module)
    return float(bw.max() if max else bw.mean()) if module.qscheme == QScheme.PER_CHANNEL else float(bw)


def get_true_weights_width(model, max=True):
    layers = [m for m in model.modules() if isinstance(m, (NoisyConv2d, NoisyLinear))]
    per_layer = torch.stack([_layer_bit_widths(m).max() for m in layers])      # reference: layer max always
    return float(per_layer.max() if max else per_layer.mean())                  # the single host sync


def get_true_activations_width(model, max=True):
    bws = [m.bw.to(torch.float32).reshape(()) for m in model.modules() if isinstance(m, NoisyAct)]
    dev = next((b.device for b in bws if b.is_cuda), torch.device("cpu"))
    bws = torch.stack([b.to(dev) for b in bws])          # one host sync for the whole model, not one per quantizer
    return float(bws.max() if max else bws.mean())


def get_activations_bit_width(log_q, log_s, b):
    return (log_q - log_s).mean()


def get_activations_bit_width_mean(model):
    acts = [m for m in model.modules() if isinstance(m, NoisyAct)]
    return torch.stack([get_activations_bit_width(m.log_act_q.detach(), m.log_act_s.detach(), m.act_b.detach())
                        for m in acts]).mean()


@torch.no_grad()
def get_layer_wnb_bit_width(layer_weights, log_s, config=QScheme.PER_TENSOR):
    if config == QScheme.PER_TENSOR:
        mm = ops.minmax(layer_weights)
        mn, mx = mm[0], mm[1]
    else:
        mn, mx = ops.row_minmax(layer_weights)
    log_q = torch.log2((mx - mn).reshape(log_s.shape) + torch.exp2(log_s))
    return get_activations_bit_width(log_q, log_s, 0)


def get_weights_bit_width_mean(model):
    vals = []
    for m in model.modules():
        if isinstance(m, (NoisyConv2d, NoisyLinear)):
            bw = get_layer_wnb_bit_width(m.weight.detach(), m.log_wght_s.detach(), m.qscheme)
            vals.append(bw.mean())
    t = torch.stack(vals)
    return t[~torch.isnan(t)].mean()


def is_converged(model, criterion) -> bool:
    """model_stats.is_converged: true widths within the targets of the PotentialLoss (`wt`, `at`)."""
    return bool(get_true_weights_width(model) <= criterion.wt and get_true_activations_width(model) <= criterion.at)
