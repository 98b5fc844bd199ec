"""Bit-width statistics of a quantized model, same function names and
results as /root/reference/src/quantization/gdnsq/utils/model_stats.py:116-262, without the reference's
per-channel Python loop of `.item()` host syncs: the rounding indices come from the HIP forward kernel
(mhaq_fq_pc_fwd / mhaq_fq_pt_fwd with q_out), the per-channel min/max from two device reductions, and
each public function syncs once.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib, ops
from .enums import QScheme
from .layers import NoisyAct, NoisyConv2d, NoisyLinear


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
    """log2(#levels) per channel (PER_CHANNEL) or a 1-element tensor (PER_TENSOR), on the device."""
    q = _weight_indices(module)
    if module.qscheme == QScheme.PER_CHANNEL:
        flat = q.reshape(q.shape[0], -1)
        return torch.log2(flat.amax(1) - flat.amin(1) + 1)
    mm = q.aminmax()
    return torch.log2(mm.max - mm.min + 1).reshape(1)


def get_true_layer_bit_width(module, max=True):
    bw = _layer_bit_widths(module)
    return float(bw.max() if max else bw.mean()) if module.qscheme == QScheme.PER_CHANNEL else float(bw)


def get_true_weights_width(model, max=True):
    layers = [m for m in model.modules() if isinstance(m, (NoisyConv2d, NoisyLinear))]
    per_layer = torch.stack([_layer_bit_widths(m).max() for m in layers])      # reference: layer max always
    return float(per_layer.max() if max else per_layer.mean())                  # the single host sync


def get_true_activations_width(model, max=True):
    bws = torch.stack([m.bw.to(torch.float32).reshape(()).cpu() for m in model.modules() if isinstance(m, NoisyAct)])
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
        dims = tuple(range(1, layer_weights.dim()))
        mn, mx = layer_weights.amin(dims), layer_weights.amax(dims)
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
