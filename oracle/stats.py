"""CPU restatement of the reference's bit-width statistics  --  TEST INFRASTRUCTURE ONLY (see oracle/fq_eager.py for the
import rule; pinned against tests/golden/stats_cases.npz, recorded from the reference's own functions).

  true_layer_bit_width      get_true_layer_bit_width + val_count (gdnsq/utils/model_stats.py:116-138)
  layer_wnb_bit_width       get_layer_wnb_bit_width / get_activations_bit_width (:141-168, :229-237)
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import fq_eager as O


def val_count(q):
    mm = q.aminmax()
    return (mm.max - mm.min + 1).item()


def true_layer_bit_width(w, log_wght_s, per_channel: bool, max=True):
    """Quantize every weight like the layer's last forward did (zero point = group minimum) and count levels."""
    s = torch.exp2(log_wght_s)
    zp = O.weight_zero_point(w, per_channel)
    q = O.quantize(w, s, zp, -math.inf, math.inf, "LSQ")        # noise(v) = rne(v) - v for every estimator
    if not per_channel:
        return np.log2(val_count(q))
    widths = [np.log2(val_count(ch)) for ch in q.reshape(q.shape[0], -1)]
    return np.max(widths) if max else np.mean(widths)


def layer_wnb_bit_width(w, log_s, per_channel: bool):
    if per_channel:
        dims = tuple(range(1, w.dim()))
        mn, mx = w.amin(dims), w.amax(dims)
    else:
        mn, mx = w.amin(), w.amax()
    log_q = torch.log2((mx - mn).reshape(log_s.shape) + torch.exp2(log_s))
    return (log_q - log_s).mean()
