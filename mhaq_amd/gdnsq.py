"""Host-side mirror of the reference's operator seam: `Quantizer`
(/root/reference/src/quantization/gdnsq/gdnsq.py:159-241), same constructor, same public
mutable attributes (`module, scale, zero_point, min_val, max_val, rnoise_ratio,
positive_scale, qnmethod`) that the layer wrappers overwrite every forward, same method
names.  All arithmetic runs in the HIP kernels behind include/mhaq_fq.h.

`fake_quant(x)` is the fused quantize->dequantize the layers use in the step loop (in the
reference the pair is always called back to back, gdnsq_act.py:50-55, gdnsq_conv2d.py:98);
`quantize` / `dequantize` keep the two-method facade for side consumers
(utils/model_stats.py:116-132).
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import ops
from .enums import QNMethod

_ASSERT_MESSAGES = (
    (1, "Not all elements in the tensor above min val"),      # gdnsq.py:213
    (2, "Not all elements in the tensor below max val"),      # gdnsq.py:215
    (4, "Not all elements in the tensor have integer values."),  # gdnsq.py:217
)


def raise_on_flags(flags) -> None:
    """Turn the device-side integrity flag word into the reference's AssertionError
    (gdnsq.py:211-217).  This is the single host sync; call it once per validation step."""
    word = int(flags.item()) if torch.is_tensor(flags) else int(flags)
    for bit, msg in _ASSERT_MESSAGES:
        if word & bit:
            raise AssertionError(msg)


def check_model_integrity(model) -> None:
    """One host sync for a whole model: OR the eval-mode flag words of every quantizer that ran and
    raise the reference's AssertionError (gdnsq.py:211-217) if any bit is set.  Call once per
    validation step instead of the reference's three syncs per quantizer per batch."""
    words = [m.Q.last_flags for m in model.modules()
             if hasattr(m, "Q") and getattr(m.Q, "last_flags", None) is not None]
    if words:
        raise_on_flags(_or_reduce(words))


def _or_reduce(words) -> int:
    acc = 0
    for v in torch.stack([w.reshape(()) for w in words]).cpu().tolist():
        acc |= int(v)
    return acc


class Quantizer:
    def __init__(
        self,
        module: torch.nn.modules.Module,
        scale: torch.Tensor,
        zero_point: torch.Tensor,
        min_val: torch.Tensor,
        max_val: torch.Tensor,
        rnoise_ratio: torch.Tensor = torch.Tensor([-1.0, ]),
        qnmethod: QNMethod = QNMethod.STE,
    ) -> None:
        self.module = module
        self.scale = scale
        self.zero_point = zero_point
        self.min_val = min_val
        self.max_val = max_val
        self.rnoise_ratio = torch.Tensor([rnoise_ratio])
        self.positive_scale = torch.all(torch.as_tensor(self.scale) > 0).item()
        self.qnmethod = qnmethod
        self.last_flags = None  # device int32[1]: OR of MHAQ_FQ_FLAG_* from the last eval forward

    def __deepcopy__(self, memo):
        # the layers re-assign scale / zero_point / bounds every forward, often as non-leaf tensors (exp2 of the
        # log-parameter, as in the reference's constructor): a copy carries their values, not their graph
        import copy
        new = object.__new__(type(self))
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = v.detach().clone() if torch.is_tensor(v) else copy.deepcopy(v, memo)
        return new

    # -- fused hot path -----------------------------------------------------------------
    def _is_per_tensor(self) -> bool:
        return (not torch.is_tensor(self.scale)) or self.scale.numel() == 1

    def fake_quant(self, value: Tensor, r_sign=None) -> Tensor:
        """dequantize(quantize(value)) in one kernel (per-tensor parameters)."""
        ops._method_value(self.qnmethod)  # AttributeError for an unknown method (gdnsq.py:241)
        if not self.positive_scale:
            # gdnsq.py:201-202,226-227: degenerate quantizer built with a non-positive scale
            # skips divide/round; value-preserving clamp only (never hit by any shipped config).
            return (torch.clamp(value, min=self.min_val, max=self.max_val) - self.zero_point) + self.zero_point
        if not self._is_per_tensor():
            raise ValueError("fake_quant(): per-channel scales go through ops.fake_quant_weight_pc")
        return ops.fake_quant_per_tensor(value, self.scale, self.zero_point, self.min_val, self.max_val,
                                         self.qnmethod, r_sign)

    def fake_quant_eval(self, value: Tensor):
        """Eval-mode fused forward: (y, qstats[2], flags[1]); flags are checked lazily."""
        y, _, qstats, flags = ops.fake_quant_per_tensor_eval(value, self.scale, self.zero_point,
                                                            self.min_val, self.max_val)
        self.last_flags = flags
        return y, qstats, flags

    # -- two-method facade ---------------------------------------------------------------
    def quantize(self, value: Tensor) -> Tensor:
        """Rounding indices q (integer-valued fp32), gdnsq.py:189-219."""
        from . import ops_generic
        return ops_generic.quantize(self, value)

    def dequantize(self, quantized_value: Tensor) -> Tensor:
        """gdnsq.py:221-229."""
        from . import ops_generic
        return ops_generic.dequantize(self, quantized_value)

    def _get_rnoise(self, value: Tensor, scale: Tensor):
        from . import ops_generic
        return ops_generic.round_noise(value, scale, self.qnmethod)

    def check_integrity(self) -> None:
        if self.last_flags is not None:
            raise_on_flags(self.last_flags)
