"""Layer wrappers with the reference's names, constructor signatures, parameter
names/shapes and attributes (`Q`, `bw`), so checkpoints, ModelHelper.get_model_values,
model_stats and the calibration observers keep working unchanged:

  NoisyAct     /root/reference/src/quantization/gdnsq/layers/gdnsq_act.py:9-55
  NoisyConv2d  /root/reference/src/quantization/gdnsq/layers/gdnsq_conv2d.py:13-119
  NoisyLinear  /root/reference/src/quantization/gdnsq/layers/gdnsq_linear.py:13-94

The forward arithmetic is delegated to the fused HIP ops (mhaq_amd/ops.py).  On the hot path
(NoisyAct with STE/LSQ/EWGS, per-channel NoisyConv2d) even the scalar chain around the
quantizer (exp2 of the log-parameters, the clamp bounds, their backward) runs inside the
kernels; the remaining variants keep that chain as torch scalar ops in autograd.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn.functional as F
from torch import inf, nn

from . import ops
from .enums import QNMethod, QScheme
from .gdnsq import Quantizer


def _rnoise_ratio(m):
    """What the reference assigns to Q.rnoise_ratio every forward (gdnsq_conv2d.py:73-75): `_noise_ratio` or
    zeros_like(it).  The attribute is dead in the arithmetic; the zeros are cached so that keeping it current
    does not cost a fill launch per layer per step."""
    if m.rand_noise:
        return m._noise_ratio
    z = getattr(m, "_zero_ratio", None)
    if z is None or z.device != m._noise_ratio.device:
        z = torch.zeros_like(m._noise_ratio)
        m._zero_ratio = z
    return z


class NoisyAct(nn.Module):
    def __init__(
        self,
        init_s=-10,
        init_q=10,
        signed=True,
        noise_ratio=1,
        disable=False,
        qnmethod: QNMethod = QNMethod.STE,
    ) -> None:
        super().__init__()
        self.disable = disable
        self.signed = signed
        zero_point = 0.0 if not signed else -torch.exp2(torch.tensor(init_q - 1).float())
        self._act_b = torch.tensor([zero_point]).float()
        self._log_act_s = torch.tensor([init_s]).float()
        self._log_act_q = torch.tensor([init_q]).float()
        self._noise_ratio = torch.tensor(noise_ratio)
        self.log_act_q = nn.Parameter(self._log_act_q, requires_grad=True)
        self.act_b = nn.Parameter(self._act_b, requires_grad=bool(signed))
        self.log_act_s = nn.Parameter(self._log_act_s, requires_grad=True)
        self.Q = Quantizer(self, torch.exp2(self._log_act_s), 0, -inf, inf, qnmethod=qnmethod)
        self.bw = torch.tensor(0.0)

    def forward(self, x):
        if self.disable:
            return x
        Q = self.Q
        qm = Q.qnmethod                      # a public, freely re-assigned attribute: memoise its integer by identity
        if qm is not Q.__dict__.get("_qm_seen"):
            Q._qm_int = ops._method_value(qm)        # AttributeError for an unknown method, at use like gdnsq.py:241
            Q._qm_seen = qm
        method = Q._qm_int
        if method == QNMethod.AEWGS.value:
            return self._forward_unfused_params(x)
        # Hot path: the scalar chain s = 2^log_s, qr = 2^log_q, [b, b + qr - s] and its backward are
        # folded into the kernels (mhaq_fq_act_fwd / mhaq_fq_act_bwd): 2 launches per direction, one compiled
        # autograd node (csrc/torch_binding.cpp).
        if self.training:
            ref = self.__dict__.get("_hub")
            routed = None
            if ref is not None and ref.hub is not None:   # one finalize launch per backward pass for all quantizers
                routed = ref.hub.take(ref.slot)           # (act_hub.py)
            if routed is not None and x.is_cuda and x.dtype is torch.float32:
                # straight into the compiled node (its argument checks are the C++ ones): ~3 us less Python per call
                y, params, s, hi = ops.act_layer_routed(x, routed, method, ref)
            elif routed is not None:
                y, params, s, hi = ops._act_layer(x, routed[0], routed[1], routed[2], method, None, ref)
            else:
                y, params, s, hi = ops._act_layer(x, self.log_act_s, self.log_act_q, self.act_b, method)
            # keep the Quantizer's public attributes current for side consumers (model_stats, observers)
            Q.scale, Q.max_val = s, hi
            Q.zero_point = Q.min_val = self._parameters["act_b"]
            return y
        needs_graph = torch.is_grad_enabled() and (
            x.requires_grad or any(p.requires_grad for p in self.parameters()))
        y, params, qstats, flags = ops.fake_quant_act_layer_eval(x, self.log_act_s, self.log_act_q, self.act_b)
        self._publish(params)
        self.Q.last_flags = flags           # gdnsq.py:211-217, checked lazily (Quantizer.check_integrity)
        self.bw = torch.log2(qstats[1] - qstats[0] + 1)      # gdnsq_act.py:51-54
        if needs_graph:
            y, _ = ops.fake_quant_act_layer(x, self.log_act_s, self.log_act_q, self.act_b, method)
        return y

    def _publish(self, params):
        """Keep the Quantizer's public attributes current for side consumers (model_stats, observers)."""
        self.Q.scale = params[0:1]
        self.Q.zero_point = self.act_b
        self.Q.min_val = self.act_b
        self.Q.max_val = params[3:4]

    def __getstate__(self):
        # per-step plumbing (the hub reference) does not travel with torch.save(model) / copy.deepcopy(model)
        state = self.__dict__.copy()
        state.pop("_hub", None)
        return state

    def _forward_unfused_params(self, x):
        """The reference's literal scalar chain around the fused per-tensor op (AEWGS activations)."""
        s = torch.exp2(self.log_act_s)
        q = torch.exp2(self.log_act_q)
        self.Q.zero_point = self.act_b
        self.Q.min_val = self.act_b
        self.Q.max_val = self.act_b + q - s
        self.Q.scale = s
        if self.training:
            return self.Q.fake_quant(x)
        y, qstats, _ = self.Q.fake_quant_eval(x)
        self.bw = torch.log2(qstats[1] - qstats[0] + 1)
        return y


class _WeightQuantMixin:
    """The weight path shared by NoisyConv2d and NoisyLinear: W viewed as [out][row], fused layer kernels where
    the reference runs amin -> sub -> div -> round -> mul -> add (+ ModelHelper's amin / amax / log2 again)."""
    quant_bias = False

    def _quantized_weight(self):
        d = self.__dict__          # per-step plumbing is kept off nn.Module.__setattr__ (2 us per assignment)
        self.Q.rnoise_ratio.data = _rnoise_ratio(self)
        weight_p, log_s_p = self.weight, self.log_wght_s
        grad_on = torch.is_grad_enabled()
        key = (weight_p._version, log_s_p._version, grad_on)
        pre = d.get("_precomputed")
        if pre is not None:     # this step's weights were quantized by the multi-tensor launch (multi.py)
            d["_precomputed"] = None
            weight, zp, s, lwq, pkey = pre
            if pkey == key:
                d["_lwq"], d["_lwq_key"] = lwq, key
                self.Q.scale, self.Q.zero_point = s, zp
                return weight, s, zp
        per_channel = self.qscheme == QScheme.PER_CHANNEL
        if per_channel and self.quant_bias:
            # the quantized bias shares s and zp and sends gradient into both: the layer op keeps both differentiable
            weight, zp, s, lwq = ops.fake_quant_weight_layer(weight_p, log_s_p, self.Q.qnmethod, zp_grad=True)
            d["_lwq"], d["_lwq_key"] = lwq, key
        elif per_channel:
            # one launch: s = 2^log_s, row min/max, quantizer, and the regulariser input
            # log2(max - min + s) that ModelHelper.get_model_values would re-derive (wrap.py)
            pre = d.get("_pre_fwd")
            group = None
            if pre is not None:     # this step's forward ran in the model-wide launch (multi.py, forward-only mode)
                d["_pre_fwd"] = None
                group = pre[2]
                pre = pre[0] if pre[1] == (key[0], key[1], weight_p.data_ptr()) else None
            if pre is not None and group is not None and grad_on:
                # the layer's backward is part of its group's single launch (multi.py: _WeightGroup)
                weight, lwq = group[0].take(group[1])
                s, zp = pre[5], pre[6]          # already in the [co, 1, ..] shape (torch_binding.cpp: plan_forward)
            else:
                weight, zp, s, lwq = ops.fake_quant_weight_layer(weight_p, log_s_p, self.Q.qnmethod, pre=pre)
            d["_lwq"], d["_lwq_key"] = lwq, key
        elif ops.small_pt_layer_supported(weight_p, self.Q.qnmethod):
            # PER_TENSOR layer that fits one workgroup: whole layer + regulariser input in one launch -- or, under a
            # trainer's model-wide forward launch (multi.py), this layer's slice of it and a slot in its backward group
            pre = d.get("_pre_fwd")
            group = None
            if pre is not None:
                d["_pre_fwd"] = None
                ok = pre[1] == (key[0], key[1], weight_p.data_ptr())
                group = pre[2]
                pre = pre[0] if (ok and group is not None and grad_on) else None
            if pre is not None:
                weight, lwq = group[0].take(group[1])
                zp, s = pre[2].reshape(()), pre[1]
            else:
                weight, zp, s, lwq = ops.fake_quant_weight_layer_pt(weight_p, log_s_p, self.Q.qnmethod)
            d["_lwq"], d["_lwq_key"] = lwq, key
        else:
            # PER_TENSOR beyond one workgroup, or AEWGS (per-position statistics): the streaming layer op
            weight, zp, s, lwq = ops.fake_quant_weight_layer_ptl(weight_p, log_s_p, self.Q.qnmethod)
            d["_lwq"], d["_lwq_key"] = lwq, key
        self.Q.scale = s
        self.Q.zero_point = zp
        return weight, s, zp

    def regulariser_input(self):
        """log2(max - min + 2^log_wght_s) per channel from this step's forward, or None if stale."""
        d = self.__dict__
        lwq = d.get("_lwq")
        if lwq is not None and d.get("_lwq_key") == (self.weight._version, self.log_wght_s._version,
                                                     torch.is_grad_enabled()):
            return lwq
        return None

    def __getstate__(self):
        # per-step plumbing (slices of this step's slabs, cached constants) does not travel with
        # torch.save(model) / copy.deepcopy(model)
        state = self.__dict__.copy()
        for k in ("_pre_fwd", "_precomputed", "_lwq", "_lwq_key", "_zero_ratio"):
            state.pop(k, None)
        return state


class NoisyConv2d(_WeightQuantMixin, nn.Conv2d):
    def __init__(
        self,
        in_channels: int,
        out_channels: int,
        kernel_size: int | Tuple[int, int],
        stride: int | Tuple[int, int] = 1,
        padding: str | int | Tuple[int, int] = 0,
        dilation: int | Tuple[int, int] = 1,
        groups: int = 1,
        bias: bool = True,
        padding_mode: str = "zeros",
        device=None,
        dtype=None,
        qscheme: QScheme = QScheme.PER_TENSOR,
        log_s_init: float = -12,
        rand_noise: bool = False,
        quant_bias: bool = False,
        qnmethod: QNMethod = QNMethod.AEWGS,
    ) -> None:
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         bias, padding_mode, device, dtype)
        self.qscheme = QScheme(qscheme.value) if not isinstance(qscheme, QScheme) else qscheme

        if self.qscheme == QScheme.PER_TENSOR:
            self.log_wght_s = nn.Parameter(torch.Tensor([log_s_init]), requires_grad=True)
        elif self.qscheme == QScheme.PER_CHANNEL:
            self.log_wght_s = nn.Parameter(
                torch.empty((out_channels, 1, 1, 1)).fill_(log_s_init), requires_grad=True)
            self.log_b_s = nn.Parameter(torch.empty(1).fill_(log_s_init), requires_grad=True)
        self._noise_ratio = nn.Parameter(torch.Tensor([1]), requires_grad=False)
        self.Q = Quantizer(self, torch.exp2(self.log_wght_s), 0, -inf, inf, qnmethod=qnmethod)
        self.rand_noise = rand_noise
        self.quant_bias = quant_bias
        if self.quant_bias:
            # like the reference this needs log_b_s, i.e. raises AttributeError for PER_TENSOR
            self.Q_b = Quantizer(self, torch.exp2(self.log_b_s), 0, -inf, inf, qnmethod=qnmethod)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        weight, s, zp = self._quantized_weight()
        if self.quant_bias:
            self.Q_b.scale = s.ravel()
            self.Q_b.zero_point = zp.ravel()
            self.Q_b.rnoise_ratio.data = _rnoise_ratio(self)
            bias = ops.fake_quant_per_element(self.bias, self.Q_b.scale, self.Q_b.zero_point,
                                              self.Q_b.qnmethod)
        else:
            bias = self.bias
        return self._conv_forward(input, weight, bias)

    def extra_repr(self) -> str:
        # nn.Conv2d's own summary plus the quantizer state (what print(model) shows; values, not the reference's text)
        ls = self.log_wght_s.detach()
        grid = f"log_wght_s[{ls.numel()}] in [{float(ls.min()):.3f}, {float(ls.max()):.3f}]"
        return (f"{super().extra_repr()}, qscheme={self.qscheme.name}, estimator={getattr(self.Q.qnmethod, 'name', self.Q.qnmethod)}, "
                f"{grid}, quant_bias={self.quant_bias}, rand_noise={self.rand_noise}")


class NoisyLinear(_WeightQuantMixin, nn.Linear):
    def __init__(
        self,
        in_features: int,
        out_features: int,
        bias: bool = True,
        device=None,
        dtype=None,
        qscheme: QScheme = QScheme.PER_TENSOR,
        log_s_init: float = -12,
        rand_noise: bool = False,
        qnmethod: QNMethod = QNMethod.STE,
    ) -> None:
        super().__init__(in_features, out_features, bias, device, dtype)
        self.qscheme = QScheme(qscheme.value) if not isinstance(qscheme, QScheme) else qscheme
        if self.qscheme == QScheme.PER_TENSOR:
            self.log_wght_s = nn.Parameter(torch.Tensor([log_s_init]), requires_grad=True)
        elif self.qscheme == QScheme.PER_CHANNEL:
            # the reference ends up with shape [out,1,1,1] (gdnsq_linear.py:59-62) and then raises
            # IndexError in forward (amin((1,2,3)) on a 2-D weight, :70-71); the shape is kept for
            # checkpoint compatibility and the forward works as intended (row minimum per output).
            self.log_wght_s = nn.Parameter(
                torch.empty((out_features, 1, 1, 1)).fill_(log_s_init), requires_grad=True)
        self._noise_ratio = nn.Parameter(torch.Tensor([1, ]), requires_grad=False)
        self.Q = Quantizer(self, torch.exp2(self.log_wght_s), 0, -inf, inf, qnmethod=qnmethod)
        self.rand_noise = rand_noise

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        # same fused weight path as NoisyConv2d (per-channel works as intended: one row per output feature)
        weight, _, _ = self._quantized_weight()
        return F.linear(input, weight, self.bias)

    def extra_repr(self) -> str:
        ls = self.log_wght_s.detach()
        grid = f"log_wght_s[{ls.numel()}] in [{float(ls.min()):.3f}, {float(ls.max()):.3f}]"
        return (f"{super().extra_repr()}, qscheme={self.qscheme.name}, estimator={getattr(self.Q.qnmethod, 'name', self.Q.qnmethod)}, "
                f"{grid}, rand_noise={self.rand_noise}")
