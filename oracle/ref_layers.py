"""CPU eager layer modules over oracle/fq_eager.py  --  TEST INFRASTRUCTURE ONLY.

Same constructor keywords as the reference's NoisyAct / NoisyConv2d / NoisyLinear
(layers/gdnsq_act.py:10-18, gdnsq_conv2d.py:14-32, gdnsq_linear.py:14-25) so that
mhaq_amd.wrap.quantize_model(layers=ORACLE_LAYERS) builds the reference-equivalent CPU model
that bench.py times as `cpu_baseline` (kind "port") and tests/ use as the model-level checker.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from . import fq_eager as O


class NoisyAct(nn.Module):
    def __init__(self, init_s=-10, init_q=10, signed=True, noise_ratio=1, disable=False, qnmethod="STE"):
        super().__init__()
        self.disable, self.signed = disable, signed
        zp = 0.0 if not signed else -float(2.0 ** (init_q - 1))
        self.log_act_q = nn.Parameter(torch.tensor([float(init_q)]))
        self.act_b = nn.Parameter(torch.tensor([zp]), requires_grad=bool(signed))
        self.log_act_s = nn.Parameter(torch.tensor([float(init_s)]))
        self.qnmethod = O._method_name(qnmethod)
        self.bw = torch.tensor(0.0)

    def forward(self, x):
        if self.disable:
            return x
        y, q = O.act_fake_quant(x, self.log_act_s, self.log_act_q, self.act_b, method=self.qnmethod)
        if not self.training:
            s, qr = torch.exp2(self.log_act_s), torch.exp2(self.log_act_q)
            O.check_integrity(q, s, self.act_b, self.act_b, self.act_b + qr - s)
            self.bw = O.act_bit_width(q)
        return y


class _WeightMixin:
    def _init_q(self, out_channels, qscheme, log_s_init, qnmethod, with_log_b_s=True):
        self.per_channel = getattr(qscheme, "value", qscheme) == 1
        shape = (out_channels, 1, 1, 1) if self.per_channel else (1,)
        self.log_wght_s = nn.Parameter(torch.full(shape, float(log_s_init)))
        if self.per_channel and with_log_b_s:
            self.log_b_s = nn.Parameter(torch.full((1,), float(log_s_init)))
        self._noise_ratio = nn.Parameter(torch.ones(1), requires_grad=False)
        self.qnmethod = O._method_name(qnmethod)

    def _wq(self):
        ls = self.log_wght_s
        if self.per_channel and self.weight.dim() != 4:
            ls = ls.reshape([self.weight.shape[0]] + [1] * (self.weight.dim() - 1))
        return O.weight_fake_quant(self.weight, ls, self.per_channel, self.qnmethod)[0]


class NoisyConv2d(nn.Conv2d, _WeightMixin):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, padding_mode="zeros", device=None, dtype=None, qscheme=0, log_s_init=-12,
                 rand_noise=False, quant_bias=False, qnmethod="AEWGS"):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         padding_mode, device, dtype)
        self._init_q(out_channels, qscheme, log_s_init, qnmethod)
        self.quant_bias = quant_bias

    def forward(self, x):
        bias = self.bias
        if self.quant_bias:
            bias = O.bias_fake_quant(self.bias, self.weight, self.log_wght_s, self.qnmethod)
        return self._conv_forward(x, self._wq(), bias)


class NoisyLinear(nn.Linear, _WeightMixin):
    def __init__(self, in_features, out_features, bias=True, device=None, dtype=None, qscheme=0,
                 log_s_init=-12, rand_noise=False, qnmethod="STE"):
        super().__init__(in_features, out_features, bias, device, dtype)
        self._init_q(out_features, qscheme, log_s_init, qnmethod, with_log_b_s=False)

    def forward(self, x):
        return F.linear(x, self._wq(), self.bias)


ORACLE_LAYERS = (NoisyAct, NoisyConv2d, NoisyLinear)
