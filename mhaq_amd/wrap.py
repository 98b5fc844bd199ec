"""The reference's layer-replacement rule and regulariser-input gathering, restated for any
nn.Module (the reference needs a LightningModule):

  quantize_model    GDNSQQuant.quantize / _quantize_module / _get_quantization_sequence
                    (/root/reference/src/quantization/gdnsq/gdnsq_quant.py:68-146, 483-545,
                     candidates: src/quantization/abc/abc_quant.py:88-113)
  get_model_values  ModelHelper.get_model_values (gdnsq/utils/model_helper.py:13-76)

`layers=(Act, Conv2d, Linear)` selects the layer classes: the product passes the HIP-backed
mhaq_amd.layers; bench.py's CPU baseline passes the oracle's eager layers.
"""
from __future__ import annotations

from collections import OrderedDict
from operator import attrgetter

import torch
from torch import nn

from .enums import QNMethod, QScheme


def _set_by_name(root: nn.Module, dotted: str, new: nn.Module) -> None:
    parent_name, _, leaf = dotted.rpartition(".")
    parent = attrgetter(parent_name)(root) if parent_name else root
    setattr(parent, leaf, new)


def quantizable_layers(model: nn.Module, excluded_layers=()):
    cands = OrderedDict((n, m) for n, m in model.named_modules() if isinstance(m, (nn.Conv2d, nn.Linear)))
    for name in excluded_layers:
        if name in cands:
            cands.pop(name)
        else:
            raise AttributeError(f"Layer name {name} is not found in the model.")
    return cands


def fuse_conv_bn(model: nn.Module, conv_name: str, bn_name: str) -> None:
    """GDNSQQuant.fuse_conv_bn (gdnsq_quant.py:163-186): fold the FOLLOWING BatchNorm's running statistics and
    affine parameters into the convolution (W * gamma/std per output channel, bias = beta + (b - mu) * gamma/std)
    and replace the BatchNorm by nn.Identity -- so the weight the quantizer sees is the folded one."""
    conv = attrgetter(conv_name)(model)
    bn = attrgetter(bn_name)(model)
    W = conv.weight.clone()
    b = conv.bias.clone() if conv.bias is not None else torch.zeros(conv.out_channels, device=W.device)
    std = torch.sqrt(bn.running_var + bn.eps)
    scale = bn.weight / std
    conv.weight.data = W * scale.view([-1] + [1] * (W.dim() - 1))
    conv.bias = nn.Parameter(bn.bias + (b - bn.running_mean) * scale)
    _set_by_name(model, bn_name, nn.Identity())


def freeze_all_batchnorm_layers(model: nn.Module, freeze=True) -> None:
    """GDNSQQuant.freeze_all_batchnorm_layers (gdnsq_quant.py:150-161): eval mode (no running-stat updates) and no
    gradients for every BatchNorm."""
    for m in model.modules():
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.eval()
            m.weight.requires_grad = not freeze
            m.bias.requires_grad = not freeze


def quantize_model(model: nn.Module, qscheme=QScheme.PER_CHANNEL, qnmethod=QNMethod.STE, excluded_layers=(),
                   quantize_bias=False, act_bit=8, layers=None, fuse_batchnorm=False,
                   freeze_batchnorm=False) -> nn.Module:
    """In-place: every non-excluded, non-1x1 nn.Conv2d becomes
    Sequential(activations_quantizer=NoisyAct(signed=?), "0"=NoisyConv2d sharing weight/bias).
    fuse_batchnorm / freeze_batchnorm: the two config.quantization switches of GDNSQQuant.quantize
    (gdnsq_quant.py:129-146; False in every shipped config)."""
    if layers is None:
        from .layers import NoisyAct, NoisyConv2d, NoisyLinear
        layers = (NoisyAct, NoisyConv2d, NoisyLinear)
    Act, Conv, Lin = layers
    qscheme = QScheme(qscheme) if isinstance(qscheme, int) else qscheme
    qnmethod = QNMethod[qnmethod] if isinstance(qnmethod, str) else qnmethod
    names, types = zip(*[(n, type(m)) for n, m in model.named_modules()])  # snapshot before surgery
    for name, module in quantizable_layers(model, excluded_layers).items():
        if module.kernel_size != (1, 1):      # nn.Linear has no kernel_size: AttributeError, as in the reference
            preceding = types[names.index(name) - 1]
            nxt = names.index(name) + 1
            if fuse_batchnorm and nxt < len(names) and issubclass(types[nxt], nn.BatchNorm2d):
                fuse_conv_bn(model, name, names[nxt])
                module = attrgetter(name)(model)          # now carries the folded weight and a bias
            signed = not issubclass(preceding, nn.ReLU)
            has_bias = module.bias is not None
            if isinstance(module, nn.Conv2d):
                q = Conv(module.in_channels, module.out_channels, module.kernel_size, module.stride,
                         module.padding, module.dilation, module.groups, has_bias, module.padding_mode,
                         qscheme=qscheme, log_s_init=-12, quant_bias=quantize_bias, qnmethod=qnmethod)
            elif isinstance(module, nn.Linear):
                q = Lin(module.in_features, module.out_features, has_bias, qscheme=qscheme, log_s_init=-12,
                        qnmethod=qnmethod)
            else:
                raise NotImplementedError(f"Module not supported {type(module)}")
            q.weight = module.weight
            if has_bias:
                q.bias = module.bias
            q.to(module.weight.device)
            seq = nn.Sequential(OrderedDict([
                ("activations_quantizer", Act(signed=signed, disable=(act_bit == -1)).to(module.weight.device)),
                ("0", q),
            ]))
            _set_by_name(model, name, seq)
    if freeze_batchnorm:
        freeze_all_batchnorm_layers(model)
    return model


def _is_act(m):
    return hasattr(m, "log_act_s") and hasattr(m, "log_act_q")


def _is_weight_layer(m):
    return hasattr(m, "log_wght_s") and hasattr(m, "weight")


def get_model_values(model: nn.Module, qscheme=QScheme.PER_TENSOR, modules=None):
    """(log_act_s, log_act_q, log_wght_s, log2(max - min + 2^log_wght_s)) concatenated over layers.
    `modules`: the model's modules in named_modules() order (or just its quantized layers), if the caller has them."""
    qscheme = QScheme(qscheme) if isinstance(qscheme, int) else qscheme
    las, laq, lws, lwq = [], [], [], []
    for m in (modules if modules is not None else model.modules()):
        if _is_weight_layer(m):
            if m.log_wght_s.requires_grad:
                if qscheme == QScheme.PER_CHANNEL:
                    lws.append(m.log_wght_s.ravel())
                    fused = m.regulariser_input() if hasattr(m, "regulariser_input") else None
                    if fused is not None:      # computed by the layer's own forward launch this step
                        lwq.append(fused)
                        continue
                    dims = tuple(range(1, m.weight.dim()))
                    mn, mx = m.weight.amin(dims), m.weight.amax(dims)
                else:
                    lws.append(m.log_wght_s)
                    fused = m.regulariser_input() if hasattr(m, "regulariser_input") else None
                    if fused is not None:
                        lwq.append(fused)
                        continue
                    mn, mx = m.weight.amin(), m.weight.amax()
                lwq.append(torch.log2(mx - mn + torch.exp2(m.log_wght_s.ravel())))
        elif _is_act(m):
            if m.log_act_s.requires_grad:
                laq.append(m.log_act_q)
                las.append(m.log_act_s)
    if qscheme == QScheme.PER_TENSOR:
        return (torch.stack(las).ravel(), torch.stack(laq).ravel(), torch.stack(lws).ravel(),
                torch.stack(lwq).ravel())
    return torch.cat(las), torch.cat(laq), torch.cat(lws), torch.cat(lwq)
