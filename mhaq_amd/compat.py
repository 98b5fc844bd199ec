"""Import-path compatibility: make the reference's own modules resolve the hot-path classes to this package.

    import mhaq_amd.compat; mhaq_amd.compat.install()

registers stand-ins for the four modules the rest of MHAQ imports the path from,
    src.quantization.gdnsq.gdnsq              (Quantizer, QNoise, QNSTE, QNLSQ, QNEWGS, QNAEWGS, reduce_to_shape)
    src.quantization.gdnsq.layers.gdnsq_act   (NoisyAct)
    src.quantization.gdnsq.layers.gdnsq_conv2d (NoisyConv2d)
    src.quantization.gdnsq.layers.gdnsq_linear (NoisyLinear)
so that `gdnsq_quant.py`, `utils/model_helper.py`, `utils/model_stats.py`, `calib/minmaxobserver.py` pick up the
HIP-backed classes without being edited (INTEGRATION.md section 1 as a one-liner).  `gdnsq_utils.QNMethod` and
`src.aux.types.QScheme` stay the reference's own enums: the ops accept any enum with the same member names.
Call it before the reference's modules are imported.
"""
from __future__ import annotations

import sys
import types

import torch


def reduce_to_shape(t: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """gdnsq.py:150-152."""
    dims = [i for i, size in enumerate(like.shape) if size == 1]
    return torch.mean(t, dim=tuple(dims), keepdim=True)


def install() -> None:
    from . import layers, ops_generic
    from .gdnsq import Quantizer

    def module(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__mhaq_amd_compat__ = True
        sys.modules[name] = m
        parent_name, _, leaf = name.rpartition(".")
        try:                                   # bind as an attribute of the parent package when it is importable
            import importlib
            setattr(importlib.import_module(parent_name), leaf, m)
        except ImportError:
            pass
        return m

    module("src.quantization.gdnsq.gdnsq", Quantizer=Quantizer, QNoise=ops_generic.QNoise,
           QNSTE=ops_generic.QNSTE, QNLSQ=ops_generic.QNLSQ, QNEWGS=ops_generic.QNEWGS,
           QNAEWGS=ops_generic.QNAEWGS, reduce_to_shape=reduce_to_shape, scaled_noise=ops_generic.scaled_noise)
    module("src.quantization.gdnsq.layers.gdnsq_act", NoisyAct=layers.NoisyAct)
    module("src.quantization.gdnsq.layers.gdnsq_conv2d", NoisyConv2d=layers.NoisyConv2d)
    module("src.quantization.gdnsq.layers.gdnsq_linear", NoisyLinear=layers.NoisyLinear)
