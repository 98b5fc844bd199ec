"""mhaq_amd -- MI355X-native fake-quantization path for MHAQ (gfx950 HIP kernels behind a C ABI).

Public surface mirrors the reference's operator seam:
    Quantizer, NoisyAct, NoisyConv2d, NoisyLinear, QNMethod, QScheme
"""
from .enums import QNMethod, QScheme  # noqa: F401
from .gdnsq import Quantizer  # noqa: F401
from .layers import NoisyAct, NoisyConv2d, NoisyLinear  # noqa: F401
