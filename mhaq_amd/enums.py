"""Selector enums of the fake-quant path.

Values mirror the reference so YAML configs / checkpoints keep their meaning:
  * QNMethod  -- /root/reference/src/quantization/gdnsq/gdnsq_utils.py:9-13
  * QScheme   -- /root/reference/src/aux/types.py:19-21
The integer value of QNMethod is also the `method` argument of the C-ABI
(include/mhaq_fq.h).
"""
from enum import Enum


class QNMethod(Enum):
    STE = 0
    EWGS = 1
    AEWGS = 2
    LSQ = 3


class QScheme(Enum):
    PER_TENSOR = 0
    PER_CHANNEL = 1
