"""Loss wrappers the QAT step needs:

  FusedPotentialLoss        task loss + bit-width hinge, prediction/target form
                            (/root/reference/src/quantization/gdnsq/gdnsq_loss.py:6-86)
  FusedPotentialLossNoPred  same with a precomputed base loss (gdnsq_loss.py:88-168)
                            -- both with the hinge arithmetic in the HIP library (mhaq_fq_potential_loss_fwd / _bwd)
  SymmetricalKL             distillation loss of the ResNet-18 configs (src/aux/loss/symm_kl_loss.py)

(The torch restatement of the two PotentialLoss modules is checker code and lives in oracle/loss.py.)
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn


class _FusedPotential(nn.Module):
    """The same loss with its arithmetic in one HIP launch per direction (ops.potential_loss): ~25 scalar
    launches forward and ~30 backward become 1 + 1.  The module state {loss_sum, cnt, t} lives in a 3-float
    device block that the kernel reads (and, in training mode, advances) itself, so a step captured in a
    hipGraph sees the current values at every replay.  This is what the QAT trainer uses on the GPU; the
    attribute names are the reference's (gdnsq_loss.py:14-30, 73-84)."""

    def __init__(self, criterion, p=1, a=8, w=4, lossless=False):
        super().__init__()
        self.criterion = criterion
        self.p = p
        self.at, self.wt = a, w
        self.lossless = lossless
        self.register_buffer("_state", torch.tensor([0.0, 1.0, 0.0]), persistent=False)   # loss_sum, cnt, t
        self._t = 0.0
        self._stats = None

    # -- state, with the reference's attribute names ---------------------------------------------------
    @property
    def loss_sum(self):
        return self._state[0]

    @loss_sum.setter
    def loss_sum(self, v):
        self._state[0:1].copy_(torch.as_tensor(v, dtype=torch.float32).reshape(1))

    @property
    def cnt(self) -> int:
        return int(round(float(self._state[1])))           # host sync: for logging / tests only

    @cnt.setter
    def cnt(self, v):
        self._state[1:2].fill_(float(v))

    @property
    def t(self) -> float:
        return self._t

    @t.setter
    def t(self, v):
        v = float(v)
        if v != self._t:
            self._state[2:3].fill_(v)      # one tiny launch, only when the temperature moves
        self._t = v

    def _combine(self, base, las, laq, lws, lwq):
        from . import ops
        if self._state.device != lws.device:
            self._state = self._state.to(lws.device)
        self.base_loss = base
        ploss, self._stats = ops.potential_loss(base, las, laq, lws, lwq, self._state, self.at, self.wt, self.p,
                                                self.lossless, self.training)
        return ploss

    def _stat(i):  # noqa: N805 -- attribute views of the stats block of the last forward
        return property(lambda self: torch.tensor(1.0) if self._stats is None else self._stats[i])

    wloss, aloss, rloss = _stat(1), _stat(2), _stat(3)
    s_weight_loss, q_weight_loss, s_act_loss, q_act_loss = _stat(7), _stat(8), _stat(9), _stat(10)
    weight_reg_loss = _stat(11)
    del _stat


class FusedPotentialLoss(_FusedPotential):
    def forward(self, output, target):
        prd, las, laq, lws, lwq = output
        return self._combine(self.criterion(prd, target), las, laq, lws, lwq)


class FusedPotentialLossNoPred(_FusedPotential):
    def forward(self, output):
        bloss, las, laq, lws, lwq = output
        return self._combine(bloss, las, laq, lws, lwq)


class SymmetricalKL(nn.Module):
    def forward(self, input, target):
        x = F.log_softmax(input, dim=1)
        y = F.log_softmax(target, dim=1)
        return (F.kl_div(x, y, log_target=True, reduction="batchmean")
                + F.kl_div(y, x, log_target=True, reduction="batchmean"))
