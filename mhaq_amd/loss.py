"""Loss wrappers the QAT step needs, restated from the reference:

  PotentialLoss        task loss + bit-width hinge, prediction/target form
                       (/root/reference/src/quantization/gdnsq/gdnsq_loss.py:6-86)
  PotentialLossNoPred  same with a precomputed base loss (gdnsq_loss.py:88-168)
  FusedPotentialLoss / FusedPotentialLossNoPred
                       the same two with the hinge arithmetic in the HIP library (GPU trainer)
  SymmetricalKL        distillation loss of the ResNet-18 configs (src/aux/loss/symm_kl_loss.py)
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn


class _Potential(nn.Module):
    def __init__(self, criterion, p=1, a=8, w=4, lossless=False):
        super().__init__()
        self.criterion = criterion
        self.p = p
        self.at, self.wt = a, w
        self.lossless = lossless
        self.l_eps = 1e-3
        self.loss_sum = 0.0     # running sum of the task loss (calibrates the hinge weight)
        self.cnt = 1
        self.t = 0.0            # temperature, ramped by TemperatureSchedule
        self.aloss = self.wloss = torch.tensor(1.0)

    def _combine(self, base, las, laq, lws, lwq):
        self.base_loss = base
        zero = torch.zeros((), device=lws.device)
        wloss0 = torch.max(zero, (lwq - lws) - (self.wt - self.l_eps)).pow(self.p)
        wloss, wact = wloss0.mean(), (wloss0 > 0).sum()
        aloss0 = torch.max(zero, (laq - las) - (self.at - self.l_eps)).pow(self.p)
        aloss, aact = aloss0.mean(), (aloss0 > 0).sum()
        rloss = base.pow(self.p)
        calib_mul = self.loss_sum / self.cnt
        wmul = (wact + self.l_eps) / (wact + aact + self.l_eps)
        amul = (aact + self.l_eps) / (wact + aact + self.l_eps)
        l1, l2 = (1.0, self.t) if self.lossless else (self.t, 1.0)
        ploss = calib_mul * l1 * (wmul * wloss + amul * aloss) + l2 * rloss
        if self.training:
            self.loss_sum = self.loss_sum + rloss.detach()
            self.cnt += 1
        self.wloss, self.aloss, self.rloss = wloss, aloss, rloss
        self.s_weight_loss, self.q_weight_loss = -lws.mean(), lwq.mean()
        self.s_act_loss, self.q_act_loss = -las.mean(), laq.mean()
        self.weight_reg_loss = (lwq - lws).max()
        return ploss


class PotentialLoss(_Potential):
    def forward(self, output, target):
        prd, las, laq, lws, lwq = output
        return self._combine(self.criterion(prd, target), las, laq, lws, lwq)


class PotentialLossNoPred(_Potential):
    def forward(self, output):
        bloss, las, laq, lws, lwq = output
        return self._combine(bloss, las, laq, lws, lwq)


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
