"""Loss wrappers the QAT step needs, restated from the reference:

  PotentialLoss        task loss + bit-width hinge, prediction/target form
                       (/root/reference/src/quantization/gdnsq/gdnsq_loss.py:6-86)
  PotentialLossNoPred  same with a precomputed base loss (gdnsq_loss.py:88-168)
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


class SymmetricalKL(nn.Module):
    def forward(self, input, target):
        x = F.log_softmax(input, dim=1)
        y = F.log_softmax(target, dim=1)
        return (F.kl_div(x, y, log_target=True, reduction="batchmean")
                + F.kl_div(y, x, log_target=True, reduction="batchmean"))
