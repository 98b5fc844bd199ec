"""PotentialLoss / PotentialLossNoPred as torch modules on the CPU  --  TEST INFRASTRUCTURE ONLY.

The module form of oracle/fq_eager.potential_loss: task loss + bit-width hinge with the running loss_sum / cnt state
and the temperature `t`, attribute names as in /root/reference/src/quantization/gdnsq/gdnsq_loss.py:6-86 (prediction /
target form) and :88-168 (precomputed base loss).  Used by the CPU host-logic tests (the trainer over the oracle's
layers) and by bench.py's `cpu_baseline` leg; the product computes the same numbers in one HIP launch per direction
(mhaq_amd/loss.py: FusedPotentialLoss*).  tests/test_wrap_loss_cpu.py holds these modules to the vectors recorded from
the reference's own classes (tests/golden/model_cases.npz).
"""
from __future__ import annotations

import torch
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


LOSS_CLASSES = (PotentialLoss, PotentialLossNoPred)      # QATTrainer(loss_classes=...)
