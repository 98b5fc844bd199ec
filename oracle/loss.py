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
    """State (running task-loss sum, step count, temperature) around oracle/fq_eager.potential_loss; the attribute names
    are the ones the reference's callbacks and loggers read (gdnsq_loss.py:14-30, 73-84)."""

    def __init__(self, criterion, p=1, a=8, w=4, lossless=False):
        super().__init__()
        self.criterion = criterion
        self.p = p
        self.at, self.wt = a, w
        self.lossless = lossless
        self.loss_sum = 0.0     # running sum of the task loss: calibrates the weight of the hinge
        self.cnt = 1
        self.t = 0.0            # temperature, ramped by TemperatureSchedule
        self.aloss = self.wloss = torch.tensor(1.0)

    def _combine(self, base, las, laq, lws, lwq):
        from . import fq_eager as O
        self.base_loss = base
        total, task = O.potential_loss(base, las, laq, lws, lwq, self.at, self.wt, self.t, self.loss_sum, self.cnt,
                                       lossless=self.lossless, p=self.p)
        if self.training:
            self.loss_sum = self.loss_sum + task.detach()
            self.cnt += 1
        # what the reference logs next to the loss (gdnsq_loss.py:73-84)
        margin_w = ((lwq - lws) - (self.wt - 1e-3)).clamp_min(0).pow(self.p)
        margin_a = ((laq - las) - (self.at - 1e-3)).clamp_min(0).pow(self.p)
        self.wloss, self.aloss, self.rloss = margin_w.mean(), margin_a.mean(), task
        self.s_weight_loss, self.q_weight_loss = -lws.mean(), lwq.mean()
        self.s_act_loss, self.q_act_loss = -las.mean(), laq.mean()
        self.weight_reg_loss = (lwq - lws).max()
        return total


class PotentialLoss(_Potential):
    def forward(self, output, target):
        prd, las, laq, lws, lwq = output
        return self._combine(self.criterion(prd, target), las, laq, lws, lwq)


class PotentialLossNoPred(_Potential):
    def forward(self, output):
        bloss, las, laq, lws, lwq = output
        return self._combine(bloss, las, laq, lws, lwq)


LOSS_CLASSES = (PotentialLoss, PotentialLossNoPred)      # QATTrainer(loss_classes=...)
