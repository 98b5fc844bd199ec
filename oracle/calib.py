"""CPU restatement of the reference's calibration arithmetic  --  TEST INFRASTRUCTURE ONLY (see oracle/fq_eager.py
for the import rule; pinned against tests/golden/calib_cases.npz, which oracle/gen_golden.py records from the
reference's own functions).

  observe                  MinMaxObserver._hook (gdnsq/calib/minmaxobserver.py:26-36): per batch, min and max of the
                           tensor entering a NoisyAct
  mean_stats_activations   apply_mean_stats_activations (minmaxobserver.py:39-66)
  quantile_weights_s       apply_quantile_weights_s (minmaxobserver.py:69-88)

Both `apply_*` functions work on plain state records (one dict per quantizer, in model order) so that the same
restatement checks the reference fixtures on the CPU and the product's calibrate_* on the GPU.

Reference behaviour kept on purpose:
  * `abits = max_bits` / `wbits = max_bits` is an assignment to the loop's own variable (minmaxobserver.py:52-53,
    78-79): once one frozen quantizer has been met, every LATER quantizer is calibrated to max_bits too;
  * a zero-width activation range ("pruned", :63-66) sets log_act_q = log_act_s = 0 and freezes all three parameters;
  * a constant weight channel gives log2(0) = -inf, so max(log_wght_s, -inf) keeps the old value.
"""
from __future__ import annotations

import torch


def observe(batches):
    """batches: iterable of tensors seen by one NoisyAct -> (min over batches, max over batches), 0-dim fp32."""
    mins = torch.stack([torch.min(x) for x in batches])
    maxs = torch.stack([torch.max(x) for x in batches])
    return mins.min(), maxs.max()


def mean_stats_activations(acts, abits=8, max_bits=24):
    """acts: list of dicts {min, max, grad_s, grad_q, grad_b} (0-dim fp32 tensors and the three requires_grad flags)
    in model order.  Returns a list of dicts {log_act_s, log_act_q, act_b, grad_s, grad_q, grad_b}."""
    out = []
    for a in acts:
        mn, mx = a["min"], a["max"]
        if not a["grad_q"] and not a["grad_s"]:
            abits = max_bits                           # persists for the quantizers that follow (reference quirk)
        if mx - mn > 0:
            log_s = torch.log2((mx - mn) / (2 ** abits - 1))
            log_q = log_s + abits
            out.append(dict(log_act_s=log_s.reshape(1), log_act_q=log_q.reshape(1), act_b=mn.reshape(1),
                            grad_s=a["grad_s"], grad_q=a["grad_q"], grad_b=a["grad_b"]))
        else:                                           # pruned
            z = torch.zeros(1)
            out.append(dict(log_act_s=z, log_act_q=z.clone(), act_b=mn.reshape(1).clone(),
                            grad_s=False, grad_q=False, grad_b=False))
    return out


def quantile_weights_s(layers, wbits=8, max_bits=24):
    """layers: list of dicts {weight [Co, ...], log_wght_s [Co,1,1,1], grad} in model order -> list of new
    log_wght_s tensors (per-channel, the only scheme the reference's function supports)."""
    out = []
    for m in layers:
        w = m["weight"]
        dims = tuple(range(1, w.dim()))
        max_, min_ = w.amax(dims), w.amin(dims)
        if not m["grad"]:
            wbits = max_bits                            # persists (reference quirk)
        floor = torch.log2((max_ - min_) / (2 ** wbits - 1)).reshape(m["log_wght_s"].shape)
        out.append(torch.max(m["log_wght_s"], floor))
    return out
