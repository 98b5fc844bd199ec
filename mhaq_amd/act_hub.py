"""One finalize launch for every activation quantizer of a backward pass.

A NoisyAct backward is a streaming launch (mhaq::pt_bwd_kernel, 12 B/elem) plus a latency-bound finalize
that turns its per-block partial sums into dL/dlog_act_s, dL/dlog_act_q, dL/dact_b.  Those three scalars
are only read by the optimizer, so nothing in the backward pass waits for them: with a hub installed, each
quantizer's backward leaves its partials in a workspace the hub keeps alive across steps, and ONE launch
(mhaq_fq_act_bwd_finalize_multi, 3 workgroups per quantizer) reduces them all once the last quantizer's
backward has run.  On the ResNet-18 W4A4 activation set that takes 15 finalize launches (and their two
kernel boundaries each) off the critical path of every step; results are bit-identical to the per-quantizer
finalize (same partition, same order).

Mechanics (plain autograd, no hooks): `begin()` passes the quantizers' parameters through an identity
Function whose backward is the joint finalize.  The layers consume ITS outputs, so autograd runs it after
every quantizer's backward -- which hand it placeholder gradients and record where their partials are.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, ops


class _Desc(C.Structure):          # mhaq_act_finalize_desc
    _fields_ = [("partials", C.c_void_p), ("nparts", C.c_int64)]


class _HubFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hub, *params):
        ctx.hub = hub
        ctx.set_materialize_grads(False)
        return tuple(p.view_as(p) for p in params)      # aliases: the kernels read the parameters in place

    @staticmethod
    def backward(ctx, *grads):
        return (None, *ctx.hub._finalize(grads))


class ActGradHub:
    """Collects the NoisyAct quantizers of `model` that run on the fused layer path (STE / LSQ / EWGS)."""

    def __init__(self, model: torch.nn.Module):
        from .layers import NoisyAct
        self.acts = [m for m in model.modules() if isinstance(m, NoisyAct) and not m.disable]
        for i, a in enumerate(self.acts):
            a._hub = (self, i)
        self._outs = None            # this step's aliases of (log_act_s, log_act_q, act_b) per quantizer
        self._taken = []
        self._pending = []           # (slot, nparts, workspace tensor) in backward order
        self._ws = [None] * len(self.acts)
        # Device memory a captured hipGraph may have baked into its launches must never be freed: descriptor tables
        # are kept per key (one per distinct set of batch shapes), outgrown workspaces are retired, not released.
        self._tables = {}            # key -> (device table, pinned host copy)
        self._table = None           # the table of the last finalize
        self._retired = []

    def __len__(self):
        return len(self.acts)

    # -- forward side -------------------------------------------------------------------------
    def begin(self) -> None:
        """Start of a training step (grad mode on): route the parameters through the joint-finalize node."""
        self._pending.clear()
        if not torch.is_grad_enabled() or not self.acts:
            self._outs = None
            return
        params = [p for a in self.acts for p in (a.log_act_s, a.log_act_q, a.act_b)]
        self._outs = _HubFn.apply(self, *params)
        self._taken = [False] * len(self.acts)

    def end(self) -> None:
        self._outs = None

    def take(self, slot: int):
        """The routed parameters of quantizer `slot`, once per step (a module called twice in one forward keeps
        the immediate finalize for its second call: its workspace holds one set of partials)."""
        if self._outs is None or self._taken[slot] or not torch.is_grad_enabled():
            return None
        self._taken[slot] = True
        return self._outs[3 * slot], self._outs[3 * slot + 1], self._outs[3 * slot + 2]

    # -- backward side ------------------------------------------------------------------------
    def workspace(self, slot: int, nbytes: int, device):
        ws = self._ws[slot]
        if ws is None or ws.numel() < nbytes or ws.device != device:
            if ws is not None:
                self._retired.append(ws)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self._ws[slot] = ws
        return ws

    def record(self, slot: int, nparts: int, ws: torch.Tensor) -> None:
        self._pending.append((slot, int(nparts), ws))

    def _finalize(self, grads):
        """Backward of the identity node: every recorded quantizer's partials -> its three gradients."""
        n = len(self.acts)
        out = [None] * (3 * n)
        pending, self._pending = self._pending, []
        if not pending:
            return out
        dev = pending[0][2].device
        key = tuple((s, k, w.data_ptr()) for s, k, w in pending)
        entry = self._tables.get(key)
        if entry is None:
            arr = (_Desc * len(pending))()
            for j, (_, k, w) in enumerate(pending):
                arr[j] = _Desc(w.data_ptr(), k)
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).pin_memory()
            entry = self._tables[key] = (host.to(dev, non_blocking=True), host)
        self._table = entry[0]
        slab = torch.empty(len(pending), 3, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().mhaq_fq_act_bwd_finalize_multi(self._table.data_ptr(), len(pending), slab.data_ptr(),
                                                            ops._stream()), "mhaq_fq_act_bwd_finalize_multi")
        for j, (s, _, _) in enumerate(pending):
            a = self.acts[s]
            for c, p in enumerate((a.log_act_s, a.log_act_q, a.act_b)):
                if grads[3 * s + c] is not None:          # autograd asked for it (requires_grad + reached)
                    out[3 * s + c] = slab[j, c:c + 1].view(p.shape)
        return out
