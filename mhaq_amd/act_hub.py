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

The node, the workspaces and the descriptor tables live in the compiled binding (csrc/torch_binding.cpp: Hub, HubFn):
the whole backward side runs on the autograd thread without the GIL.  Retention: a descriptor table or workspace that
was handed out while a hipGraph capture was active is held (the captured launches have its address baked in) until
`release_captured()` -- the capturing trainer calls it when it drops its graph; everything else is ordinary
caching-allocator memory: eager descriptor tables sit in a 4-entry LRU, an outgrown eager workspace is dropped.  A
loop over many batch shapes therefore holds a bounded number of tables and exactly one workspace per quantizer.
"""
from __future__ import annotations

import torch

from ._ext import ext as _ext


class HubRef:
    """What a NoisyAct keeps of its hub: (hub, slot).  Copies and pickles of the module come out detached
    (hub None): the hub's workspaces never travel with torch.save(model) / copy.deepcopy(model)."""
    __slots__ = ("hub", "slot")

    def __init__(self, hub=None, slot=0):
        self.hub, self.slot = hub, slot

    def __deepcopy__(self, memo):
        return HubRef()

    def __reduce__(self):
        return (HubRef, ())

    def __iter__(self):                      # (hub, slot) unpacking, as ops.fake_quant_act_layer(hub_slot=...) takes it
        return iter((self.hub, self.slot))

    def __getitem__(self, i):
        return (self.hub, self.slot)[i]


class ActGradHub:
    """Collects the NoisyAct quantizers of `model` that run on the fused layer path (STE / LSQ / EWGS)."""

    def __init__(self, model: torch.nn.Module):
        from .layers import NoisyAct
        self.acts = [m for m in model.modules() if isinstance(m, NoisyAct) and not m.disable]
        self.id = _ext().hub_create(len(self.acts))
        for i, a in enumerate(self.acts):
            a._hub = HubRef(self, i)
        self._outs = None            # this step's aliases of (log_act_s, log_act_q, act_b) per quantizer
        self._taken = []
        self._params = None

    def __len__(self):
        return len(self.acts)

    def __del__(self):
        try:
            _ext().hub_destroy(self.id)
        except Exception:             # interpreter shutdown
            pass

    def __deepcopy__(self, memo):      # a hub belongs to the modules it was built over
        raise TypeError("ActGradHub cannot be copied: build one over the copied model")

    # -- forward side -------------------------------------------------------------------------
    def begin(self) -> None:
        """Start of a training step (grad mode on): route the parameters through the joint-finalize node."""
        if not torch.is_grad_enabled() or not self.acts:
            _ext().hub_clear_pending(self.id)
            self._outs = None
            return
        if self._params is None or self._params[0] is not self.acts[0].log_act_s:
            self._params = [p for a in self.acts for p in (a.log_act_s, a.log_act_q, a.act_b)]
        self._outs = _ext().hub_begin(self.id, self._params)
        self._taken = [False] * len(self.acts)

    def end(self) -> None:
        self._outs = None

    def take(self, slot: int):
        """The routed parameters of quantizer `slot`, once per step (a module called twice in one forward keeps
        the immediate finalize for its second call: its workspace holds one set of partials)."""
        outs = self._outs
        if outs is None or self._taken[slot] or not torch.is_grad_enabled():
            return None
        self._taken[slot] = True
        k = 3 * slot
        return outs[k], outs[k + 1], outs[k + 2]

    # -- retention ------------------------------------------------------------------------------
    def state(self) -> dict:
        """{'tables', 'captured_tables', 'retired', 'pending', 'has_table', 'workspace_bytes'} of the C++ hub."""
        return _ext().hub_state(self.id)

    def release_captured(self) -> None:
        """Drop the descriptor tables and outgrown workspaces held for captured hipGraphs (call when the graphs that
        baked their addresses in are gone)."""
        _ext().hub_release_captured(self.id)

    def refresh_parameters(self) -> None:
        """After a quantizer's parameters were REPLACED (not filled in place): route the new tensors."""
        self._params = None
