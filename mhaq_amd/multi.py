"""Multi-tensor weight fake-quant: every PER_CHANNEL NoisyConv2d of a model in ONE launch per direction
(mhaq_fq_wlayer_fwd_multi / mhaq_fq_wlayer_bwd_multi, a device-resident pointer table).

The weights do not depend on the activations, so all 16-33 weight quantizers of a step can run before the
forward pass starts; autograd calls the joint backward once every layer's dL/dwq has arrived.  That trades
16-33 launches per direction for one, and delays the weight gradients to the end of backward -- which is
why the data-parallel trainer keeps the per-layer ops (DDP overlaps the gradient all-reduce with backward)
and uses this path only when asked to (`QATTrainer(..., multi_tensor_weights=True)`, single GPU).

What the data-parallel trainer does use (joint_backward=False): the model-wide FORWARD launch, and -- with
`backward_group_elems` > 0 -- the backward in GROUPS of consecutive layers (mhaq_fq_wlayer_bwd_group).  Groups are
cut from the END of the model, each at least `backward_group_elems` weights (16 MB by default): a group's gradients
leave as soon as its earliest layer's dL/dWq has arrived, so the big late layers still overlap their all-reduce
with the rest of backward (torch DDP itself sends gradients in 25 MB buckets, and its last bucket -- the small
early layers -- only goes out at the end of backward either way), while AEWGS exchanges ONE packed [3, group_co]
message per group instead of one per layer (ResNet-18: 3 exchanges per step instead of 16; the reference: 48).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, ops
from ._ext import ext as _ext
from .enums import QScheme


class _Desc(C.Structure):          # mhaq_wlayer_desc
    _fields_ = [("w", C.c_void_p), ("log_s", C.c_void_p), ("G", C.c_void_p), ("g_lwq", C.c_void_p),
                ("co", C.c_int64), ("row", C.c_int64), ("elem_offset", C.c_int64), ("chan_offset", C.c_int64)]


def _upload(descs, device):
    """Host table -> device bytes (pinned staging + async copy on the current stream)."""
    raw = bytes(descs)
    host = torch.frombuffer(bytearray(raw), dtype=torch.uint8).pin_memory()
    return host.to(device, non_blocking=True), host   # keep `host` alive until the copy has run


class _MultiWeightFn(torch.autograd.Function):
    @staticmethod
    @ops._on_device
    def forward(ctx, plan, *tensors):
        L = _lib.lib()
        n = plan.nlayers
        ws, lss = tensors[:n], tensors[n:]
        dev = ws[0].device
        arr = (_Desc * n)()
        for i in range(n):
            arr[i] = _Desc(ws[i].data_ptr(), lss[i].data_ptr(), None, None, plan.co[i], plan.row[i],
                           plan.elem_off[i], plan.chan_off[i])
        table, keep = _upload(arr, dev)
        wq_all = torch.empty(plan.total_elems, dtype=torch.float32, device=dev)
        aux_all = torch.empty(4, plan.total_co, dtype=torch.float32, device=dev)
        _lib.check(L.mhaq_fq_wlayer_fwd_multi(table.data_ptr(), n, plan.total_co, plan.max_row, wq_all.data_ptr(),
                                              aux_all.data_ptr(), ops._stream()), "mhaq_fq_wlayer_fwd_multi")
        ctx.plan, ctx.keep = plan, (table, keep)
        ctx.save_for_backward(aux_all, *ws, *lss)
        ctx.set_materialize_grads(False)
        outs = []
        for i in range(n):      # per-layer views of the two slabs
            outs.append(wq_all[plan.elem_off[i]:plan.elem_off[i] + plan.co[i] * plan.row[i]].view(plan.shape[i]))
        for i in range(n):
            outs.append(aux_all[3, plan.chan_off[i]:plan.chan_off[i] + plan.co[i]])
        ctx.mark_non_differentiable(aux_all)
        return (aux_all, *outs)

    @staticmethod
    def backward(ctx, _gaux, *grads):
        L = _lib.lib()
        plan = ctx.plan
        n = plan.nlayers
        saved = ctx.saved_tensors
        aux_all, ws, lss = saved[0], saved[1:1 + n], saved[1 + n:]
        dev = aux_all.device
        Gs = [torch.zeros_like(ws[i]) if grads[i] is None else grads[i].contiguous() for i in range(n)]
        gl = [None if grads[n + i] is None else grads[n + i].contiguous() for i in range(n)]
        arr = (_Desc * n)()
        for i in range(n):
            arr[i] = _Desc(ws[i].data_ptr(), lss[i].data_ptr(), Gs[i].data_ptr(),
                           gl[i].data_ptr() if gl[i] is not None else None, plan.co[i], plan.row[i],
                           plan.elem_off[i], plan.chan_off[i])
        table, keep = _upload(arr, dev)
        gw_all = torch.empty(plan.total_elems, dtype=torch.float32, device=dev)
        gls_all = torch.empty(plan.total_co, dtype=torch.float32, device=dev)
        _, seed, offset, odev = ops._signs(None, plan.method, aux_all)
        _lib.check(L.mhaq_fq_wlayer_bwd_multi(table.data_ptr(), n, plan.total_co, plan.max_row, aux_all.data_ptr(),
                                              gw_all.data_ptr(), gls_all.data_ptr(), plan.method, None, seed,
                                              offset, odev, ops._stream()), "mhaq_fq_wlayer_bwd_multi")
        ctx.keep_bwd = (table, keep, Gs, gl)
        gws = [gw_all[plan.elem_off[i]:plan.elem_off[i] + plan.co[i] * plan.row[i]].view(plan.shape[i])
               for i in range(n)]
        glss = [gls_all[plan.chan_off[i]:plan.chan_off[i] + plan.co[i]].view(lss[i].shape) for i in range(n)]
        return (None, *gws, *glss)


def backward_groups(sizes, methods, min_elems):
    """[first, last) layer ranges whose backward is one launch: cut from the END of the model, each group at least
    `min_elems` weights (the last-cut one, at the front of the model, takes what is left), never mixing estimators.
    A layer left alone keeps its own (register-resident) launch and is not listed."""
    out, last, acc = [], len(sizes), 0
    for i in reversed(range(len(sizes))):
        acc += sizes[i]
        if acc >= min_elems or i == 0 or methods[i - 1] != methods[i]:
            if last - i > 1:
                out.append((i, last))
            last, acc = i, 0
    return out


class _PoolView:
    """Read-only picture of a group's descriptor-table pool in the compiled binding (torch_binding.cpp: TablePool):
    device tables + pinned staging buffers allocated up front (nothing may be allocated inside a hipGraph capture, and
    a captured upload re-reads its staging buffer at every replay -- so an entry filled during a capture is held
    until release_captured()); eager entries are recycled least-recently-used, after their previous upload has run."""

    def __init__(self, state):
        self.keys = [object() if i < state["used"] else None for i in range(state["size"])]
        self.held = [i < state["held"] for i in range(state["size"])]


class _WeightGroup:
    """Consecutive per-channel layers whose backward is ONE launch (plus, for AEWGS under data parallelism, one
    statistics launch and ONE packed all-reduce).  The forward is the model-wide launch of this step; the group's
    autograd node (compiled: torch_binding.cpp WeightGroupFn) is created when its first layer runs, so autograd
    schedules it the moment that layer's dL/dWq -- the last of the group to arrive -- is there."""

    def __init__(self, plan, index, first, last):
        self.plan, self.index, self.first, self.n = plan, index, first, last - first
        self.idx = list(range(first, last))
        self.chan0, self.elem0 = plan.chan_off[first], plan.elem_off[first]
        self.co = sum(plan.co[i] for i in self.idx)
        self.elems = sum(plan.co[i] * plan.row[i] for i in self.idx)
        self.max_row = max(plan.row[i] for i in self.idx)
        self.method = plan.methods[first]
        self.outs = None

    @property
    def pool(self):
        return _PoolView(_ext().plan_state(self.plan.plan_id)["pools"][self.index])

    def take(self, i):
        """(wq, lwq) of layer `i` of the plan as outputs of the group's autograd node."""
        if self.outs is None:
            p = self.plan
            ws = [p.layers[j].weight for j in self.idx]
            lss = [p.layers[j].log_wght_s for j in self.idx]
            ops.rng.ensure_seeded()
            self.outs = _ext().plan_group_apply(p.plan_id, self.index, ws, lss, ops._sync_dist_state())
        k = i - self.first
        return self.outs[k], self.outs[self.n + k]


class MultiTensorWeightQuant:
    """Plan + driver.  `run()` quantizes every per-channel layer's weight in one launch and parks the
    results on the layers; each NoisyConv2d.forward of this step then just picks its slice up."""

    # the longest per-channel row the model-wide grids keep in registers: 8 float4 per thread at 256 threads (fq_pc.hip
    # multi_reg_nv).  A longer row would walk its data two or three times out of L2 inside the grid; on its own the layer
    # gets a 1024-thread register-resident launch -- a [64, 25088] Linear: forward 7.1 us / backward 12.4 us on its own
    # against ~17 / ~24 us inside the launches of VGG-16's convolutions (profiles/r05_ab_logs.txt) -- so such layers stay
    # out of the plan (`long_rows=True` keeps them in: the C ABI serves any row length, and the tests say so)
    # Sign streams: a planned layer's signs sit at its element offset inside the PLAN's group, so leaving a layer out shifts
    # the offsets of the planned layers behind it and gives the excluded layer a stream of its own (one more draw per
    # backward): the same distribution, other bits than a long_rows=True plan of the same model -- pinned per layer by
    # tests/test_gpu_weight_groups.py::test_default_plan_next_to_excluded_long_row_layers_keeps_the_per_layer_bits
    MAX_PLAN_ROW = 8192

    def __init__(self, model: torch.nn.Module, joint_backward: bool = True, backward_group_elems: int = 0,
                 long_rows: bool = False):
        """backward_group_elems > 0 (with joint_backward=False): the backward runs in groups of consecutive layers
        of at least that many weights each, cut from the end of the model (see the module docstring).
        joint_backward=True: one launch per direction (single GPU: every weight gradient arrives at the end of
        backward).  False: only the FORWARD is batched -- the weights do not depend on the activations, so one launch
        quantizes them all before the forward pass starts -- and the backward stays per layer (or per group of layers), which is
        what data-parallel training needs (gradient overlap, the AEWGS statistics exchange)."""
        from .layers import NoisyConv2d, NoisyLinear
        self.joint_backward = bool(joint_backward)

        def batched(m):
            if isinstance(m, NoisyLinear):          # same weight path (layers._WeightQuantMixin): row = in_features
                if self.joint_backward:
                    return False
            elif not isinstance(m, NoisyConv2d) or m.quant_bias:
                return False
            if m.qscheme == QScheme.PER_CHANNEL:
                return long_rows or m.weight[0].numel() <= self.MAX_PLAN_ROW
            # A PER_TENSOR layer that fits one workgroup is one "channel" whose row is the whole tensor: same minimum,
            # same quantizer, same sums, so the per-channel grids serve it as co = 1 (forward-only / grouped mode;
            # AEWGS keeps its own path: its statistics are per position for a [1]-shaped scale, gdnsq.py:150-152)
            return (not self.joint_backward and m.weight.is_cuda
                    and ops.small_pt_layer_supported(m.weight, m.Q.qnmethod))
        self.layers = [m for m in model.modules() if batched(m)]
        if not self.layers:
            raise ValueError("no NoisyConv2d layers to batch")
        methods = {ops._method_value(m.Q.qnmethod) for m in self.layers}
        if len(methods) != 1 and self.joint_backward:
            raise ValueError("all batched layers must use the same estimator")
        self.method = methods.pop()
        self.methods = [ops._method_value(m.Q.qnmethod) for m in self.layers]
        self._joint_tables = {}
        self.nlayers = len(self.layers)
        self.shape = [tuple(m.weight.shape) for m in self.layers]
        self.per_tensor = [m.qscheme != QScheme.PER_CHANNEL for m in self.layers]
        self.co = [1 if pt else s[0] for s, pt in zip(self.shape, self.per_tensor)]
        self.row = [int(torch.Size(s).numel()) // co for s, co in zip(self.shape, self.co)]
        self.elem_off, self.chan_off = [], []
        e = c = 0
        for co, row in zip(self.co, self.row):
            self.elem_off.append(e)
            self.chan_off.append(c)
            e += co * row
            c += co
        self.total_elems, self.total_co, self.max_row = e, c, max(self.row)
        # backward groups, cut from the end of the model; a group never mixes estimators
        self.groups, self.group_of = [], [None] * self.nlayers
        self.cur_wq = self.cur_aux = None
        self.plan_id = None
        if not self.joint_backward:
            ranges = []
            if backward_group_elems > 0:
                sizes = [co * row for co, row in zip(self.co, self.row)]
                ranges = backward_groups(sizes, self.methods, backward_group_elems)
            for k, (first, last) in enumerate(ranges):
                g = _WeightGroup(self, k, first, last)
                self.groups.append(g)
                for j in range(first, last):
                    self.group_of[j] = g
            # the host side of the model-wide forward and of the grouped backward lives in the compiled binding
            self.plan_id = _ext().plan_create(self.co, self.row, self.methods, ranges)
        self._shapes4 = [[co] + [1] * (len(shp) - 1) for co, shp in zip(self.co, self.shape)]

    def __del__(self):
        try:
            if self.plan_id is not None:
                _ext().plan_destroy(self.plan_id)
        except Exception:             # interpreter shutdown
            pass

    @property
    def _tables(self):
        """One entry per uploaded forward pointer table (joint mode: the dict of device tables itself)."""
        if self.plan_id is None:
            return self._joint_tables
        return [None] * _ext().plan_state(self.plan_id)["fwd_tables"]

    def release_captured(self) -> None:
        """Drop the descriptor tables held for captured hipGraphs (call when those graphs are gone)."""
        if self.plan_id is not None:
            _ext().plan_release_captured(self.plan_id)

    @torch.no_grad()
    def _run_forward_only(self):
        layers = self.layers
        ws = [m.weight for m in layers]
        lss = [m.log_wght_s for m in layers]
        try:
            wq_all, aux_all, per = _ext().plan_forward(self.plan_id, ws, lss)
        except RuntimeError:
            # not the dense float32 device tensors the launch takes: let the argument checks name the problem
            for m in layers:
                ops._require_cuda_f32(m.weight, "weight", any_dense_layout=True)
                ops._require_cuda_f32(m.log_wght_s, "log_wght_s")
            raise
        self.cur_wq, self.cur_aux = wq_all, aux_all
        for g in self.groups:
            g.outs = None
        group_of = self.group_of
        for i, m in enumerate(layers):
            g = group_of[i]
            w = ws[i]
            m.__dict__["_pre_fwd"] = (per[i], (w._version, lss[i]._version, w.data_ptr()),
                                      None if g is None else (g, i))

    def run(self):
        if not self.joint_backward:
            return self._run_forward_only()
        ws = [ops._require_cuda_f32(m.weight, "weight") for m in self.layers]
        lss = [ops._require_cuda_f32(m.log_wght_s, "log_wght_s") for m in self.layers]
        out = _MultiWeightFn.apply(self, *ws, *lss)
        aux_all, wqs, lwqs = out[0], out[1:1 + self.nlayers], out[1 + self.nlayers:]
        for i, m in enumerate(self.layers):
            sl = slice(self.chan_off[i], self.chan_off[i] + self.co[i])
            shp = [self.co[i]] + [1] * (len(self.shape[i]) - 1)
            m.__dict__["_precomputed"] = (wqs[i], aux_all[1, sl].view(shp), aux_all[0, sl].view(shp), lwqs[i],
                                          (m.weight._version, m.log_wght_s._version, torch.is_grad_enabled()))
        return wqs
