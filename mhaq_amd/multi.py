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
from .enums import QNMethod, QScheme


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


class _TablePool:
    """Device descriptor tables for launches whose pointers (the dL/dWq tensors autograd hands over) are only
    known at backward time.  Both the device tables and their pinned staging buffers are allocated up front:
    inside a hipGraph capture nothing may be allocated, and a captured upload (a memcpy node from pinned memory)
    re-reads its staging buffer at every replay -- so an entry filled during a capture is never reused."""

    def __init__(self, nbytes, device, size=8):
        self.dev = [torch.empty(nbytes, dtype=torch.uint8, device=device) for _ in range(size)]
        self.host = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(size)]
        self.keys = [None] * size
        self.held = [False] * size          # baked into a captured graph
        self.events = [None] * size         # last eager upload from this staging buffer
        self.stamp = [0] * size
        self.clock = 0

    def get(self, key, fill):
        """The device table for `key`; `fill()` -> the descriptor bytes when it has to be uploaded."""
        self.clock += 1
        capturing = torch.cuda.is_current_stream_capturing()
        for i, k in enumerate(self.keys):
            if k == key and (self.held[i] or not capturing):
                self.stamp[i] = self.clock
                return self.dev[i]
        free = [i for i, k in enumerate(self.keys) if k is None]
        if free:
            i = free[0]
        elif capturing:
            # (waiting on an eager upload's event is not a capturable call: a capture only takes unused entries)
            raise _lib.MhaqFqError("weight-group descriptor tables: no unused entry left for a captured launch")
        else:
            cand = [i for i in range(len(self.keys)) if not self.held[i]]
            if not cand:
                raise _lib.MhaqFqError("weight-group descriptor tables exhausted by captured graphs")
            i = min(cand, key=lambda j: self.stamp[j])
            if self.events[i] is not None:
                self.events[i].synchronize()    # the staging buffer's previous upload must have run before it changes
        raw = fill()
        self.host[i][:len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
        self.dev[i].copy_(self.host[i], non_blocking=True)
        self.keys[i], self.held[i], self.stamp[i] = key, capturing, self.clock
        if capturing:
            self.events[i] = None
        else:
            ev = torch.cuda.Event()
            ev.record()
            self.events[i] = ev
        return self.dev[i]


class _WeightGroup:
    """Consecutive per-channel layers whose backward is ONE launch (plus, for AEWGS under data parallelism, one
    statistics launch and ONE packed all-reduce).  The forward is the model-wide launch of this step; the group's
    autograd node is created when its first layer runs, so autograd schedules it the moment that layer's dL/dWq --
    the last of the group to arrive -- is there."""

    def __init__(self, plan, first, last):
        self.plan, self.first, self.n = plan, first, last - first
        self.idx = list(range(first, last))
        self.chan0, self.elem0 = plan.chan_off[first], plan.elem_off[first]
        self.co = sum(plan.co[i] for i in self.idx)
        self.elems = sum(plan.co[i] * plan.row[i] for i in self.idx)
        self.max_row = max(plan.row[i] for i in self.idx)
        self.method = plan.methods[first]
        self.outs = None
        self.pool = None

    def take(self, i):
        """(wq, lwq) of layer `i` of the plan as outputs of the group's autograd node."""
        if self.outs is None:
            p = self.plan
            ws = [p.layers[j].weight for j in self.idx]
            lss = [p.layers[j].log_wght_s for j in self.idx]
            self.outs = _WeightGroupFn.apply(self, *ws, *lss)
        k = i - self.first
        return self.outs[k], self.outs[self.n + k]


class _WeightGroupFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grp, *tensors):
        p, n = grp.plan, grp.n
        ws = tensors[:n]
        wq_all, aux_all = p.cur_wq, p.cur_aux     # this step's model-wide forward (not autograd inputs)
        ctx.grp, ctx.aux_all = grp, aux_all
        ctx.ls_shapes = [t.shape for t in tensors[n:]]
        ctx.save_for_backward(*ws)
        ctx.set_materialize_grads(False)
        outs = []
        for k, i in enumerate(grp.idx):     # no launch here: slices of the model-wide forward of this step
            flat = wq_all[p.elem_off[i]:p.elem_off[i] + p.co[i] * p.row[i]]
            outs.append(torch.as_strided(flat, ws[k].shape, ws[k].stride()))
        for i in grp.idx:
            outs.append(aux_all[3, p.chan_off[i]:p.chan_off[i] + p.co[i]])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        L = _lib.lib()
        grp = ctx.grp
        p, n = grp.plan, grp.n
        ws = ctx.saved_tensors
        aux_all = ctx.aux_all
        dev = aux_all.device
        Gs = [torch.zeros_like(ws[k]) if grads[k] is None else ops._like_layout(grads[k], ws[k]) for k in range(n)]
        gl = [None if grads[n + k] is None else grads[n + k].contiguous() for k in range(n)]
        if grp.pool is None:
            grp.pool = _TablePool(C.sizeof(_Desc) * n, dev)
        key = tuple(t.data_ptr() for t in (*ws, *Gs)) + tuple(0 if g is None else g.data_ptr() for g in gl)

        def fill():
            arr = (_Desc * n)()
            for k, i in enumerate(grp.idx):
                arr[k] = _Desc(ws[k].data_ptr(), None, Gs[k].data_ptr(),
                               gl[k].data_ptr() if gl[k] is not None else None, p.co[i], p.row[i],
                               p.elem_off[i] - grp.elem0, p.chan_off[i] - grp.chan0)
            return bytes(arr)
        table = grp.pool.get(key, fill)
        aux = aux_all.data_ptr() + 4 * grp.chan0           # the group's first channel in row 0 of [4][total_co]
        stats = None
        if grp.method == QNMethod.AEWGS.value and ops._dist_active():
            stats = torch.empty(3, grp.co, dtype=torch.float32, device=dev)
            _lib.check(L.mhaq_fq_wlayer_aewgs_stats_group(table.data_ptr(), n, grp.co, aux, p.total_co,
                                                          stats.data_ptr(), ops._stream()),
                       "mhaq_fq_wlayer_aewgs_stats_group")
            ops._allreduce_avg_(stats)                     # gdnsq.py:126-129, one message for the whole group
        gw = torch.empty(grp.elems, dtype=torch.float32, device=dev)
        gls = torch.empty(grp.co, dtype=torch.float32, device=dev)
        _, seed, offset, odev = ops._signs(None, grp.method, aux_all)
        _lib.check(L.mhaq_fq_wlayer_bwd_group(table.data_ptr(), n, grp.co, grp.max_row, aux, p.total_co,
                                              gw.data_ptr(), gls.data_ptr(), grp.method,
                                              stats.data_ptr() if stats is not None else None, seed, offset, odev,
                                              ops._stream()), "mhaq_fq_wlayer_bwd_group")
        out_w, out_ls = [], []
        for k, i in enumerate(grp.idx):
            e0, c0 = p.elem_off[i] - grp.elem0, p.chan_off[i] - grp.chan0
            out_w.append(torch.as_strided(gw[e0:e0 + p.co[i] * p.row[i]], ws[k].shape, ws[k].stride()))
            out_ls.append(gls[c0:c0 + p.co[i]].view(ctx.ls_shapes[k]))
        return (None, *out_w, *out_ls)


class MultiTensorWeightQuant:
    """Plan + driver.  `run()` quantizes every per-channel layer's weight in one launch and parks the
    results on the layers; each NoisyConv2d.forward of this step then just picks its slice up."""

    def __init__(self, model: torch.nn.Module, joint_backward: bool = True, backward_group_elems: int = 0):
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
                return True
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
        self._tables = {}            # (pointers) -> device table: never freed (a captured hipGraph may hold it)
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
        if backward_group_elems > 0 and not self.joint_backward:
            sizes = [co * row for co, row in zip(self.co, self.row)]
            for first, last in backward_groups(sizes, self.methods, backward_group_elems):
                g = _WeightGroup(self, first, last)
                self.groups.append(g)
                for j in range(first, last):
                    self.group_of[j] = g

    @torch.no_grad()
    def _run_forward_only(self):
        L = _lib.lib()
        ws = [ops._require_cuda_f32(m.weight, "weight", any_dense_layout=True) for m in self.layers]
        lss = [ops._require_cuda_f32(m.log_wght_s, "log_wght_s") for m in self.layers]
        dev = ws[0].device
        key = tuple(t.data_ptr() for t in (*ws, *lss))
        table = self._tables.get(key)
        if table is None:
            arr = (_Desc * self.nlayers)()
            for i in range(self.nlayers):
                arr[i] = _Desc(ws[i].data_ptr(), lss[i].data_ptr(), None, None, self.co[i], self.row[i],
                               self.elem_off[i], self.chan_off[i])
            table = self._tables[key] = _upload(arr, dev)
        wq_all = torch.empty(self.total_elems, dtype=torch.float32, device=dev)
        aux_all = torch.empty(4, self.total_co, dtype=torch.float32, device=dev)
        _lib.check(L.mhaq_fq_wlayer_fwd_multi(table[0].data_ptr(), self.nlayers, self.total_co, self.max_row,
                                              wq_all.data_ptr(), aux_all.data_ptr(), ops._stream()),
                   "mhaq_fq_wlayer_fwd_multi")
        self.cur_wq, self.cur_aux = wq_all, aux_all
        for g in self.groups:
            g.outs = None
        for i, m in enumerate(self.layers):
            sl = slice(self.chan_off[i], self.chan_off[i] + self.co[i])
            wq = wq_all[self.elem_off[i]:self.elem_off[i] + self.co[i] * self.row[i]]
            # the slab holds each layer in the physical order of its weight (a channels_last weight is
            # [Co][kh][kw][Ci] in memory): give the slice the weight's own strides
            wq = torch.as_strided(wq, ws[i].shape, ws[i].stride())
            m._pre_fwd = ((wq, aux_all[0, sl], aux_all[1, sl], aux_all[2, sl], aux_all[3, sl]),
                          (m.weight._version, m.log_wght_s._version, m.weight.data_ptr()),
                          None if self.group_of[i] is None else (self.group_of[i], i))

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
            m._precomputed = (wqs[i], aux_all[1, sl].view(shp), aux_all[0, sl].view(shp), lwqs[i],
                              (m.weight._version, m.log_wght_s._version, torch.is_grad_enabled()))
        return wqs
