"""Multi-tensor weight fake-quant: every PER_CHANNEL NoisyConv2d of a model in ONE launch per direction
(mhaq_fq_wlayer_fwd_multi / mhaq_fq_wlayer_bwd_multi, a device-resident pointer table).

The weights do not depend on the activations, so all 16-33 weight quantizers of a step can run before the
forward pass starts; autograd calls the joint backward once every layer's dL/dwq has arrived.  That trades
16-33 launches per direction for one, and delays the weight gradients to the end of backward -- which is
why the data-parallel trainer keeps the per-layer ops (DDP overlaps the gradient all-reduce with backward)
and uses this path only when asked to (`QATTrainer(..., multi_tensor_weights=True)`, single GPU).
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


class MultiTensorWeightQuant:
    """Plan + driver.  `run()` quantizes every per-channel layer's weight in one launch and parks the
    results on the layers; each NoisyConv2d.forward of this step then just picks its slice up."""

    def __init__(self, model: torch.nn.Module, joint_backward: bool = True):
        """joint_backward=True: one launch per direction (single GPU: every weight gradient arrives at the end of
        backward).  False: only the FORWARD is batched -- the weights do not depend on the activations, so one launch
        quantizes them all before the forward pass starts -- and every layer keeps its own backward launch, which is
        what data-parallel training needs (gradient overlap, the AEWGS statistics exchange)."""
        from .layers import NoisyConv2d
        self.joint_backward = bool(joint_backward)
        self.layers = [m for m in model.modules()
                       if isinstance(m, NoisyConv2d) and m.qscheme == QScheme.PER_CHANNEL and not m.quant_bias]
        if not self.layers:
            raise ValueError("no PER_CHANNEL NoisyConv2d layers to batch")
        methods = {ops._method_value(m.Q.qnmethod) for m in self.layers}
        if len(methods) != 1 and self.joint_backward:
            raise ValueError("all batched layers must use the same estimator")
        self.method = methods.pop()
        self._tables = {}            # (pointers) -> device table: never freed (a captured hipGraph may hold it)
        self.nlayers = len(self.layers)
        self.shape = [tuple(m.weight.shape) for m in self.layers]
        self.co = [s[0] for s in self.shape]
        self.row = [int(torch.Size(s[1:]).numel()) for s in self.shape]
        self.elem_off, self.chan_off = [], []
        e = c = 0
        for co, row in zip(self.co, self.row):
            self.elem_off.append(e)
            self.chan_off.append(c)
            e += co * row
            c += co
        self.total_elems, self.total_co, self.max_row = e, c, max(self.row)

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
        for i, m in enumerate(self.layers):
            sl = slice(self.chan_off[i], self.chan_off[i] + self.co[i])
            wq = wq_all[self.elem_off[i]:self.elem_off[i] + self.co[i] * self.row[i]]
            # the slab holds each layer in the physical order of its weight (a channels_last weight is
            # [Co][kh][kw][Ci] in memory): give the slice the weight's own strides
            wq = torch.as_strided(wq, ws[i].shape, ws[i].stride())
            m._pre_fwd = ((wq, aux_all[0, sl], aux_all[1, sl], aux_all[2, sl], aux_all[3, sl]),
                          (m.weight._version, m.log_wght_s._version, m.weight.data_ptr()))

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
