"""torch.autograd.Function ops over the C-ABI HIP kernels (include/mhaq_fq.h).

PyTorch is plumbing here: it owns device memory, streams and the autograd graph; all
arithmetic of the fake-quant path runs in mhaq_amd/csrc/*.hip.  Each op states the
reference lines it replaces.  There is no eager / CPU fallback: a missing library or a
non-CUDA tensor raises.

  fake_quant_per_tensor       Quantizer.quantize+dequantize for a per-tensor quantizer
                              (gdnsq.py:189-229 as driven by gdnsq_act.py:39-55); _eval: + q range, flags
  fake_quant_act_layer        NoisyAct.forward from log_act_s / log_act_q / act_b (gdnsq_act.py:39-55)
  fake_quant_weight_layer     NoisyConv2d weight path from log_wght_s, PER_CHANNEL, with the regulariser
                              input log2(max-min+s) (gdnsq_conv2d.py:71-98, model_helper.py:24-44)
  fake_quant_weight_layer_pt  the same for a small PER_TENSOR layer (one workgroup)
  fake_quant_weight_pc / _pt  the weight path from an explicit scale tensor (Quantizer facade, large
                              per-tensor layers)
  fake_quant_per_element      per-element scale / zero point (quantized bias, gdnsq_conv2d.py:86-94)
  potential_loss              PotentialLoss arithmetic (gdnsq_loss.py:47-71)
  minmax, fill_r              fused min/max sweep (calibration, zero point); the sign stream, materialised
"""
from __future__ import annotations

import contextlib
import ctypes
import math
import os
import threading

import torch
import torch.distributed as dist

from . import _lib
from .enums import QNMethod

_MASK64 = (1 << 64) - 1
_i32 = ctypes.c_int32
_byref = ctypes.byref


# ----------------------------------------------------------------------------- RNG stream
class _Rng:
    """(seed, offset) source for the in-kernel Philox sign stream (SURVEY.md section 8e:
    ranks must draw different streams; every backward call gets a fresh offset)."""

    def __init__(self):
        self.seed = None
        self._count = 0
        self._lock = threading.Lock()
        # hipGraph capture freezes a launch's (seed, offset) arguments into the graph.  Every backward entry point
        # therefore also takes `offset_dev` (include/mhaq_fq.h): a device-resident uint64 the kernel adds to the
        # host offset.  A capturing trainer installs its word here (device_offset) and advances it by the number
        # of sign streams one step draws at the end of every replay: replay k of a launch captured with host
        # offset c then uses (seed, c + k * stride) -- the offsets the eager loop would have reached -- at 0
        # extra bytes per element.
        self.offset_base = None

    def manual_seed(self, seed: int):
        with self._lock:
            self.seed = int(seed) & _MASK64
            self._count = 0

    def next(self):
        if self.seed is None:
            self.manual_seed(torch.initial_seed())
        rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        seed = (self.seed ^ ((rank * 0x9E3779B97F4A7C15) & _MASK64)) & _MASK64
        with self._lock:      # autograd runs backward on its own thread per device
            self._count += 1
            return seed, self._count

    def drawn(self) -> int:
        """How many sign streams have been handed out since the last manual_seed."""
        return self._count

    @contextlib.contextmanager
    def device_offset(self, base):
        """While active, backward launches add the uint64 at `base` (an int64[1] device tensor) to their host
        offset.  Scoped to the caller (a capturing trainer wraps its step in it) and restored on exit."""
        if base is not None and (base.dtype != torch.int64 or base.numel() != 1 or not base.is_cuda):
            raise ValueError("device_offset: base must be a one-element int64 device tensor")
        prev, self.offset_base = self.offset_base, base
        try:
            yield base
        finally:
            self.offset_base = prev


rng = _Rng()


def manual_seed(seed: int) -> None:
    """Seed the stochastic scale-gradient sign stream (gdnsq.py:54 randint_like)."""
    rng.manual_seed(seed)


# ----------------------------------------------------------------------------- helpers
def _method_value(method) -> int:
    if isinstance(method, QNMethod):
        return method.value
    if isinstance(method, str):
        try:
            return QNMethod[method].value
        except KeyError:
            raise AttributeError(f"Unknown method {method}!")  # gdnsq.py:241
    if isinstance(method, int) and 0 <= method <= 3:
        return method
    if hasattr(method, "name") and method.name in QNMethod.__members__:  # foreign Enum with the same names
        return QNMethod[method.name].value
    raise AttributeError(f"Unknown method {method}!")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current HIP stream handle (hipStream_t as an int) of the current device."""
    if _raw_stream is not None:     # 0.3 us instead of 2.9 us for the Stream-object round trip
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _require_cuda_f32(t: torch.Tensor, name: str, any_dense_layout: bool = False) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.MhaqFqError(
            f"{name} is on {t.device}: the fake-quant path runs only as HIP kernels on an MI355X "
            "(no CPU fallback by design)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (the reference forces fp32), got {t.dtype}")
    if any_dense_layout and _is_dense(t):
        return t          # e.g. channels_last: a per-tensor quantizer is elementwise, memory order is free
    return t.contiguous()


def _is_dense(t: torch.Tensor) -> bool:
    """Non-overlapping and dense in SOME dimension order (contiguous, channels_last, ...)."""
    if t.is_contiguous() or t.dim() < 2:
        return t.is_contiguous()
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return True
    if t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d):
        return True
    return False


def _like_layout(g: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """The upstream gradient in the memory order of x (the kernels walk both as flat streams)."""
    if g.stride() == x.stride() and g.shape == x.shape:
        return g
    out = torch.empty_like(x)       # preserve_format: x's strides
    out.copy_(g)
    return out


_const_cache = {}


def _scalar(v, device, name):
    """A 1-element fp32 device tensor for a quantizer parameter (tensor or python number)."""
    if torch.is_tensor(v):
        if v.numel() != 1:
            raise ValueError(f"{name} must have one element for a per-tensor quantizer, got {tuple(v.shape)}")
        if v.device != device or v.dtype != torch.float32:
            v = v.to(device=device, dtype=torch.float32)
        return v
    key = (float(v), device)
    t = _const_cache.get(key)
    if t is None:
        t = torch.full((1,), float(v), dtype=torch.float32, device=device)
        _const_cache[key] = t
    return t


def _workspace(nbytes, device):
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def _r_ptr(r_sign, like):
    if r_sign is None:
        return None
    if r_sign.dtype != torch.int8 or r_sign.numel() != like.numel() or r_sign.device != like.device:
        raise ValueError("r_sign must be an int8 tensor (positive = +0.5, else -0.5) with the size and device "
                         "of the input")
    return r_sign.contiguous()


def _dist_active() -> bool:
    """True when collectives must run.  MHAQ_FORCE_COLLECTIVES=1 also runs them at world size 1 (used to
    rehearse the multi-GPU code path on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("MHAQ_FORCE_COLLECTIVES") == "1"


def _allreduce_avg_(t: torch.Tensor) -> None:
    """The AEWGS statistics exchange of gdnsq.py:126-129, packed into ONE message."""
    if _dist_active():
        if dist.get_backend() == "gloo":
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t.div_(dist.get_world_size())
        else:
            dist.all_reduce(t, op=dist.ReduceOp.AVG)


def _signs(r_sign, method: int, like: torch.Tensor):
    """(r_sign, seed, offset, offset_dev) for one backward call: explicit signs (checker) / none needed (LSQ) /
    the in-kernel Philox stream (default), shifted by the device-resident base when one is installed."""
    if r_sign is not None or method == QNMethod.LSQ.value:
        return r_sign, 0, 0, None
    seed, offset = rng.next()
    base = rng.offset_base
    if base is not None and base.device != like.device:
        base = None
    return None, seed, offset, (base.data_ptr() if base is not None else None)


def fill_r(n: int, seed: int, offset: int, device) -> torch.Tensor:
    """Materialise the in-kernel sign stream as int8 +-1 (checker use)."""
    out = torch.empty(n, dtype=torch.int8, device=device)
    _lib.check(_lib.lib().mhaq_fq_fill_r(out.data_ptr(), n, seed & _MASK64, offset & _MASK64, _stream()),
               "mhaq_fq_fill_r")
    return out


def minmax(x: torch.Tensor) -> torch.Tensor:
    """[min, max] of a tensor in one fused sweep (weight zero point; min/max observer)."""
    x = _require_cuda_f32(x.detach(), "x")
    L = _lib.lib()
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    nb = L.mhaq_fq_minmax_workspace_bytes(x.numel())
    ws = _workspace(nb, x.device)
    _lib.check(L.mhaq_fq_minmax(x.data_ptr(), x.numel(), out.data_ptr(), ws.data_ptr(), nb, _stream()),
               "mhaq_fq_minmax")
    return out


def row_minmax(w: torch.Tensor):
    """(min[co], max[co]) over everything but dim 0 in one read-only sweep (weight-scale calibration)."""
    w = _require_cuda_f32(w.detach(), "weight", any_dense_layout=True)   # a channels_last row is a permuted row
    co = w.shape[0]
    mn = torch.empty(co, dtype=torch.float32, device=w.device)
    mx = torch.empty(co, dtype=torch.float32, device=w.device)
    if co:
        _lib.check(_lib.lib().mhaq_fq_row_minmax(w.data_ptr(), co, w.numel() // co, mn.data_ptr(), mx.data_ptr(),
                                                 _stream()), "mhaq_fq_row_minmax")
    return mn, mx


# ----------------------------------------------------------------------------- per-tensor op
def _pt_forward(x, s, zp, lo, hi, want_q=False, want_stats=False):
    L = _lib.lib()
    y = torch.empty_like(x)
    q = torch.empty_like(x) if want_q else None
    qstats = flags = ws = None
    nb = 0
    if want_stats:
        qstats = torch.empty(2, dtype=torch.float32, device=x.device)
        flags = torch.empty(1, dtype=torch.int32, device=x.device)
        nb = L.mhaq_fq_pt_fwd_workspace_bytes(x.numel())
        ws = _workspace(nb, x.device)
    _lib.check(L.mhaq_fq_pt_fwd(x.data_ptr(), y.data_ptr(), x.numel(), s.data_ptr(), zp.data_ptr(),
                                lo.data_ptr(), hi.data_ptr(), q.data_ptr() if want_q else None,
                                qstats.data_ptr() if want_stats else None,
                                flags.data_ptr() if want_stats else None,
                                ws.data_ptr() if ws is not None else None, nb, _stream()),
               "mhaq_fq_pt_fwd")
    return y, q, qstats, flags


def _pt_backward(x, g, s, zp, lo, hi, method, col_stats, period, r_sign, count_ties=False):
    L = _lib.lib()
    gx = torch.empty_like(x)
    grads = torch.empty(5, dtype=torch.float32, device=x.device)
    nb = L.mhaq_fq_pt_bwd_workspace_bytes(x.numel())
    ws = _workspace(nb, x.device)
    r_sign, seed, offset, odev = _signs(r_sign, method, x)
    _lib.check(L.mhaq_fq_pt_bwd(x.data_ptr(), g.data_ptr(), gx.data_ptr(), x.numel(), s.data_ptr(),
                                zp.data_ptr(), lo.data_ptr(), hi.data_ptr(), method,
                                col_stats.data_ptr() if col_stats is not None else None, period,
                                r_sign.data_ptr() if r_sign is not None else None, seed, offset, odev,
                                1 if count_ties else 0, grads.data_ptr(), ws.data_ptr(), nb, _stream()),
               "mhaq_fq_pt_bwd")
    return gx, grads


def _col_stats(x, g, s, zp, lo, hi):
    """AEWGS statistics for a [1]-shaped scale: means over dim 0 (gdnsq.py:150-152 quirk)."""
    co = x.shape[0]
    row = x.numel() // co
    stats = torch.empty(3, row, dtype=torch.float32, device=x.device)
    L = _lib.lib()
    nb = L.mhaq_fq_pt_aewgs_colstats_workspace_bytes(co, row)
    ws = _workspace(nb, x.device) if nb else None
    _lib.check(L.mhaq_fq_pt_aewgs_colstats(x.data_ptr(), g.data_ptr(), co, row, s.data_ptr(), zp.data_ptr(),
                                           lo.data_ptr(), hi.data_ptr(), stats.data_ptr(),
                                           ws.data_ptr() if ws is not None else None, nb, _stream()),
               "mhaq_fq_pt_aewgs_colstats")
    _allreduce_avg_(stats)
    return stats, row


class FakeQuantPerTensor(torch.autograd.Function):
    """y = dequantize(quantize(x)) with per-tensor (s, zp, lo, hi); fused fwd and fused bwd kernels."""

    @staticmethod
    def forward(ctx, x, s, zp, lo, hi, method, r_sign):
        y, _, _, _ = _pt_forward(x, s, zp, lo, hi)
        ctx.save_for_backward(x, s, zp, lo, hi)
        ctx.method = method
        ctx.r_sign = r_sign
        return y

    @staticmethod
    def backward(ctx, g):
        x, s, zp, lo, hi = ctx.saved_tensors
        g = g.contiguous()
        col_stats, period = None, 0
        if ctx.method == QNMethod.AEWGS.value:
            col_stats, period = _col_stats(x, g, s, zp, lo, hi)
        gx, grads = _pt_backward(x, g, s, zp, lo, hi, ctx.method, col_stats, period, ctx.r_sign)
        return (gx, grads[0].reshape(s.shape), grads[1].reshape(zp.shape), grads[2].reshape(lo.shape),
                grads[3].reshape(hi.shape), None, None)


def fake_quant_per_tensor(x, scale, zero_point, min_val, max_val, method=QNMethod.STE, r_sign=None):
    """Fused Quantizer.dequantize(Quantizer.quantize(x)) (gdnsq.py:189-229), differentiable w.r.t.
    x, scale, zero_point, min_val, max_val (tensor-valued ones)."""
    x = _require_cuda_f32(x, "x")
    dev = x.device
    s = _scalar(scale, dev, "scale")
    zp = _scalar(zero_point, dev, "zero_point")
    lo = _scalar(min_val, dev, "min_val")
    hi = _scalar(max_val, dev, "max_val")
    return FakeQuantPerTensor.apply(x, s, zp, lo, hi, _method_value(method), _r_ptr(r_sign, x))


@torch.no_grad()
def fake_quant_per_tensor_eval(x, scale, zero_point, min_val, max_val, want_q=False):
    """Eval-mode forward: returns (y, q or None, qstats[2] = {min q, max q}, flags[1]).
    Replaces gdnsq.py:211-217 (asserts, as a device-side flag word) and gdnsq_act.py:51-54."""
    x = _require_cuda_f32(x, "x")
    dev = x.device
    return _pt_forward(x, _scalar(scale, dev, "scale"), _scalar(zero_point, dev, "zero_point"),
                       _scalar(min_val, dev, "min_val"), _scalar(max_val, dev, "max_val"),
                       want_q=want_q, want_stats=True)


# ----------------------------------------------------------------------------- NoisyAct layer op
_placeholders = {}
_act_ws_bytes = {}


def _placeholder(device):
    """Stand-in gradient a deferred activation backward hands to autograd; the hub's node replaces it."""
    t = _placeholders.get(device)
    if t is None:
        t = torch.zeros(1, dtype=torch.float32, device=device)
        _placeholders[device] = t
    return t


class FakeQuantActLayer(torch.autograd.Function):
    """NoisyAct.forward from its learnable parameters in two launches per direction
    (gdnsq_act.py:39-55): returns (y, params[5] = {s, zp, lo, hi, qr}).
    `hub_slot` = (ActGradHub, slot): the backward leaves its partial sums with the hub, whose single finalize
    launch serves every quantizer of the pass (act_hub.py); None = finalize right here."""

    @staticmethod
    def forward(ctx, x, log_s, log_q, b, method, r_sign, hub_slot):
        L = _lib.lib()
        y = torch.empty_like(x)
        params = torch.empty(5, dtype=torch.float32, device=x.device)
        _lib.check(L.mhaq_fq_act_fwd(x.data_ptr(), y.data_ptr(), x.numel(), log_s.data_ptr(), log_q.data_ptr(),
                                     b.data_ptr(), params.data_ptr(), None, None, None, 0, _stream()),
                   "mhaq_fq_act_fwd")
        ctx.save_for_backward(x, params)
        ctx.method, ctx.r_sign, ctx.hub_slot = method, r_sign, hub_slot
        ctx.shapes = (log_s.shape, log_q.shape, b.shape)
        ctx.mark_non_differentiable(params)
        return y, params

    @staticmethod
    def backward(ctx, g, _gparams):
        L = _lib.lib()
        x, params = ctx.saved_tensors
        g = _like_layout(g, x)
        gx = torch.empty_like(x)
        nb = _act_ws_bytes.get(x.numel())
        if nb is None:                      # one C call per op on the hot path: the size query is memoised
            nb = _act_ws_bytes[x.numel()] = L.mhaq_fq_act_bwd_workspace_bytes(x.numel())
        r_sign, seed, offset, odev = _signs(ctx.r_sign, ctx.method, x)
        need = ctx.needs_input_grad
        if ctx.hub_slot is not None:
            hub, slot = ctx.hub_slot
            ws = hub.workspace(slot, nb, x.device)
            nparts = _i32()
            _lib.check(L.mhaq_fq_act_bwd_partials(x.data_ptr(), g.data_ptr(), gx.data_ptr(), x.numel(),
                                                  params.data_ptr(), ctx.method,
                                                  r_sign.data_ptr() if r_sign is not None else None, seed, offset,
                                                  odev, ws.data_ptr(), nb, _byref(nparts), _stream()),
                       "mhaq_fq_act_bwd_partials")
            hub.record(slot, nparts.value, ws)
            ph = _placeholder(x.device)
            return (gx if need[0] else None, ph if need[1] else None, ph if need[2] else None,
                    ph if need[3] else None, None, None, None)
        grads = torch.empty(3, dtype=torch.float32, device=x.device)
        ws = _workspace(nb, x.device)
        _lib.check(L.mhaq_fq_act_bwd(x.data_ptr(), g.data_ptr(), gx.data_ptr(), x.numel(), params.data_ptr(),
                                     ctx.method, r_sign.data_ptr() if r_sign is not None else None, seed, offset,
                                     odev, grads.data_ptr(), ws.data_ptr(), nb, _stream()), "mhaq_fq_act_bwd")
        return (gx if need[0] else None,
                grads[0].reshape(ctx.shapes[0]) if need[1] else None,
                grads[1].reshape(ctx.shapes[1]) if need[2] else None,
                grads[2].reshape(ctx.shapes[2]) if need[3] else None, None, None, None)


def fake_quant_act_layer(x, log_act_s, log_act_q, act_b, method=QNMethod.STE, r_sign=None, hub_slot=None):
    """Fused NoisyAct training forward: (y, params).  AEWGS is not offered here (the reference never
    builds an AEWGS activation quantizer); use fake_quant_per_tensor for it."""
    x = _require_cuda_f32(x, "x", any_dense_layout=True)
    dev = x.device
    m = _method_value(method)
    if m == QNMethod.AEWGS.value:
        raise NotImplementedError("AEWGS activations go through fake_quant_per_tensor")
    return FakeQuantActLayer.apply(x, _scalar(log_act_s, dev, "log_act_s"), _scalar(log_act_q, dev, "log_act_q"),
                                   _scalar(act_b, dev, "act_b"), m, _r_ptr(r_sign, x), hub_slot)


@torch.no_grad()
def fake_quant_act_layer_eval(x, log_act_s, log_act_q, act_b):
    """Eval-mode NoisyAct in one launch (+ a tiny finalize): (y, params, qstats[2], flags[1])."""
    x = _require_cuda_f32(x, "x", any_dense_layout=True)
    dev = x.device
    L = _lib.lib()
    y = torch.empty_like(x)
    params = torch.empty(5, dtype=torch.float32, device=dev)
    qstats = torch.empty(2, dtype=torch.float32, device=dev)
    flags = torch.empty(1, dtype=torch.int32, device=dev)
    nb = L.mhaq_fq_pt_fwd_workspace_bytes(x.numel())
    ws = _workspace(nb, dev)
    _lib.check(L.mhaq_fq_act_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _scalar(log_act_s, dev, "s").data_ptr(),
                                 _scalar(log_act_q, dev, "q").data_ptr(), _scalar(act_b, dev, "b").data_ptr(),
                                 params.data_ptr(), qstats.data_ptr(), flags.data_ptr(), ws.data_ptr(), nb,
                                 _stream()), "mhaq_fq_act_fwd")
    return y, params, qstats, flags


# ----------------------------------------------------------------------------- NoisyConv2d layer op (per-channel)
class FakeQuantWeightLayer(torch.autograd.Function):
    """Per-channel NoisyConv2d weight path from log_wght_s, plus the layer's regulariser input
    lwq = log2(max - min + s) (model_helper.py:24-44): returns (wq, zp, s, lwq)."""

    @staticmethod
    def forward(ctx, w, log_s, method, r_sign, zp_grad, pre):
        L = _lib.lib()
        co = w.shape[0]
        row = w.numel() // co
        if pre is not None:
            # this step's forward already ran in the model-wide launch (multi.py, forward-only mode): take its
            # slices; the backward below stays this layer's own launch (DDP overlap, AEWGS exchange)
            wq, s, zp, mx, lwq = pre
        else:
            wq = torch.empty_like(w)
            aux = torch.empty(4, co, dtype=torch.float32, device=w.device)   # s, zp, mx, lwq
            s, zp, mx, lwq = aux[0], aux[1], aux[2], aux[3]
            _lib.check(L.mhaq_fq_wlayer_fwd(w.data_ptr(), wq.data_ptr(), log_s.data_ptr(), co, row, s.data_ptr(),
                                            zp.data_ptr(), mx.data_ptr(), lwq.data_ptr(), _stream()),
                       "mhaq_fq_wlayer_fwd")
        ctx.save_for_backward(w, s, zp, mx)
        ctx.method, ctx.r_sign, ctx.log_s_shape = method, r_sign, log_s.shape
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(s)
        if not zp_grad:
            ctx.mark_non_differentiable(zp)
        return wq, zp, s, lwq

    @staticmethod
    def backward(ctx, G, gzp_extra, _gs, g_lwq):
        L = _lib.lib()
        w, s, zp, mx = ctx.saved_tensors
        G = torch.zeros_like(w) if G is None else _like_layout(G, w)
        gzp_extra = gzp_extra.contiguous() if gzp_extra is not None else None
        g_lwq = g_lwq.contiguous() if g_lwq is not None else None
        co = w.shape[0]
        row = w.numel() // co
        stats = None
        distributed = _dist_active()
        if ctx.method == QNMethod.AEWGS.value and distributed:
            stats = torch.empty(3, co, dtype=torch.float32, device=w.device)
            _lib.check(L.mhaq_fq_pc_aewgs_stats(w.data_ptr(), G.data_ptr(), s.data_ptr(), zp.data_ptr(), co, row,
                                                stats.data_ptr(), _stream()), "mhaq_fq_pc_aewgs_stats")
            _allreduce_avg_(stats)
        gw = torch.empty_like(w)
        gls = torch.empty(co, dtype=torch.float32, device=w.device)
        r_sign, seed, offset, odev = _signs(ctx.r_sign, ctx.method, w)
        _lib.check(L.mhaq_fq_wlayer_bwd(w.data_ptr(), G.data_ptr(), gw.data_ptr(), gls.data_ptr(), s.data_ptr(),
                                        zp.data_ptr(), mx.data_ptr(),
                                        g_lwq.data_ptr() if g_lwq is not None else None, co, row, ctx.method,
                                        stats.data_ptr() if stats is not None else None,
                                        gzp_extra.data_ptr() if gzp_extra is not None else None,
                                        r_sign.data_ptr() if r_sign is not None else None, seed, offset, odev,
                                        _stream()), "mhaq_fq_wlayer_bwd")
        return gw, gls.reshape(ctx.log_s_shape), None, None, None, None


def fake_quant_weight_layer(w, log_wght_s, method=QNMethod.AEWGS, r_sign=None, zp_grad=False, pre=None):
    """Returns (wq, zp[co,1,..], s[co,1,..], lwq[co]) for a PER_CHANNEL layer.
    `pre` = (wq, s, zp, mx, lwq) of this layer from the model-wide forward launch (multi.py), or None."""
    # a channels_last weight [Co,Ci,kh,kw] is physically [Co][kh][kw][Ci]: every output channel is still one
    # contiguous row, and min / quantize / per-channel sums do not care about the order inside a row
    w = _require_cuda_f32(w, "weight", any_dense_layout=True)
    ls = _require_cuda_f32(log_wght_s, "log_wght_s")
    if ls.numel() != w.shape[0]:
        raise ValueError(f"per-channel log scale must have {w.shape[0]} elements, got {tuple(log_wght_s.shape)}")
    wq, zp, s, lwq = FakeQuantWeightLayer.apply(w, ls, _method_value(method), _r_ptr(r_sign, w), bool(zp_grad), pre)
    shp = [w.shape[0]] + [1] * (w.dim() - 1)
    return wq, zp.view(shp), s.view(shp), lwq


class FakeQuantWeightLayerPT(torch.autograd.Function):
    """PER_TENSOR weight layer small enough for one workgroup (every CIFAR ResNet-20 / RFDN layer):
    one launch per direction from log_wght_s, regulariser input included.  Returns (wq, aux[4])
    with aux = {s, zp, max, lwq}; lwq = aux[3:4] is differentiable."""

    @staticmethod
    def forward(ctx, w, log_s, method, r_sign):
        L = _lib.lib()
        wq = torch.empty_like(w)
        aux = torch.empty(4, dtype=torch.float32, device=w.device)
        _lib.check(L.mhaq_fq_wlayer_pt_fwd(w.data_ptr(), wq.data_ptr(), log_s.data_ptr(), w.numel(), aux.data_ptr(),
                                           _stream()), "mhaq_fq_wlayer_pt_fwd")
        lwq = aux[3:4].clone()
        ctx.save_for_backward(w, aux)
        ctx.method, ctx.r_sign, ctx.log_s_shape = method, r_sign, log_s.shape
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(aux)
        return wq, aux, lwq

    @staticmethod
    def backward(ctx, G, _gaux, g_lwq):
        L = _lib.lib()
        w, aux = ctx.saved_tensors
        G = torch.zeros_like(w) if G is None else G.contiguous()
        g_lwq = g_lwq.contiguous() if g_lwq is not None else None
        gw = torch.empty_like(w)
        gls = torch.empty(1, dtype=torch.float32, device=w.device)
        r_sign, seed, offset, odev = _signs(ctx.r_sign, ctx.method, w)
        _lib.check(L.mhaq_fq_wlayer_pt_bwd(w.data_ptr(), G.data_ptr(), gw.data_ptr(), gls.data_ptr(), aux.data_ptr(),
                                           g_lwq.data_ptr() if g_lwq is not None else None, w.numel(), ctx.method,
                                           r_sign.data_ptr() if r_sign is not None else None, seed, offset, odev,
                                           _stream()), "mhaq_fq_wlayer_pt_bwd")
        return gw, gls.reshape(ctx.log_s_shape), None, None


def small_pt_layer_supported(w, method) -> bool:
    return w.numel() <= _lib.lib().mhaq_fq_wlayer_pt_max_elements() and _method_value(method) != QNMethod.AEWGS.value


def fake_quant_weight_layer_pt(w, log_wght_s, method=QNMethod.STE, r_sign=None):
    """Returns (wq, zp 0-dim, s [1], lwq [1]) for a small PER_TENSOR layer (see small_pt_layer_supported)."""
    w = _require_cuda_f32(w, "weight")
    ls = _scalar(log_wght_s, w.device, "log_wght_s")
    wq, aux, lwq = FakeQuantWeightLayerPT.apply(w, ls, _method_value(method), _r_ptr(r_sign, w))
    return wq, aux[1].reshape(()), aux[0:1], lwq


# ----------------------------------------------------------------------------- per-channel weight op
class FakeQuantWeightPC(torch.autograd.Function):
    """Per-channel weight fake-quant; returns (wq, zp[co]).  zp is the row minimum; its gradient
    (amin backward, tie split) is folded into gw by the kernel, including any gradient that
    reaches the zp OUTPUT from another consumer (the quantized bias)."""

    @staticmethod
    def forward(ctx, w, s, method, r_sign, zp_grad):
        L = _lib.lib()
        co = w.shape[0]
        row = w.numel() // co
        wq = torch.empty_like(w)
        zp = torch.empty(co, dtype=torch.float32, device=w.device)
        _lib.check(L.mhaq_fq_pc_fwd(w.data_ptr(), wq.data_ptr(), zp.data_ptr(), None, s.data_ptr(), co, row,
                                    _stream()), "mhaq_fq_pc_fwd")
        ctx.save_for_backward(w, s, zp)
        ctx.method = method
        ctx.r_sign = r_sign
        ctx.set_materialize_grads(False)
        if not zp_grad:
            ctx.mark_non_differentiable(zp)
        return wq, zp

    @staticmethod
    def backward(ctx, G, gzp_extra):
        L = _lib.lib()
        w, s, zp = ctx.saved_tensors
        if G is None:
            G = torch.zeros_like(w)
        G = G.contiguous()
        if gzp_extra is not None:
            gzp_extra = gzp_extra.contiguous()
        co = w.shape[0]
        row = w.numel() // co
        stats = None
        distributed = _dist_active()
        if ctx.method == QNMethod.AEWGS.value and distributed:
            stats = torch.empty(3, co, dtype=torch.float32, device=w.device)
            _lib.check(L.mhaq_fq_pc_aewgs_stats(w.data_ptr(), G.data_ptr(), s.data_ptr(), zp.data_ptr(), co, row,
                                                stats.data_ptr(), _stream()), "mhaq_fq_pc_aewgs_stats")
            _allreduce_avg_(stats)
        gw = torch.empty_like(w)
        gs = torch.empty(co, dtype=torch.float32, device=w.device)
        r_sign, seed, offset, odev = _signs(ctx.r_sign, ctx.method, w)
        _lib.check(L.mhaq_fq_pc_bwd(w.data_ptr(), G.data_ptr(), gw.data_ptr(), gs.data_ptr(), s.data_ptr(),
                                    zp.data_ptr(), co, row, ctx.method,
                                    stats.data_ptr() if stats is not None else None,
                                    gzp_extra.data_ptr() if gzp_extra is not None else None,
                                    r_sign.data_ptr() if r_sign is not None else None, seed, offset, odev, _stream()),
                   "mhaq_fq_pc_bwd")
        return gw, gs.reshape(s.shape), None, None, None


def fake_quant_weight_pc(w, scale, method=QNMethod.AEWGS, r_sign=None, zp_grad=False):
    """NoisyConv2d per-channel weight path (gdnsq_conv2d.py:71-98): returns (wq, zp).
    zp_grad=True keeps zp differentiable (needed when the bias is quantized with it)."""
    w = _require_cuda_f32(w, "weight")
    s = _require_cuda_f32(scale, "scale")
    if s.numel() != w.shape[0]:
        raise ValueError(f"per-channel scale must have {w.shape[0]} elements, got {tuple(scale.shape)}")
    wq, zp = FakeQuantWeightPC.apply(w, s, _method_value(method), _r_ptr(r_sign, w), bool(zp_grad))
    return wq, zp.view([w.shape[0]] + [1] * (w.dim() - 1))


# ----------------------------------------------------------------------------- per-element op (bias)
class FakeQuantPerElement(torch.autograd.Function):
    """x[i] fake-quantized with s[i], zp[i] (gdnsq_conv2d.py:86-94, quant_bias=True)."""

    @staticmethod
    def forward(ctx, x, s, zp, method, r_sign):
        y = torch.empty_like(x)
        _lib.check(_lib.lib().mhaq_fq_vec_fwd(x.data_ptr(), y.data_ptr(), None, s.data_ptr(), zp.data_ptr(),
                                              x.numel(), _stream()), "mhaq_fq_vec_fwd")
        ctx.save_for_backward(x, s, zp)
        ctx.method = method
        ctx.r_sign = r_sign
        return y

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        x, s, zp = ctx.saved_tensors
        g = g.contiguous()
        n = x.numel()
        stats = None
        if ctx.method == QNMethod.AEWGS.value:
            stats = torch.empty(3, dtype=torch.float32, device=x.device)
            _lib.check(L.mhaq_fq_vec_aewgs_stats(x.data_ptr(), g.data_ptr(), s.data_ptr(), zp.data_ptr(), n,
                                                 stats.data_ptr(), _stream()), "mhaq_fq_vec_aewgs_stats")
            _allreduce_avg_(stats)
        gx, gs, gzp = torch.empty_like(x), torch.empty_like(s), torch.empty_like(zp)
        r_sign, seed, offset, odev = _signs(ctx.r_sign, ctx.method, x)
        _lib.check(L.mhaq_fq_vec_bwd(x.data_ptr(), g.data_ptr(), gx.data_ptr(), gs.data_ptr(), gzp.data_ptr(),
                                     s.data_ptr(), zp.data_ptr(), n, ctx.method,
                                     stats.data_ptr() if stats is not None else None,
                                     r_sign.data_ptr() if r_sign is not None else None, seed, offset, odev, _stream()),
                   "mhaq_fq_vec_bwd")
        return gx, gs, gzp, None, None


def fake_quant_per_element(x, scale, zero_point, method=QNMethod.AEWGS, r_sign=None):
    x = _require_cuda_f32(x, "x")
    s = _require_cuda_f32(scale, "scale")
    zp = _require_cuda_f32(zero_point, "zero_point")
    if s.shape != x.shape or zp.shape != x.shape:
        raise ValueError("per-element quantizer needs scale / zero_point of the input's shape")
    return FakeQuantPerElement.apply(x, s, zp, _method_value(method), _r_ptr(r_sign, x))


# ----------------------------------------------------------------------------- per-tensor weight op
class FakeQuantWeightPT(torch.autograd.Function):
    """Per-tensor weight fake-quant; zp = global minimum (gdnsq_conv2d.py:82-83)."""

    @staticmethod
    def forward(ctx, w, s, method, r_sign):
        mm = minmax(w)
        zp = mm[0:1]
        ninf = _scalar(-math.inf, w.device, "lo")
        pinf = _scalar(math.inf, w.device, "hi")
        wq, _, _, _ = _pt_forward(w, s, zp, ninf, pinf)
        ctx.save_for_backward(w, s, zp)
        ctx.method = method
        ctx.r_sign = r_sign
        ctx.mark_non_differentiable(zp)
        return wq, zp

    @staticmethod
    def backward(ctx, G, _gzp):
        w, s, zp = ctx.saved_tensors
        G = G.contiguous()
        ninf = _scalar(-math.inf, w.device, "lo")
        pinf = _scalar(math.inf, w.device, "hi")
        col_stats, period = None, 0
        if ctx.method == QNMethod.AEWGS.value:
            col_stats, period = _col_stats(w, G, s, zp, ninf, pinf)
        gw, grads = _pt_backward(w, G, s, zp, ninf, pinf, ctx.method, col_stats, period, ctx.r_sign,
                                 count_ties=True)
        _lib.check(_lib.lib().mhaq_fq_pt_tie_scatter(w.data_ptr(), gw.data_ptr(), w.numel(), zp.data_ptr(),
                                                     grads.data_ptr(), _stream()), "mhaq_fq_pt_tie_scatter")
        return gw, grads[0].reshape(s.shape), None, None


def fake_quant_weight_pt(w, scale, method=QNMethod.AEWGS, r_sign=None):
    """NoisyConv2d / NoisyLinear per-tensor weight path: returns (wq, zp 0-dim)."""
    w = _require_cuda_f32(w, "weight")
    s = _scalar(scale, w.device, "scale")
    wq, zp = FakeQuantWeightPT.apply(w, s, _method_value(method), _r_ptr(r_sign, w))
    return wq, zp.reshape(())


# ------------------------------------------------------------------ PotentialLoss
class PotentialLossFn(torch.autograd.Function):
    """gdnsq_loss.py:47-71 / 129-153 in one launch per direction (mhaq_fq_potential_loss_fwd/bwd).
    Returns (ploss, stats[12]); `state` = {loss_sum, cnt, t} is the module's device-resident state, advanced
    in the same launch when `update_state`."""

    @staticmethod
    def forward(ctx, base, las, laq, lws, lwq, state, a_bits, w_bits, p, lossless, update_state):
        L = _lib.lib()
        ctx.shapes = tuple(v.shape for v in (base, las, laq, lws, lwq))
        base = _require_cuda_f32(base.reshape(1), "base_loss")
        vecs = [_require_cuda_f32(v.reshape(-1), n) for v, n in
                ((las, "log_act_s"), (laq, "log_act_q"), (lws, "log_wght_s"), (lwq, "log_w"))]
        las, laq, lws, lwq = vecs
        if las.numel() != laq.numel() or lws.numel() != lwq.numel():
            raise ValueError("potential_loss: scale and range vectors must pair up")
        out = torch.empty(12, dtype=torch.float32, device=base.device)
        _lib.check(L.mhaq_fq_potential_loss_fwd(base.data_ptr(), las.data_ptr(), laq.data_ptr(), las.numel(),
                                                lws.data_ptr(), lwq.data_ptr(), lws.numel(), float(a_bits),
                                                float(w_bits), float(p), int(bool(lossless)), state.data_ptr(),
                                                int(bool(update_state)), out.data_ptr(), _stream()),
                   "mhaq_fq_potential_loss_fwd")
        ctx.save_for_backward(out, las, laq, lws, lwq)
        ctx.cfg = (float(a_bits), float(w_bits), float(p))
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g, _gstats):
        L = _lib.lib()
        out, las, laq, lws, lwq = ctx.saved_tensors
        a_bits, w_bits, p = ctx.cfg
        g = g.reshape(1).contiguous()
        na, nw = las.numel(), lws.numel()
        slab = torch.empty(1 + 2 * na + 2 * nw, dtype=torch.float32, device=out.device)
        g_base, g_las, g_laq, g_lws, g_lwq = torch.split(slab, [1, na, na, nw, nw])
        _lib.check(L.mhaq_fq_potential_loss_bwd(g.data_ptr(), out.data_ptr(), las.data_ptr(), laq.data_ptr(), na,
                                                lws.data_ptr(), lwq.data_ptr(), nw, a_bits, w_bits, p,
                                                g_base.data_ptr(), g_las.data_ptr(), g_laq.data_ptr(),
                                                g_lws.data_ptr(), g_lwq.data_ptr(), _stream()),
                   "mhaq_fq_potential_loss_bwd")
        grads = [v.reshape(shp) for v, shp in zip((g_base, g_las, g_laq, g_lws, g_lwq), ctx.shapes)]
        return (*grads, *(None,) * 6)


def potential_loss(base, las, laq, lws, lwq, state, a_bits, w_bits, p=1, lossless=False, update_state=False):
    """`state`: float32 device tensor {loss_sum, cnt, t} (see include/mhaq_fq.h)."""
    if state.dtype != torch.float32 or state.numel() != 3 or not state.is_cuda or not state.is_contiguous():
        raise ValueError("potential_loss: state must be a contiguous float32 device tensor {loss_sum, cnt, t}")
    return PotentialLossFn.apply(base, las, laq, lws, lwq, state, a_bits, w_bits, p, lossless, update_state)
