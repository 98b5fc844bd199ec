"""torch.autograd.Function ops over the C-ABI HIP kernels (include/mhaq_fq.h).

PyTorch is plumbing here: it owns device memory, streams and the autograd graph; all
arithmetic of the fake-quant path runs in mhaq_amd/csrc/*.hip.  Each op states the
reference lines it replaces.  There is no eager / CPU fallback: a missing library or a
non-CUDA tensor raises.

The ops of the step loop -- fake_quant_act_layer, fake_quant_weight_layer, fake_quant_weight_layer_pt,
potential_loss (and the activation hub / weight groups of act_hub.py / multi.py) -- are torch::autograd::Function
nodes COMPILED in mhaq_amd/csrc/torch_binding.cpp (loaded by _ext.py) over the same C ABI: one pybind11 call per
forward, a backward that never takes the GIL.  The remaining, colder ops below are Python autograd Functions over
ctypes.  Both draw their random sign streams from one source (the extension's, `rng` is its facade).

  fake_quant_per_tensor       Quantizer.quantize+dequantize for a per-tensor quantizer
                              (gdnsq.py:189-229 as driven by gdnsq_act.py:39-55); _eval: + q range, flags
  fake_quant_act_layer        NoisyAct.forward from log_act_s / log_act_q / act_b (gdnsq_act.py:39-55)
  fake_quant_weight_layer     NoisyConv2d weight path from log_wght_s, PER_CHANNEL, with the regulariser
                              input log2(max-min+s) (gdnsq_conv2d.py:71-98, model_helper.py:24-44)
  fake_quant_weight_layer_pt  the same for a small PER_TENSOR layer (one workgroup)
  fake_quant_weight_layer_ptl the same for a PER_TENSOR layer of any size / any estimator (streaming launches)
  fake_quant_weight_pc / _pt  the weight path from an explicit scale tensor (Quantizer facade, large
                              per-tensor layers)
  fake_quant_per_element      per-element scale / zero point (quantized bias, gdnsq_conv2d.py:86-94)
  potential_loss              PotentialLoss arithmetic (gdnsq_loss.py:47-71)
  minmax, fill_r              fused min/max sweep (calibration, zero point); the sign stream, materialised
"""
from __future__ import annotations

import contextlib
import functools
import ctypes
import math
import os

import torch
import torch.distributed as dist

from . import _lib
from ._ext import ext as _ext
from .enums import QNMethod

_MASK64 = (1 << 64) - 1
_i32 = ctypes.c_int32
_byref = ctypes.byref


# ----------------------------------------------------------------------------- RNG stream
_rank_cache = [None, 0]


def _rank() -> int:
    """This process' rank in the default process group (0 without one); cached per group object: dist.get_rank()
    walks Python-side group tables, and every op forward asks."""
    if not dist.is_available():
        return 0
    world = dist.distributed_c10d.GroupMember.WORLD
    if world is None:
        return 0
    if _rank_cache[0] is not world:
        _rank_cache[0], _rank_cache[1] = world, dist.get_rank()
    return _rank_cache[1]


class _Rng:
    """(seed, offset) source for the in-kernel Philox sign stream (SURVEY.md section 8e:
    ranks must draw different streams; every backward call gets a fresh offset).
    The state lives in the compiled binding (the C++ backward nodes draw from it without the GIL); this is its
    Python face, used by the ctypes ops below, the trainer and the tests."""

    # hipGraph capture freezes a launch's (seed, offset) arguments into the graph.  Every backward entry point
    # therefore also takes `offset_dev` (include/mhaq_fq.h): a device-resident uint64 the kernel adds to the
    # host offset.  A capturing trainer installs its word here (device_offset) and advances it by the number
    # of sign streams one step draws at the end of every replay: replay k of a launch captured with host
    # offset c then uses (seed, c + k * stride) -- the offsets the eager loop would have reached -- at 0
    # extra bytes per element.

    @property
    def seed(self):
        return _ext().rng_seed()

    @property
    def offset_base(self):
        return _ext().rng_base()

    def manual_seed(self, seed: int):
        _ext().rng_manual_seed(int(seed) & _MASK64)

    def ensure_seeded(self):
        """Called by every op's FORWARD: the compiled backward nodes cannot ask torch for its seed."""
        E = _ext()
        if E.rng_seed() is None:
            E.rng_manual_seed(torch.initial_seed() & _MASK64)

    def next(self):
        self.ensure_seeded()
        return _ext().rng_next(_rank())      # (seed mixed with the rank, next offset): under a lock in the extension

    def drawn(self) -> int:
        """How many sign streams have been handed out since the last manual_seed."""
        return _ext().rng_drawn()

    def set_drawn(self, n: int) -> None:
        """Move the host counter (a capturing trainer re-aligns it around an eagerly run odd-shaped step)."""
        _ext().rng_set_drawn(int(n))

    @contextlib.contextmanager
    def device_offset(self, base):
        """While active, backward launches add the uint64 at `base` (an int64[1] device tensor) to their host
        offset.  Scoped to the caller (a capturing trainer wraps its step in it) and restored on exit."""
        if base is not None and (base.dtype != torch.int64 or base.numel() != 1 or not base.is_cuda):
            raise ValueError("device_offset: base must be a one-element int64 device tensor")
        E = _ext()
        prev = E.rng_base()
        E.rng_set_base(base)
        try:
            yield base
        finally:
            E.rng_set_base(prev)


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


def _on_device(fn):
    """Run `fn` with the device of its first tensor (or module, or torch.device) argument CURRENT.  The C ABI launches on
    the stream it is handed and expects the calling thread's current device to be that stream's (include/mhaq_fq.h,
    "Devices"); torch ops work on a tensor's device whatever the current one is -- the reference is plain torch ops, so a
    model moved to cuda:1 in a process whose current device is cuda:0 just works there -- and so must these.  Applied to
    the entry points that launch through ctypes (the compiled nodes guard themselves, torch_binding.cpp); backward
    passes run on the autograd engine's thread of their device, which has made it current.  ~0.5 us when the device
    already is the current one."""
    @functools.wraps(fn)
    def run(*args, **kw):
        cands = args if not kw else (*args, *kw.values())      # positional or by keyword (fill_r(..., device=...), x=...)
        d = kw.get("device") if kw else None
        if isinstance(d, int) and not isinstance(d, bool):     # device=1: an index, as torch's factories accept it
            cands = (*cands, torch.device("cuda", d))
        for a in cands:
            if isinstance(a, torch.nn.Module):
                a = next(a.parameters(), None)
            if isinstance(a, torch.Tensor):
                dev = a.device
            elif isinstance(a, torch.device):
                dev = a
            elif isinstance(a, str) and a.startswith("cuda"):
                dev = torch.device(a)
            else:
                continue
            if dev.type == "cuda" and dev.index is not None and dev.index != torch.cuda.current_device():
                with torch.cuda.device(dev):
                    return fn(*args, **kw)
            break
        return fn(*args, **kw)
    return run


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


def _sync_dist_state() -> int:
    """Tell the compiled nodes whether collectives must run (and with what), and return this process' rank: called
    by the forward of every op whose backward may exchange AEWGS statistics or draw a sign stream."""
    E = _ext()
    active = _dist_active()
    E.set_dist_active(active)
    if active and not getattr(_sync_dist_state, "installed", False):
        E.set_allreduce_avg(lambda t: _allreduce_avg_(t))      # looked up per call: this module's current function
        _sync_dist_state.installed = True
    return _rank()


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


@_on_device
def fill_r(n: int, seed: int, offset: int, device) -> torch.Tensor:
    """Materialise the in-kernel sign stream as int8 +-1 (checker use)."""
    out = torch.empty(n, dtype=torch.int8, device=device)
    _lib.check(_lib.lib().mhaq_fq_fill_r(out.data_ptr(), n, seed & _MASK64, offset & _MASK64, _stream()),
               "mhaq_fq_fill_r")
    return out


@_on_device
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


@_on_device
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


@_on_device
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


@_on_device
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
def fake_quant_act_layer(x, log_act_s, log_act_q, act_b, method=QNMethod.STE, r_sign=None, hub_slot=None):
    """Fused NoisyAct training forward (gdnsq_act.py:39-55) in two launches per direction: (y, params) with
    params[5] = {s, zp, lo, hi, qr}.  The autograd node is compiled (torch_binding.cpp: ActLayerFn): forward =
    mhaq_fq_act_fwd, backward = mhaq_fq_act_bwd -- or, with `hub_slot` = (ActGradHub, slot), mhaq_fq_act_bwd_partials,
    leaving the partial sums with the hub, whose single finalize launch serves every quantizer of the pass (act_hub.py).
    AEWGS is not offered here (the reference never builds an AEWGS activation quantizer); use fake_quant_per_tensor."""
    return _act_layer(x, log_act_s, log_act_q, act_b, method, r_sign, hub_slot)[:2]


def _act_layer(x, log_act_s, log_act_q, act_b, method, r_sign=None, hub_slot=None):
    """As fake_quant_act_layer, returning (y, params, params[0:1], params[3:4]): the two views NoisyAct publishes."""
    m = method if method.__class__ is int else _method_value(method)
    if m == QNMethod.AEWGS.value:
        raise NotImplementedError("AEWGS activations go through fake_quant_per_tensor")
    if not (x.is_cuda and x.dtype is torch.float32):
        x = _require_cuda_f32(x, "x", any_dense_layout=True)        # raises (CPU tensor / wrong dtype)
    dev = x.device
    if not (torch.is_tensor(log_act_s) and log_act_s.is_cuda and log_act_s.dtype is torch.float32):
        log_act_s = _scalar(log_act_s, dev, "log_act_s")
    if not (torch.is_tensor(log_act_q) and log_act_q.is_cuda and log_act_q.dtype is torch.float32):
        log_act_q = _scalar(log_act_q, dev, "log_act_q")
    if not (torch.is_tensor(act_b) and act_b.is_cuda and act_b.dtype is torch.float32):
        act_b = _scalar(act_b, dev, "act_b")
    for t, name in ((log_act_s, "log_act_s"), (log_act_q, "log_act_q"), (act_b, "act_b")):
        if t.numel() != 1:
            raise ValueError(f"{name} must have one element for a per-tensor quantizer, got {tuple(t.shape)}")
    if r_sign is not None:
        r_sign = _r_ptr(r_sign, x)
    rng.ensure_seeded()
    hub, slot = (hub_slot[0].id, hub_slot[1]) if hub_slot is not None else (0, 0)
    return _ext().act_layer(x, log_act_s, log_act_q, act_b, m, r_sign, hub, slot, _rank())


_seeded = False
_E = None


def act_layer_routed(x, routed, method: int, ref):
    """NoisyAct's hot path under an ActGradHub: x is a float32 device tensor, `routed` the hub's aliases of the three
    parameters (device tensors by construction), `ref` the module's HubRef."""
    global _seeded, _E
    if not _seeded:
        rng.ensure_seeded()
        _seeded = True           # a seed, once set, is only ever replaced by another seed
        _E = _ext()
    return _E.act_layer(x, routed[0], routed[1], routed[2], method, None, ref.hub.id, ref.slot, _rank())


@_on_device
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
def fake_quant_weight_layer(w, log_wght_s, method=QNMethod.AEWGS, r_sign=None, zp_grad=False, pre=None):
    """Per-channel NoisyConv2d weight path from log_wght_s plus the layer's regulariser input
    lwq = log2(max - min + s) (gdnsq_conv2d.py:71-98, model_helper.py:24-44): returns
    (wq, zp[co,1,..], s[co,1,..], lwq[co]).  Compiled autograd node (torch_binding.cpp: WeightLayerFn) over
    mhaq_fq_wlayer_fwd / _bwd (+ mhaq_fq_pc_aewgs_stats and ONE packed all-reduce for AEWGS under data parallelism).
    `pre` = (wq, s, zp, mx, lwq) of this layer from the model-wide forward launch (multi.py), or None."""
    # a channels_last weight [Co,Ci,kh,kw] is physically [Co][kh][kw][Ci]: every output channel is still one
    # contiguous row, and min / quantize / per-channel sums do not care about the order inside a row
    w = _require_cuda_f32(w, "weight", any_dense_layout=True)
    ls = _require_cuda_f32(log_wght_s, "log_wght_s")
    if ls.numel() != w.shape[0]:
        raise ValueError(f"per-channel log scale must have {w.shape[0]} elements, got {tuple(log_wght_s.shape)}")
    rng.ensure_seeded()
    return _ext().weight_layer(w, ls, _method_value(method), _r_ptr(r_sign, w), bool(zp_grad),
                               None if pre is None else list(pre), _sync_dist_state())


_pt_max = None


def small_pt_layer_supported(w, method) -> bool:
    global _pt_max
    if _pt_max is None:
        _pt_max = int(_lib.lib().mhaq_fq_wlayer_pt_max_elements())
    return w.numel() <= _pt_max and _method_value(method) != QNMethod.AEWGS.value


def fake_quant_weight_layer_pt(w, log_wght_s, method=QNMethod.STE, r_sign=None):
    """PER_TENSOR weight layer small enough for one workgroup (every CIFAR ResNet-20 / RFDN layer): one launch per
    direction from log_wght_s, regulariser input included (compiled node WeightLayerPTFn over mhaq_fq_wlayer_pt_fwd /
    _bwd).  Returns (wq, zp 0-dim, s [1], lwq [1]); see small_pt_layer_supported."""
    w = _require_cuda_f32(w, "weight")
    ls = _scalar(log_wght_s, w.device, "log_wght_s")
    rng.ensure_seeded()
    return _ext().weight_layer_pt(w, ls, _method_value(method), _r_ptr(r_sign, w), _rank())


def fake_quant_weight_layer_ptl(w, log_wght_s, method=QNMethod.AEWGS, r_sign=None):
    """PER_TENSOR weight layer of ANY size, every estimator (AEWGS with its per-position statistics included): the
    streaming form of fake_quant_weight_layer_pt -- min / max sweep, scalar chain, quantizer; backward with the tie-split
    scatter to the minima and maxima and the regulariser gradient (compiled node WeightLayerPTLFn over
    mhaq_fq_wlayer_ptl_fwd / _bwd; gdnsq_conv2d.py:71-98 + model_helper.py:36-37,44).
    Returns (wq, zp 0-dim, s [1], lwq [1])."""
    w = _require_cuda_f32(w, "weight", any_dense_layout=True)
    ls = _scalar(log_wght_s, w.device, "log_wght_s")
    rng.ensure_seeded()
    return _ext().weight_layer_ptl(w, ls, _method_value(method), _r_ptr(r_sign, w), _sync_dist_state())


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


@_on_device
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


@_on_device
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


@_on_device
def fake_quant_weight_pt(w, scale, method=QNMethod.AEWGS, r_sign=None):
    """NoisyConv2d / NoisyLinear per-tensor weight path: returns (wq, zp 0-dim)."""
    w = _require_cuda_f32(w, "weight")
    s = _scalar(scale, w.device, "scale")
    wq, zp = FakeQuantWeightPT.apply(w, s, _method_value(method), _r_ptr(r_sign, w))
    return wq, zp.reshape(())


# ------------------------------------------------------------------ PotentialLoss
def potential_loss(base, las, laq, lws, lwq, state, a_bits, w_bits, p=1, lossless=False, update_state=False):
    """gdnsq_loss.py:47-71 / 129-153 in one launch per direction (mhaq_fq_potential_loss_fwd / _bwd; compiled node
    PotentialLossFn).  Returns (ploss, stats[12]); `state`: float32 device tensor {loss_sum, cnt, t} (see
    include/mhaq_fq.h), the module's device-resident state, advanced in the same launch when `update_state`."""
    if state.dtype != torch.float32 or state.numel() != 3 or not state.is_cuda or not state.is_contiguous():
        raise ValueError("potential_loss: state must be a contiguous float32 device tensor {loss_sum, cnt, t}")
    vecs = [_require_cuda_f32(v, n) for v, n in ((base, "base_loss"), (las, "log_act_s"), (laq, "log_act_q"),
                                                 (lws, "log_wght_s"), (lwq, "log_w"))]
    if vecs[1].numel() != vecs[2].numel() or vecs[3].numel() != vecs[4].numel():
        raise ValueError("potential_loss: scale and range vectors must pair up")
    return _ext().potential_loss(*vecs, state, float(a_bits), float(w_bits), float(p), bool(lossless),
                                 bool(update_state))
