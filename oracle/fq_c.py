"""ctypes face of the plain-C oracle (oracle/fq_ref.c -> oracle/libfq_ref.so)  --  TEST INFRASTRUCTURE ONLY.

Only tests/ and __graft_entry__.build() touch this module; nothing under mhaq_amd/ imports it.  numpy in, numpy
out; the scales are exponentiated here with numpy's exp2 on fp32 (the bits the reference's torch.exp2 produces
for the fixtures' parameters are checked by the golden test through y)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfq_ref.so")
METHODS = {"STE": 0, "EWGS": 1, "AEWGS": 2, "LSQ": 3}
_lib = None


def build() -> str:
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        p, i64, f32, i32 = C.c_void_p, C.c_int64, C.c_float, C.c_int
        L.mhaq_ref_act.restype = i32
        L.mhaq_ref_act.argtypes = [p, p, p, i64, i64, f32, f32, f32, i32, p, p, p, p]
        L.mhaq_ref_weight.restype = i32
        L.mhaq_ref_weight.argtypes = [p, p, p, i64, i64, p, i32, i32, p, p, p, p]
        f64 = C.c_double
        L.mhaq_ref_potential_loss.restype = f64
        L.mhaq_ref_potential_loss.argtypes = [f64, p, p, i64, p, p, i64, f64, f64, f64, i32, f64, f64, f64, p]
        L.mhaq_ref_regulariser_input.restype = i32
        L.mhaq_ref_regulariser_input.argtypes = [p, i64, i64, p, i32, p]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def act(x, g, r_sign, s, qr, b, method):
    """NoisyAct forward + backward: returns dict(y, q, gx, g_log_act_s, g_log_act_q, g_act_b)."""
    x, g = _f32(x), _f32(g)
    r = np.ascontiguousarray(np.asarray(r_sign, dtype=np.int8))
    n = x.size
    d0 = x.shape[0] if x.ndim > 0 else 1
    y, q, gx = (np.empty_like(x) for _ in range(3))
    grads = np.zeros(3, dtype=np.float64)
    rc = lib().mhaq_ref_act(_ptr(x), _ptr(g), _ptr(r), n, d0, float(s), float(qr), float(b), METHODS[method], _ptr(y),
                            _ptr(q), _ptr(gx), _ptr(grads))
    assert rc == 0
    return dict(y=y, q=q, gx=gx, g_log_act_s=grads[0], g_log_act_q=grads[1], g_act_b=grads[2])


def weight(w, G, r_sign, s, per_channel, method):
    """Weight path: returns dict(wq, zp, gw, g_log_wght_s); s: [co] (per-channel) or one element."""
    w, G = _f32(w), _f32(G)
    r = np.ascontiguousarray(np.asarray(r_sign, dtype=np.int8))
    co = w.shape[0]
    row = w.size // co
    s = _f32(s).reshape(-1)
    groups = co if per_channel else 1
    assert s.size == groups
    wq, gw = np.empty_like(w), np.empty_like(w)
    zp = np.empty(groups, dtype=np.float32)
    gls = np.zeros(groups, dtype=np.float64)
    rc = lib().mhaq_ref_weight(_ptr(w), _ptr(G), _ptr(r), co, row, _ptr(s), int(bool(per_channel)), METHODS[method],
                               _ptr(wq), _ptr(zp), _ptr(gw), _ptr(gls))
    assert rc == 0
    return dict(wq=wq, zp=zp, gw=gw, g_log_wght_s=gls)


def potential_loss(base, las, laq, lws, lwq, a_bits, w_bits, t, loss_sum, cnt, lossless=False, p=1):
    """PotentialLoss value (gdnsq_loss.py:47-71): returns (ploss, rloss)."""
    las, laq, lws, lwq = (_f32(v).reshape(-1) for v in (las, laq, lws, lwq))
    r = C.c_double(0.0)
    v = lib().mhaq_ref_potential_loss(float(base), _ptr(las), _ptr(laq), las.size, _ptr(lws), _ptr(lwq), lws.size,
                                      float(a_bits), float(w_bits), float(p), int(bool(lossless)), float(t),
                                      float(loss_sum), float(cnt), C.byref(r))
    return v, r.value


def regulariser_input(w, log_s, per_channel):
    """log2(max - min + 2^log_s) per channel or per tensor (model_helper.py:21-25, 44)."""
    w, ls = _f32(w), _f32(log_s).reshape(-1)
    co = w.shape[0]
    out = np.empty(co if per_channel else 1, dtype=np.float32)
    assert lib().mhaq_ref_regulariser_input(_ptr(w), co, w.size // co, _ptr(ls), int(bool(per_channel)), _ptr(out)) == 0
    return out
