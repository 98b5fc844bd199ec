"""GPU (-m gpu): the AEWGS *apply* arithmetic bit for bit against the reference's golden vectors.

The AEWGS estimator (gdnsq.py:113-147) is group statistics -> delta -> an elementwise gradient scale.  The kernels sum the
statistics in fp64 and round once, torch sums them in fp32: that last-bit difference of three means is the only reason
the golden comparisons of AEWGS input gradients elsewhere carry a propagated bound instead of equality.  Here the
statistics are taken out of the picture: num, e2, me are computed on the CPU in fp32 exactly as gdnsq.py:118-124 does
(torch.mean over the dims where the scale has extent 1, incl. the [1]-shaped-scale quirk of reduce_to_shape), handed to the
kernels through their `stats` / `col_stats` arguments (the path the data-parallel trainer uses after its all-reduce), and
everything downstream -- delta = num / max(e2 - me^2, 1e-3), the clamp at 0.99, gv = gq - gq * g_scale, gv / s, the
straight-through mask -- must give the reference's bits: value-equal gx on every element, value-equal gW on every element
that is not a tied extreme of its group (those also carry a share of a REDUCED gradient).
All 4 activation and all 9 weight AEWGS cases of tests/golden/, through the C ABI."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.golden_util import T, exact_off_extremes, load_cases, value_equal  # noqa: E402

ACT = {k: v for k, v in load_cases("act_cases.npz").items() if "aewgs" in k}
WGT = {k: v for k, v in load_cases("weight_cases.npz").items() if "aewgs" in k}
DEV = "cuda:0"
AEWGS = 2


def _reference_statistics(v, gq, scale_shape):
    """gdnsq.py:118-124 + reduce_to_shape (gdnsq.py:150-152), fp32 on the CPU: [3, ...] = num, e2, me."""
    e = torch.round(v) - v
    num_full = gq.sign() * e
    dims = tuple(i for i, n in enumerate(scale_shape) if n == 1)
    return torch.stack([torch.mean(t, dim=dims, keepdim=True) for t in (num_full, e.square(), e)])


def _signs(c, key="r"):
    return torch.from_numpy(c[key].astype(np.int8)).to(DEV)


def test_the_fixture_sets_are_complete():
    assert len(ACT) == 4 and len(WGT) == 9


@pytest.mark.parametrize("name", sorted(ACT))
def test_activation_aewgs_input_gradient_equals_the_reference_given_its_statistics(name):
    from mhaq_amd import _lib
    L = _lib.lib()
    c = ACT[name]
    x, g = T(c["x"]), T(c["g"])
    s = torch.exp2(T(c["log_act_s"]).reshape(1))
    qr = torch.exp2(T(c["log_act_q"]).reshape(1))
    b = T(c["act_b"]).reshape(1)
    hi = b + qr - s
    v = (torch.clamp(x, b, hi) - b) / s
    stats = _reference_statistics(v, g * s, (1,))              # [3, 1, C, H, W]: means over dim 0 only
    period = x.numel() // x.shape[0]
    assert stats.numel() == 3 * period
    xd, gd, sd, bd, hd = (t.to(DEV).contiguous() for t in (x, g, s, b, hi))
    st = stats.reshape(3, period).contiguous().to(DEV)
    gx = torch.empty_like(xd)
    grads = torch.empty(5, device=DEV)
    nb = L.mhaq_fq_pt_bwd_workspace_bytes(x.numel())
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    r = _signs(c)
    rc = L.mhaq_fq_pt_bwd(xd.data_ptr(), gd.data_ptr(), gx.data_ptr(), x.numel(), sd.data_ptr(), bd.data_ptr(),
                          bd.data_ptr(), hd.data_ptr(), AEWGS, st.data_ptr(), period, r.data_ptr(), 0, 0, None, 0,
                          grads.data_ptr(), ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert value_equal(gx.cpu().numpy(), c["gx"]), f"{name}: AEWGS gx differs from the reference given its own statistics"


@pytest.mark.parametrize("name", sorted(WGT))
def test_weight_aewgs_gradient_equals_the_reference_given_its_statistics(name):
    from mhaq_amd import _lib
    L = _lib.lib()
    c = WGT[name]
    w, G = T(c["w"]), T(c["G"])
    pc = bool(c["per_channel"])
    s = torch.exp2(T(c["log_wght_s"]))
    dims = tuple(range(1, w.dim()))
    zp = w.amin(dims, keepdim=True) if pc else w.amin()
    v = (w - zp) / s
    stats = _reference_statistics(v, G * s, tuple(s.shape))
    wd, Gd = w.to(DEV).contiguous(), G.to(DEV).contiguous()
    gw = torch.empty_like(wd)
    r = _signs(c)
    stream = torch.cuda.current_stream().cuda_stream
    if pc:
        co, row = w.shape[0], w.numel() // w.shape[0]
        sd, zd = s.reshape(co).contiguous().to(DEV), zp.reshape(co).contiguous().to(DEV)
        st = stats.reshape(3, co).contiguous().to(DEV)
        gs = torch.empty(co, device=DEV)
        rc = L.mhaq_fq_pc_bwd(wd.data_ptr(), Gd.data_ptr(), gw.data_ptr(), gs.data_ptr(), sd.data_ptr(), zd.data_ptr(),
                              co, row, AEWGS, st.data_ptr(), None, r.data_ptr(), 0, 0, None, stream)
        assert rc == 0
        # ... and the layer entry point (what NoisyConv2d's compiled node calls after the all-reduce of the statistics)
        gw2 = torch.empty_like(wd)
        gls = torch.empty(co, device=DEV)
        mx = w.amax(dims).reshape(co).contiguous().to(DEV)
        rc = L.mhaq_fq_wlayer_bwd(wd.data_ptr(), Gd.data_ptr(), gw2.data_ptr(), gls.data_ptr(), sd.data_ptr(),
                                  zd.data_ptr(), mx.data_ptr(), None, co, row, AEWGS, st.data_ptr(), None, r.data_ptr(),
                                  0, 0, None, stream)
        assert rc == 0
        assert exact_off_extremes(gw2.cpu().numpy(), c["gw"], c["w"], True, also_max=True), f"{name}: wlayer gW"
    else:
        period = w.numel() // w.shape[0]                 # the [1]-shaped scale: statistics per position, over dim 0
        st = stats.reshape(3, period).contiguous().to(DEV)
        sd = s.reshape(1).to(DEV)
        zd = zp.reshape(1).to(DEV)
        ninf = torch.tensor([-float("inf")], device=DEV)
        pinf = torch.tensor([float("inf")], device=DEV)
        grads = torch.empty(5, device=DEV)
        nb = L.mhaq_fq_pt_bwd_workspace_bytes(w.numel())
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        rc = L.mhaq_fq_pt_bwd(wd.data_ptr(), Gd.data_ptr(), gw.data_ptr(), w.numel(), sd.data_ptr(), zd.data_ptr(),
                              ninf.data_ptr(), pinf.data_ptr(), AEWGS, st.data_ptr(), period, r.data_ptr(), 0, 0, None,
                              1, grads.data_ptr(), ws.data_ptr(), nb, stream)
        assert rc == 0
    assert exact_off_extremes(gw.cpu().numpy(), c["gw"], c["w"], pc), \
        f"{name}: AEWGS gW off the tied minima differs from the reference given its own statistics"


def test_nan_statistics_propagate_like_clamp_max():
    """torch.clamp_max passes NaN on (gdnsq.py:139): a NaN statistic must reach every gradient of its group, not be
    replaced by the 0.99 bound."""
    from mhaq_amd import _lib
    L = _lib.lib()
    torch.manual_seed(3)
    co, row = 4, 64
    w = (torch.randn(co, row) * 0.1).to(DEV)
    G = torch.randn(co, row).to(DEV)
    s = torch.full((co,), 2.0 ** -5, device=DEV)
    zp = w.amin(1).contiguous()
    stats = torch.tensor([[0.1] * co, [0.09] * co, [0.0] * co], device=DEV)
    stats[0, 2] = float("nan")
    gw = torch.empty_like(w)
    gs = torch.empty(co, device=DEV)
    rc = L.mhaq_fq_pc_bwd(w.data_ptr(), G.data_ptr(), gw.data_ptr(), gs.data_ptr(), s.data_ptr(), zp.data_ptr(), co, row,
                          AEWGS, stats.data_ptr(), None, None, 5, 1, None, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert torch.isnan(gw[2]).all() and torch.isfinite(gw[[0, 1, 3]]).all()
    # ... and a NaN in the DENOMINATOR's statistics with a finite numerator: clamp_min(e2 - me^2, 1e-3) is NaN in torch
    # (gdnsq.py:132), not the bound
    for row_of_stats in (1, 2):
        stats2 = torch.tensor([[0.1] * co, [0.09] * co, [0.0] * co], device=DEV)
        stats2[row_of_stats, 1] = float("nan")
        rc = L.mhaq_fq_pc_bwd(w.data_ptr(), G.data_ptr(), gw.data_ptr(), gs.data_ptr(), s.data_ptr(), zp.data_ptr(), co,
                              row, AEWGS, stats2.data_ptr(), None, None, 5, 1, None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        assert torch.isnan(gw[1]).all() and torch.isfinite(gw[[0, 2, 3]]).all(), row_of_stats


# ---------------------------------------------------------------------------------------------------------------
# Round 4: the remaining three entry points that take EXTERNAL statistics -- the facade of QNAEWGS.backward itself
# (mhaq_fq_noise_bwd, gdnsq.py:113-147: per-group statistics, `period == 0`, and the per-position form of the [1]-shaped
# scale, `period > 0`), the streaming PER_TENSOR layer (mhaq_fq_wlayer_ptl_bwd) and the grouped per-channel backward
# (mhaq_fq_wlayer_bwd_group) -- on the same 13 golden cases, the same way: the reference's fp32 statistics in, the
# reference's bits out.
def _noise_bwd(v, gq, groups, length, stats, period, r):
    """QNAEWGS.backward's grad_input (gdnsq.py:141) from the facade kernel, given the statistics."""
    from mhaq_amd import _lib
    L = _lib.lib()
    vd, gd = v.contiguous().to(DEV), gq.contiguous().to(DEV)
    gv = torch.empty_like(vd)
    gs = torch.empty(groups, device=DEV)
    nb = L.mhaq_fq_noise_bwd_workspace_bytes(groups, length)
    ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=DEV)
    rc = L.mhaq_fq_noise_bwd(vd.data_ptr(), gd.data_ptr(), gv.data_ptr(), gs.data_ptr(), groups, length, AEWGS,
                             stats.data_ptr(), period, r.data_ptr(), 0, 0, None, ws.data_ptr(), nb,
                             torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    return gv.cpu()


@pytest.mark.parametrize("name", sorted(ACT))
def test_facade_noise_backward_gives_the_reference_activation_gradient_given_its_statistics(name):
    c = ACT[name]
    x, g = T(c["x"]), T(c["g"])
    s = torch.exp2(T(c["log_act_s"]).reshape(1))
    qr = torch.exp2(T(c["log_act_q"]).reshape(1))
    b = T(c["act_b"]).reshape(1)
    hi = b + qr - s
    v = (torch.clamp(x, b, hi) - b) / s
    gq = g * s                                                  # dL/dq = dL/dnoise: the Function's grad_output
    period = x.numel() // x.shape[0]
    stats = _reference_statistics(v, gq, (1,)).reshape(3, period).contiguous().to(DEV)
    g_noise = _noise_bwd(v, gq, 1, x.numel(), stats, period, _signs(c))
    # the rest of the reference's graph on the CPU, op for op: q = v + noise -> +, v = v1 / s -> /, clamp -> mask
    gx = ((gq + g_noise) / s) * ((x >= b) & (x <= hi))
    assert value_equal(gx.numpy(), c["gx"]), f"{name}: facade AEWGS grad_input differs from the reference's bits"


@pytest.mark.parametrize("name", sorted(WGT))
def test_facade_noise_backward_gives_the_reference_weight_gradient_given_its_statistics(name):
    c = WGT[name]
    w, G = T(c["w"]), T(c["G"])
    pc = bool(c["per_channel"])
    s = torch.exp2(T(c["log_wght_s"]))
    dims = tuple(range(1, w.dim()))
    zp = w.amin(dims, keepdim=True) if pc else w.amin()
    v = (w - zp) / s
    gq = G * s
    st = _reference_statistics(v, gq, tuple(s.shape))
    if pc:                                                       # `period == 0`: one statistics triple per scale group
        co, row = w.shape[0], w.numel() // w.shape[0]
        g_noise = _noise_bwd(v, gq, co, row, st.reshape(3, co).contiguous().to(DEV), 0, _signs(c))
    else:                                                        # `period > 0`: per position, means over dim 0
        period = w.numel() // w.shape[0]
        g_noise = _noise_bwd(v, gq, 1, w.numel(), st.reshape(3, period).contiguous().to(DEV), period, _signs(c))
    gw = (gq + g_noise) / s
    assert exact_off_extremes(gw.numpy(), c["gw"], c["w"], pc), f"{name}: facade AEWGS grad_input (weights)"


@pytest.mark.parametrize("name", sorted(k for k in WGT if not bool(WGT[k]["per_channel"])))
def test_streaming_per_tensor_layer_aewgs_gradient_equals_the_reference_given_its_statistics(name):
    """mhaq_fq_wlayer_ptl_bwd with `col_stats` (what the data-parallel trainer hands it after the all-reduce).  aux[7]
    carries the reference's own scale bits (the device's exp2 may differ from the host's by an ulp)."""
    from mhaq_amd import _lib
    L = _lib.lib()
    c = WGT[name]
    w, G = T(c["w"]), T(c["G"])
    s = torch.exp2(T(c["log_wght_s"])).reshape(())
    zp, mx = w.amin(), w.amax()
    v = (w - zp) / s
    period = w.numel() // w.shape[0]
    st = _reference_statistics(v, G * s, (1,)).reshape(3, period).contiguous().to(DEV)
    inf = float("inf")
    aux = torch.tensor([float(s), float(zp), float(mx), float(torch.log2((mx - zp) + s)), -inf, inf, float(mx)],
                       dtype=torch.float32, device=DEV)
    wd, Gd = w.contiguous().to(DEV), G.contiguous().to(DEV)
    gw = torch.empty_like(wd)
    gls = torch.empty(1, device=DEV)
    nb = L.mhaq_fq_wlayer_ptl_workspace_bytes(w.numel())
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    rc = L.mhaq_fq_wlayer_ptl_bwd(wd.data_ptr(), Gd.data_ptr(), gw.data_ptr(), gls.data_ptr(), aux.data_ptr(), None,
                                  w.numel(), AEWGS, st.data_ptr(), period, _signs(c).data_ptr(), 0, 0, None,
                                  ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert exact_off_extremes(gw.cpu().numpy(), c["gw"], c["w"], False, also_max=True), f"{name}: streaming layer gW"


class _Desc(ctypes.Structure):          # mhaq_wlayer_desc (include/mhaq_fq.h)
    _fields_ = [("w", ctypes.c_void_p), ("log_s", ctypes.c_void_p), ("G", ctypes.c_void_p), ("g_lwq", ctypes.c_void_p),
                ("co", ctypes.c_int64), ("row", ctypes.c_int64), ("elem_offset", ctypes.c_int64),
                ("chan_offset", ctypes.c_int64)]


def _pc_case_tensors(c):
    w, G = T(c["w"]), T(c["G"])
    s = torch.exp2(T(c["log_wght_s"]))
    dims = tuple(range(1, w.dim()))
    zp, mx = w.amin(dims, keepdim=True), w.amax(dims, keepdim=True)
    st = _reference_statistics((w - zp) / s, G * s, tuple(s.shape))
    co = w.shape[0]
    aux = torch.stack([s.reshape(co), zp.reshape(co), mx.reshape(co), torch.log2((mx - zp) + s).reshape(co)])
    return w, G, aux, st.reshape(3, co)


def _run_group(cases):
    """One mhaq_fq_wlayer_bwd_group launch over the per-channel layers `cases`, statistics supplied."""
    from mhaq_amd import _lib
    L = _lib.lib()
    parts = [_pc_case_tensors(c) for c in cases]
    keep, descs, eo, cho = [], (_Desc * len(parts))(), 0, 0
    for i, (w, G, _, _) in enumerate(parts):
        wd, Gd = w.contiguous().to(DEV), G.contiguous().to(DEV)
        keep += [wd, Gd]
        co, row = w.shape[0], w.numel() // w.shape[0]
        descs[i] = _Desc(wd.data_ptr(), None, Gd.data_ptr(), None, co, row, eo, cho)
        eo, cho = eo + co * row, cho + co
    table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(DEV)
    aux = torch.cat([p[2] for p in parts], dim=1).contiguous().to(DEV)           # [4][group_co]
    stats = torch.cat([p[3] for p in parts], dim=1).contiguous().to(DEV)         # [3][group_co]
    gw = torch.empty(eo, device=DEV)
    gls = torch.empty(cho, device=DEV)
    max_row = max(p[0].numel() // p[0].shape[0] for p in parts)
    rc = L.mhaq_fq_wlayer_bwd_group(table.data_ptr(), len(parts), cho, max_row, aux.data_ptr(), cho, gw.data_ptr(),
                                    gls.data_ptr(), AEWGS, stats.data_ptr(), 11, 1, None,
                                    torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    out, e = [], 0
    gwc = gw.cpu()
    for w, _, _, _ in parts:
        out.append(gwc[e:e + w.numel()].reshape(w.shape).numpy())
        e += w.numel()
    return out


PC_NAMES = sorted(k for k in WGT if bool(WGT[k]["per_channel"]))


@pytest.mark.parametrize("name", PC_NAMES)
def test_grouped_backward_aewgs_gradient_equals_the_reference_given_its_statistics(name):
    """A group of one layer: gW does not depend on the random signs (only the scale gradient does), so the in-kernel
    stream of this entry point (it takes no explicit signs) is irrelevant to the comparison."""
    c = WGT[name]
    (gw,) = _run_group([c])
    assert exact_off_extremes(gw, c["gw"], c["w"], True, also_max=True), f"{name}: grouped backward gW"


def test_one_group_launch_over_all_per_channel_cases_equals_the_reference_given_its_statistics():
    """All per-channel AEWGS cases (rows of 36 and 144 floats, tied minima, saturated clamps) as ONE group: one launch,
    one packed [3][group_co] statistics block -- the data-parallel trainer's form."""
    cases = [WGT[k] for k in PC_NAMES]
    for c, gw in zip(cases, _run_group(cases)):
        assert exact_off_extremes(gw, c["gw"], c["w"], True, also_max=True)
