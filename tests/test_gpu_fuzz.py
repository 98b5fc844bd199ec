"""GPU (-m gpu): seeded fuzzing of the fused layer entry points against the eager oracle run on the same
device.  Random ranks / shapes (odd and even lengths, so both the float4 and the dword kernels run), random
non-power-of-two scales, clamp ranges from heavy clipping to inverted (qr < s), inputs sitting exactly on the
bounds and on .5 rounding ties, +-inf inputs, tied minima / maxima and constant rows for the weights.

Bar: elementwise outputs (y, wq, lwq, gx) bit-exact; reduced gradients within 1e-6 * sum|terms| (the yardsticks of
oracle/fq_closed_form.py); AEWGS adds the propagated slack of its group means (fp64 here, fp32 in torch:
tests/aewgs_bound.py) -- derived per case, no blanket factor."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fq_closed_form as CF  # noqa: E402
from oracle import fq_eager as O  # noqa: E402
from tests.aewgs_bound import aewgs_gx_bound, aewgs_weight_slacks  # noqa: E402
from tests.golden_util import bit_equal, exact_off_extremes, off_extremes_mask, value_equal  # noqa: E402


def _ulp_close(a, b, n=2):
    """|a - b| <= n ulps of the larger magnitude (fp32), elementwise; NaN never passes"""
    a, b = a.detach().double(), b.detach().double()
    return bool(((a - b).abs() <= n * 2.0 ** -23 * torch.maximum(a.abs(), b.abs()) + 1e-45).all())

DEV = "cuda:0"
LN2 = math.log(2.0)
# MHAQ_FUZZ_SCALE=50 runs 50x the seeds (a one-off soak; the default keeps the suite at a few seconds)
_K = int(__import__("os").environ.get("MHAQ_FUZZ_SCALE", "1"))


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from mhaq_amd import _lib, ops
    _lib.lib()
    return ops


def P(v, grad=True):
    return torch.tensor([float(v)], device=DEV, requires_grad=grad)


def _shape(rng, max_elems):
    nd = int(rng.integers(1, 5))
    while True:
        shp = tuple(int(v) for v in rng.integers(1, 14, size=nd))
        if 1 <= int(np.prod(shp)) <= max_elems:
            return shp


@pytest.mark.parametrize("seed", range(32 * _K))
def test_fuzz_act_layer(ops, seed):
    rng = np.random.default_rng(7000 + seed)
    gen = torch.Generator().manual_seed(7000 + seed)
    shape = _shape(rng, 20000)
    method = ["STE", "LSQ", "EWGS"][seed % 3]
    log_s = float(rng.uniform(-9.0, 1.0))
    log_q = log_s + float(rng.uniform(-1.5, 7.0))          # below log_s: inverted bounds (hi < lo)
    b = float(rng.normal() * 2.0)
    signed = bool(rng.random() < 0.7)
    s, qr = 2.0 ** log_s, 2.0 ** log_q
    x = torch.randn(*shape, generator=gen) * float(np.exp(rng.uniform(-2, 2))) + b + 0.5 * qr
    flat = x.flatten()
    k = flat.numel()
    if k >= 8:                                             # exact bound hits, .5 ties, infinities
        flat[0], flat[1] = b, (b + qr) - s
        flat[2] = b + 2.5 * s
        flat[3] = b + 3.5 * s
        if rng.random() < 0.5:
            flat[4], flat[5] = float("inf"), float("-inf")
    g = torch.randn(*shape, generator=gen)
    r = torch.randint(0, 2, shape, generator=gen).float() - 0.5
    x, g, r = x.to(DEV), g.to(DEV), r.to(DEV)
    xr = x.clone().requires_grad_(True)
    ls_r, lq_r, b_r = P(log_s), P(log_q), P(b, signed)
    y_r, _ = O.act_fake_quant(xr, ls_r, lq_r, b_r, r=r, method=method)
    y_r.backward(g)
    xg = x.clone().requires_grad_(True)
    ls, lq, bb = P(log_s), P(log_q), P(b, signed)
    y, params = ops.fake_quant_act_layer(xg, ls, lq, bb, method, r_sign=(r * 2).to(torch.int8))
    y.backward(g)
    assert bit_equal(y.detach().cpu().numpy(), y_r.detach().cpu().numpy())
    assert value_equal(xg.grad.cpu().numpy(), xr.grad.cpu().numpy())
    sd, qd = torch.exp2(ls.detach()), torch.exp2(lq.detach())
    hi = (bb.detach() + qd) - sd
    cf = CF.per_tensor(x.cpu(), g.cpu(), r.cpu(), sd.cpu(), bb.detach().cpu(), bb.detach().cpu(), hi.cpu(), method)
    abs_g, abs_s = float(cf["abs_g"]), float(cf["abs_s"])
    if not (math.isfinite(abs_g) and math.isfinite(abs_s)):
        return                                             # an infinity reached a sum: nothing finite to compare
    assert abs(float(ls.grad) - float(ls_r.grad)) <= 1e-6 * (abs_s + abs_g) * float(sd) * LN2 + 1e-30
    assert abs(float(lq.grad) - float(lq_r.grad)) <= 1e-6 * abs_g * float(qd) * LN2 + 1e-30
    if signed:
        assert abs(float(bb.grad) - float(b_r.grad)) <= 1e-6 * abs_g + 1e-30
    else:
        assert bb.grad is None and b_r.grad is None


@pytest.mark.parametrize("seed", range(32 * _K))
def test_fuzz_weight_layer(ops, seed):
    rng = np.random.default_rng(9000 + seed)
    gen = torch.Generator().manual_seed(9000 + seed)
    method = ["STE", "LSQ", "EWGS", "AEWGS"][seed % 4]
    co = int(rng.integers(1, 34))
    if seed % 8 == 7:
        tail = (int(rng.integers(3000, 5200)),)            # a long row
    else:
        nd = int(rng.integers(1, 4))
        tail = tuple(int(v) for v in rng.integers(1, 12, size=nd))
    shape = (co,) + tail
    row = int(np.prod(tail))
    dims = tuple(range(1, len(shape)))
    w = torch.randn(*shape, generator=gen) * float(np.exp(rng.uniform(-3, 1)))
    w2 = w.view(co, row)
    if row >= 6:
        for c in range(0, co, 3):                          # tied minima and maxima
            w2[c, [0, row // 2]] = w2[c].min() - 0.01
            w2[c, [1, row - 1]] = w2[c].max() + 0.02
    if co >= 2 and row >= 2 and rng.random() < 0.5:
        w2[co - 1, :] = 0.125                              # a constant row: max == min, every element tied
    G = torch.randn(*shape, generator=gen)
    h = torch.randn(co, generator=gen)
    r = torch.randint(0, 2, shape, generator=gen).float() - 0.5
    span = (w.amax(dims) - w.amin(dims)).clamp_min(1e-3)
    ls0 = (torch.log2(span / float(rng.choice([3.0, 15.0, 255.0]))) + 0.3 * torch.randn(co, generator=gen))
    ls0 = ls0.reshape([co] + [1] * len(dims))
    w, G, h, r, ls0 = (t.to(DEV) for t in (w, G, h, r, ls0))
    wr, lsr = w.clone().requires_grad_(True), ls0.clone().requires_grad_(True)
    wq_r, _, zp_r = O.weight_fake_quant(wr, lsr, True, method, r=r)
    lwq_r = torch.log2(wr.amax(dims) - wr.amin(dims) + torch.exp2(lsr.ravel()))
    ((wq_r * G).sum() + (lwq_r * h).sum()).backward()
    wg, lsg = w.clone().requires_grad_(True), ls0.clone().requires_grad_(True)
    wq, zp, s, lwq = ops.fake_quant_weight_layer(wg, lsg, method, r_sign=(r * 2).to(torch.int8))
    ((wq * G).sum() + (lwq * h).sum()).backward()
    assert torch.equal(zp.ravel(), zp_r.detach().ravel())
    assert bit_equal(wq.detach().cpu().numpy(), wq_r.detach().cpu().numpy())
    assert bit_equal(lwq.detach().cpu().numpy(), lwq_r.detach().cpu().numpy())
    cf = CF.per_channel(w.cpu(), G.cpu(), r.cpu(), s.detach().cpu().reshape(-1), method)
    bshape = [-1] + [1] * len(dims)
    habs = np.abs(h.cpu().numpy())
    # the regulariser's share t = h / (u ln2) enters gw at the tied extremes and d/dlog_s: part of the yardstick
    u = (w.amax(dims) - w.amin(dims) + s.detach().reshape(-1)).cpu().numpy()
    t = habs / (u * LN2)
    abs_g = (cf["abs_g"].numpy() + 4 * t).reshape(bshape)
    # AEWGS: 1e-6 * sum|terms| + the propagated slack of its group means (tests/aewgs_bound.py::aewgs_weight_slacks), which
    # grows where e2 - me^2 cancels in a short row; rounds 2-5 carried a blanket 5e-6 here
    sl_gw, sl_ls = aewgs_weight_slacks(w, G, s.detach().reshape(-1), True) if method == "AEWGS" else (0.0, 0.0)
    if method != "AEWGS":     # elementwise everywhere but at the row extremes (amin / amax backward shares)
        assert exact_off_extremes(wg.grad.cpu().numpy(), wr.grad.cpu().numpy(), w.cpu().numpy(), True, also_max=True)
    err = np.abs(wg.grad.cpu().numpy() - wr.grad.cpu().numpy())
    assert np.all(err <= 1e-6 * (abs_g + np.abs(wr.grad.cpu().numpy())) + sl_gw), float(err.max())
    sv = s.detach().cpu().numpy().reshape(-1)
    yard = (cf["abs_s"].numpy() + 4 * t) * LN2 * sv * 2
    errs = np.abs(lsg.grad.cpu().numpy().reshape(-1) - lsr.grad.cpu().numpy().reshape(-1))
    assert np.all(errs <= 1e-6 * yard + sl_ls + 1e-9), float((errs / (yard + 1e-30)).max())


def _assert_aewgs_per_tensor(gw, gw_ref, gls, gls_ref, w, G, sd, extra_abs=0.0, also_max=False):
    """The AEWGS legs of the per-tensor weight fuzz tests against the eager oracle (whose three per-position means over
    dim 0 torch sums in fp32, the kernels in fp64): the propagated bound of tests/aewgs_bound.py instead of a flat tolerance.
      off the tied extremes   |gW - ref| <= |G| (|e| d_delta + 1e-6)                                   [elementwise]
      at the extremes         + 1e-6 sum|terms| + the summed bound: they carry a share of sum(G - gv/s)
      d/dlog_s                1e-6 sum|terms| + sum bound |v|  (d/ds holds -sum gv (v/s), gv / s moves by the bound)"""
    wc, Gc = w.detach().cpu(), G.detach().cpu()
    v = (wc - wc.min()) / sd
    bound = aewgs_gx_bound(v, Gc, (0,))
    err = (gw.detach().cpu().double() - gw_ref.detach().cpu().double()).abs()
    off = torch.from_numpy(off_extremes_mask(wc.numpy(), False, also_max=also_max))
    assert bool((err[off] <= bound[off] + 1e-30).all()), float((err - bound)[off].max())
    abs_g = float(Gc.abs().double().sum()) * 2 + extra_abs
    assert bool((err <= bound + 1e-6 * abs_g + float(bound.sum())).all())
    q = v.round()
    yard = (float((Gc * q).abs().double().sum()) * 2 + abs_g) * sd * LN2
    slack = float((bound * v.abs().double()).sum()) * sd * LN2
    assert abs(float(gls) - float(gls_ref)) <= 1e-6 * yard + slack + 1e-9


@pytest.mark.parametrize("seed", range(12 * _K))
def test_fuzz_per_tensor_weight(ops, seed):
    """PER_TENSOR weight path on both sides of the one-workgroup limit (mhaq_fq_wlayer_pt_* vs minmax + pt_* +
    tie_scatter), all estimators, global-minimum ties."""
    from mhaq_amd import _lib
    rng = np.random.default_rng(11000 + seed)
    gen = torch.Generator().manual_seed(11000 + seed)
    method = ["STE", "LSQ", "EWGS", "AEWGS"][seed % 4]
    limit = int(_lib.lib().mhaq_fq_wlayer_pt_max_elements())
    if seed % 3 == 2:
        shape = (int(rng.integers(66, 90)), int(rng.integers(1000, 1100)))    # above the one-workgroup limit
        assert shape[0] * shape[1] > limit
    else:
        shape = (int(rng.integers(1, 40)),) + tuple(int(v) for v in rng.integers(1, 9, size=int(rng.integers(1, 4))))
    w = torch.randn(*shape, generator=gen) * 0.1
    if w.numel() >= 4:
        wf = w.flatten()
        wf[[0, w.numel() - 1]] = wf.min() - 0.01                               # tied global minimum
    G = torch.randn(*shape, generator=gen)
    r = torch.randint(0, 2, shape, generator=gen).float() - 0.5
    ls0 = torch.log2((w.max() - w.min()).clamp_min(1e-3) / 15.0).reshape(1) + float(rng.normal() * 0.3)
    w, G, r, ls0 = (t.to(DEV) for t in (w, G, r, ls0))
    wr, lsr = w.clone().requires_grad_(True), ls0.clone().requires_grad_(True)
    wq_r, _, zp_r = O.weight_fake_quant(wr, lsr, False, method, r=r)
    (wq_r * G).sum().backward()
    wg, lsg = w.clone().requires_grad_(True), ls0.clone().requires_grad_(True)
    r8 = (r * 2).to(torch.int8)
    if ops.small_pt_layer_supported(wg, method):
        wq, zp, s, _ = ops.fake_quant_weight_layer_pt(wg, lsg, method, r_sign=r8)
    else:
        wq, zp = ops.fake_quant_weight_pt(wg, torch.exp2(lsg), method, r_sign=r8)
    (wq * G).sum().backward()
    assert float(zp.detach()) == float(zp_r.detach())
    assert bit_equal(wq.detach().cpu().numpy(), wq_r.detach().cpu().numpy())
    abs_g = float(G.abs().double().sum()) * 2
    sd = float(torch.exp2(ls0))
    if method == "AEWGS":
        _assert_aewgs_per_tensor(wg.grad, wr.grad, lsg.grad, lsr.grad, w, G, sd)
        return
    assert exact_off_extremes(wg.grad.cpu().numpy(), wr.grad.cpu().numpy(), w.cpu().numpy(), False)
    err = (wg.grad - wr.grad).abs().max()
    assert float(err) <= 1e-6 * abs_g, float(err)
    q = ((w - w.min()) / sd).round()
    yard = (float((G * q).abs().double().sum()) * 2 + abs_g) * sd * LN2
    assert abs(float(lsg.grad) - float(lsr.grad)) <= 1e-6 * yard + 1e-9


@pytest.mark.parametrize("seed", range(12 * _K))
def test_fuzz_streaming_per_tensor_layer(ops, seed):
    """mhaq_fq_wlayer_ptl_* (PER_TENSOR layers of any size, every estimator) with the regulariser input and its
    gradient: random shapes on both sides of 64 K elements, ragged element counts, tied global minima and maxima,
    contiguous and channels_last weights."""
    rng = np.random.default_rng(15000 + seed)
    gen = torch.Generator().manual_seed(15000 + seed)
    method = ["STE", "LSQ", "EWGS", "AEWGS"][seed % 4]
    if seed % 3 == 0:
        shape = (int(rng.integers(3, 40)), int(rng.integers(2, 30)), 3, 3)
    elif seed % 3 == 1 and seed % 4 != 1:
        shape = (int(rng.integers(60, 130)), int(rng.integers(900, 1200)))          # above one workgroup's 64 K
    elif seed % 3 == 1:
        shape = (int(rng.integers(40, 70)), int(rng.integers(8, 20)), 3, 3)         # LSQ, 4-D: the channels_last leg
    else:
        shape = (int(rng.integers(1, 30)),) + tuple(int(v) for v in rng.integers(1, 11, size=int(rng.integers(1, 4))))
    w = torch.randn(*shape, generator=gen) * 0.1
    wf = w.flatten()
    if w.numel() >= 6:
        wf[[0, w.numel() - 1]] = wf.min() - 0.01                                     # tied global minimum ...
        wf[[1, w.numel() // 2]] = wf.max() + 0.02                                    # ... and maximum
    G = torch.randn(*shape, generator=gen)
    h = torch.randn(1, generator=gen)
    r = torch.randint(0, 2, shape, generator=gen).float() - 0.5
    ls0 = torch.log2((w.max() - w.min()).clamp_min(1e-3) / 15.0).reshape(1) + float(rng.normal() * 0.3)
    w, G, h, r, ls0 = (t.to(DEV) for t in (w, G, h, r, ls0))
    wr, lsr = w.clone().requires_grad_(True), ls0.clone().requires_grad_(True)
    wq_r, _, zp_r = O.weight_fake_quant(wr, lsr, False, method, r=r)
    lwq_r = torch.log2(wr.amax() - wr.amin() + torch.exp2(lsr.ravel()))
    ((wq_r * G).sum() + (lwq_r * h).sum()).backward()
    wg = w.clone()
    if len(shape) == 4 and method == "LSQ":        # explicit signs are given in the memory order of the weight:
        wg = wg.contiguous(memory_format=torch.channels_last)       # only the sign-free estimator changes layout here
    wg, lsg = wg.requires_grad_(True), ls0.clone().requires_grad_(True)
    wq, zp, s, lwq = ops.fake_quant_weight_layer_ptl(wg, lsg, method, r_sign=(r * 2).to(torch.int8))
    ((wq * G).sum() + (lwq * h).sum()).backward()
    assert float(zp.detach()) == float(zp_r.detach())
    assert bit_equal(wq.detach().cpu().numpy(), wq_r.detach().cpu().numpy())
    assert bit_equal(lwq.detach().cpu().numpy(), lwq_r.detach().cpu().numpy())
    abs_g = float(G.abs().double().sum()) * 2 + abs(float(h)) * 4
    sd = float(torch.exp2(ls0))
    if method == "AEWGS":      # per-position means over dim 0 in fp64 here, fp32 in torch: the propagated bound
        _assert_aewgs_per_tensor(wg.grad, wr.grad, lsg.grad, lsr.grad, w, G, sd, extra_abs=abs(float(h)) * 4,
                                 also_max=True)
        return
    assert exact_off_extremes(wg.grad.cpu().numpy(), wr.grad.cpu().numpy(), w.cpu().numpy(), False, also_max=True)
    err = (wg.grad - wr.grad).abs().max()
    assert float(err) <= 1e-6 * abs_g, float(err)
    q = ((w - w.min()) / sd).round()
    yard = (float((G * q).abs().double().sum()) * 2 + abs_g) * sd * LN2
    assert abs(float(lsg.grad) - float(lsr.grad)) <= 1e-6 * yard + 1e-9


@pytest.mark.parametrize("seed", range(24 * _K))
def test_fuzz_quantizer_facade(seed):
    """Quantizer.quantize / dequantize (the two-method facade over the stand-alone QN* kernels) for every scale
    layout the reference's layers produce: [1], 0-dim, [C,1,..] and per-element, random estimators and bounds."""
    import mhaq_amd as M
    from mhaq_amd import ops_generic as G
    rng = np.random.default_rng(13000 + seed)
    gen = torch.Generator().manual_seed(13000 + seed)
    method = ["STE", "LSQ", "EWGS", "AEWGS"][seed % 4]
    kind = ["one", "zero_dim", "per_channel", "per_element"][(seed // 4) % 4]
    if kind == "per_element":
        shape = (int(rng.integers(1, 70)),)
    else:
        shape = (int(rng.integers(2, 12)),) + tuple(int(v) for v in rng.integers(1, 7, size=int(rng.integers(1, 4))))
    x = torch.randn(*shape, generator=gen) * 0.6
    g = torch.randn(*shape, generator=gen)
    r = torch.randint(0, 2, shape, generator=gen).float() - 0.5
    co = shape[0]
    if kind == "one":
        s0, zp0 = torch.rand(1, generator=gen) * 0.1 + 0.02, torch.randn(1, generator=gen)
        lo0, hi0 = zp0.clone(), zp0 + float(rng.uniform(0.3, 2.0))
    elif kind == "zero_dim":
        s0, zp0 = (torch.rand((), generator=gen) * 0.1 + 0.02), torch.randn((), generator=gen)
        lo0, hi0 = -math.inf, math.inf
    elif kind == "per_channel":
        bshape = [co] + [1] * (len(shape) - 1)
        s0 = torch.rand(bshape, generator=gen) * 0.1 + 0.02
        zp0 = x.amin(tuple(range(1, len(shape))), keepdim=True)
        lo0, hi0 = -math.inf, math.inf
    else:
        s0, zp0 = torch.rand(shape, generator=gen) * 0.1 + 0.02, -torch.rand(shape, generator=gen)
        lo0, hi0 = -math.inf, math.inf
    dev = lambda t: t.to(DEV) if torch.is_tensor(t) else t          # noqa: E731
    # oracle on the device
    xr, sr, zr = (t.clone().to(DEV).requires_grad_(True) for t in (x, s0, zp0))
    yr = O.dequantize(O.quantize(xr, sr, zr, dev(lo0), dev(hi0), method, r.to(DEV)), sr, zr)
    yr.backward(g.to(DEV))
    xg, sg, zg = (t.clone().to(DEV).requires_grad_(True) for t in (x, s0, zp0))
    Q = M.Quantizer(torch.nn.Identity().train(), sg, zg, dev(lo0), dev(hi0), qnmethod=M.QNMethod[method])
    cls = G._BY_METHOD[M.QNMethod[method]]
    cls.r_sign = (r * 2).to(torch.int8).to(DEV)
    try:
        y = Q.dequantize(Q.quantize(xg))
        y.backward(g.to(DEV))
    finally:
        cls.r_sign = None
    assert bit_equal(y.detach().cpu().numpy(), yr.detach().cpu().numpy())
    yard = float(g.abs().double().sum()) + 1e-30
    tol = 1e-6
    bound = torch.full_like(g, tol * max(1.0, float(g.abs().max())))
    if method == "AEWGS":
        # delta = num / max(e2 - me^2, 1e-3) amplifies the last bits of the three group means (fp64 sums here, fp32
        # in the eager chain) where e2 - me^2 cancels -- with 2..11 samples per group it often does (soak seed 2211:
        # three samples, e2 - me^2 = 1.3e-3, delta = -105).  Bound: the propagated summation slack of the means
        # (tests/aewgs_bound.py; rounds 2-5 added a blanket 2e-5 and a 4e-6 summation slack on top of it)
        dims = {"one": (0,), "zero_dim": None, "per_channel": tuple(range(1, x.dim())), "per_element": None}[kind]
        lo_t = lo0 if torch.is_tensor(lo0) else torch.tensor(lo0)
        hi_t = hi0 if torch.is_tensor(hi0) else torch.tensor(hi0)
        v = (torch.clamp(x, lo_t, hi_t) - zp0) / s0
        amp = aewgs_gx_bound(v, g, tuple(range(x.dim())) if dims is None else dims)    # slack of gv / s per element ...
        bound = bound.double() + amp
        amp_s, amp_zp = float((amp * v.abs().double()).sum()), float(amp.sum())      # ... it enters d/ds (x |v|) and d/dzp
    else:
        amp_s = amp_zp = 0.0
    assert bool(((xg.grad - xr.grad).abs().cpu() <= bound).all()), float((xg.grad - xr.grad).abs().max())
    q_abs = float(((x - zp0).abs() / s0).max()) + 1.0
    assert float((sg.grad - sr.grad).abs().max()) <= tol * yard * q_abs + amp_s
    assert float((zg.grad - zr.grad).abs().max()) <= tol * yard + amp_zp


@pytest.mark.parametrize("seed", range(6 * _K))
def test_fuzz_multi_tensor_mixed_alignment(ops, seed):
    """mhaq_fq_wlayer_fwd_multi / _bwd_multi over random layer sets whose rows are odd, 1x1 or multiples of four,
    so float4 and dword layers (and unaligned slab offsets) share one grid: same results as the per-layer ops."""
    import mhaq_amd as M
    from mhaq_amd.multi import MultiTensorWeightQuant
    rng = np.random.default_rng(15000 + seed)
    torch.manual_seed(15000 + seed)
    method = ["LSQ", "STE", "AEWGS"][seed % 3]
    shapes = []
    for _ in range(int(rng.integers(3, 8))):
        k = int(rng.choice([1, 3, 5]))
        shapes.append((int(rng.integers(1, 20)), int(rng.integers(1, 18)), k, k))
    net = torch.nn.ModuleList([M.NoisyConv2d(s[1], s[0], s[2], bias=False, qscheme=M.QScheme.PER_CHANNEL,
                                             log_s_init=-6, qnmethod=M.QNMethod[method]) for s in shapes]).to(DEV)
    with torch.no_grad():
        for m in net:
            m.log_wght_s.add_(torch.randn_like(m.log_wght_s) * 0.3)
    Gs = [torch.randn(s, device=DEV) for s in shapes]
    hs = [torch.randn(s[0], device=DEV) for s in shapes]
    multi = MultiTensorWeightQuant(net)
    ops.manual_seed(seed + 5)
    wqs = multi.run()
    lwqs = [m._precomputed[3] for m in net]
    (sum((wq * G).sum() for wq, G in zip(wqs, Gs)) + sum((l * h).sum() for l, h in zip(lwqs, hs))).backward()
    got = [(m.weight.grad.clone(), m.log_wght_s.grad.clone()) for m in net]
    r_all = ops.fill_r(multi.total_elems, seed + 5, 1, DEV)
    for i, m in enumerate(net):
        m.weight.grad = None
        m.log_wght_s.grad = None
        n = m.weight.numel()
        r = r_all[multi.elem_off[i]:multi.elem_off[i] + n].view(shapes[i])
        wq, zp, s, lwq = ops.fake_quant_weight_layer(m.weight, m.log_wght_s, method, r_sign=r)
        assert torch.equal(wq, wqs[i]) and torch.equal(lwq, lwqs[i]), (i, shapes[i])
        ((wq * Gs[i]).sum() + (lwq * hs[i]).sum()).backward()
        # the joint launch runs the per-layer row bodies: the same elementwise bits; a row's sums are fp64 partials of fp32
        # terms rounded ONCE, in a partition that differs (256 threads per row here, 64-256 per layer) -- which shows in the
        # fp32 result with probability ~ n 2^-29 sum|t| / |sum|: identical or one ulp apart (the share g_zp / count at a
        # row's extremes inherits that ulp)
        assert _ulp_close(m.weight.grad, got[i][0]), (i, shapes[i])
        assert _ulp_close(m.log_wght_s.grad, got[i][1]), (i, shapes[i])


@pytest.mark.parametrize("seed", range(6 * _K))
def test_fuzz_grouped_weight_backward(ops, seed):
    """The trainer's form: model-wide forward launch + the backward in groups of consecutive layers
    (mhaq_fq_wlayer_bwd_group over windows of the model-wide aux slab, group-relative offsets) on random layer sets
    with odd / 1x1 / float4 rows and random group sizes: same results as the per-layer ops, sign stream replayed."""
    import mhaq_amd as M
    from mhaq_amd.multi import MultiTensorWeightQuant
    rng = np.random.default_rng(17000 + seed)
    torch.manual_seed(17000 + seed)
    method = ["LSQ", "STE", "AEWGS", "EWGS"][seed % 4]
    shapes = []
    for _ in range(int(rng.integers(3, 9))):
        k = int(rng.choice([1, 3, 5]))
        shapes.append((int(rng.integers(1, 20)), int(rng.integers(1, 18)), k, k))
    net = torch.nn.ModuleList([M.NoisyConv2d(s[1], s[0], s[2], bias=False, qscheme=M.QScheme.PER_CHANNEL,
                                             log_s_init=-6, qnmethod=M.QNMethod[method]) for s in shapes]).to(DEV)
    with torch.no_grad():
        for m in net:
            m.log_wght_s.add_(torch.randn_like(m.log_wght_s) * 0.3)
    Gs = [torch.randn(s, device=DEV) for s in shapes]
    hs = [torch.randn(s[0], device=DEV) for s in shapes]
    total = sum(int(np.prod(s)) for s in shapes)
    plan = MultiTensorWeightQuant(net, joint_backward=False,
                                  backward_group_elems=int(rng.integers(1, max(2, total))))
    ops.manual_seed(seed + 5)
    plan.run()
    outs = []
    for m in net:
        wq, _, _ = m._quantized_weight()
        outs.append((wq, m.regulariser_input()))
    (sum((wq * G).sum() for (wq, _), G in zip(outs, Gs)) + sum((l * h).sum() for (_, l), h in zip(outs, hs))).backward()
    got = [(m.weight.grad.clone(), m.log_wght_s.grad.clone()) for m in net]
    units = [list(range(g.first, g.first + g.n)) for g in plan.groups]
    units += [[i] for i in range(len(shapes)) if plan.group_of[i] is None]

    def close(i, offset):
        m, g = net[i], plan.group_of[i]
        m.weight.grad = m.log_wght_s.grad = None
        n = m.weight.numel()
        if g is None:
            r = ops.fill_r(n, seed + 5, offset, DEV)
        else:
            e0 = plan.elem_off[i] - g.elem0
            r = ops.fill_r(g.elems, seed + 5, offset, DEV)[e0:e0 + n]
        wq, zp, s, lwq = ops.fake_quant_weight_layer(m.weight, m.log_wght_s, method, r_sign=r)
        assert torch.equal(wq, outs[i][0]) and torch.equal(lwq, outs[i][1]), (i, shapes[i])
        ((wq * Gs[i]).sum() + (lwq * hs[i]).sum()).backward()
        return _ulp_close(m.weight.grad, got[i][0]) and _ulp_close(m.log_wght_s.grad, got[i][1])    # (see the joint test above)
    for unit in units:       # every backward launch drew one stream, numbered in the order autograd ran them
        hits = [o for o in range(1, len(units) + 1) if all(close(i, o) for i in unit)]
        assert hits, (unit, [shapes[i] for i in unit])
    # (which stream each launch drew is pinned bit for bit in tests/test_gpu_weight_groups.py)
