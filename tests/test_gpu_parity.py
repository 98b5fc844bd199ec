"""GPU parity (-m gpu): the HIP path, called through the C-ABI (ctypes -> libmhaq_fq.so),
against (1) the golden vectors recorded from the real reference, (2) the CPU oracle on the
same seeded inputs, (3) size-independent properties at BASELINE.json's full tensor sizes.

Bar: bit-exact for every elementwise output (y, q, gx, wq; gw where no reduction feeds it);
reduced gradients within 1e-6 * sum|terms| of the reference (tolerance stated per assert).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fq_closed_form as CF  # noqa: E402
from oracle import fq_eager as O  # noqa: E402
from tests.golden_util import (T, bit_equal, exact_off_extremes, load_cases, max_ulp, r_from_sign,  # noqa: E402
                               value_equal)

# + EWGS: the reference's own QNEWGS.backward lines, run with the misspelled attribute of gdnsq.py:102 supplied
# (oracle/gen_golden.py `ewgs_enabled`; the shipped reference raises AttributeError there)
ACT = {**load_cases("act_cases.npz"), **load_cases("ewgs_act_cases.npz")}
WGT = {**load_cases("weight_cases.npz"), **load_cases("ewgs_weight_cases.npz")}
DEV = "cuda:0"
METHODS = ["STE", "EWGS", "AEWGS", "LSQ"]


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import mhaq_amd._lib as L
    import mhaq_amd.ops as ops
    L.lib()  # raises if libmhaq_fq.so is missing: there is no fallback
    return ops


def sign8(c, key="r"):
    return torch.from_numpy(c[key].astype(np.int8)).to(DEV)


def leaf(t):
    return t.detach().clone().to(DEV).requires_grad_(True)


def assert_reduced(got, ref, yard, what, rel=1e-6, slack=0.0):
    """|got - ref| <= rel * sum|terms| (+ slack: the propagated AEWGS term of tests/aewgs_bound.py, derived at the call)"""
    got, ref, yard = (np.asarray(a, dtype=np.float64) for a in (got, ref, yard))
    err = np.abs(got - ref)
    assert np.all(err <= rel * yard + slack + 1e-30), f"{what}: err {err.max():.3e} > {rel:g} * sum|terms| {yard.max():.3e} + {np.max(slack):.3e}"


# ------------------------------------------------------------------------------ K1 golden
@pytest.mark.parametrize("name", sorted(ACT))
def test_act_matches_reference_golden(ops, name):
    c = ACT[name]
    method = O.METHODS[int(c["method"])]            # STE, LSQ and AEWGS NoisyAct cases (gdnsq_act.py:17)
    # CPU leaf parameters; s/qr/hi are formed on the CPU exactly like gdnsq_act.py:42-47 so the
    # kernel sees the same fp32 scale bits as the reference did (device exp2 may differ by 1 ulp)
    ls = T(c["log_act_s"]).reshape(1).requires_grad_(True)
    lq = T(c["log_act_q"]).reshape(1).requires_grad_(True)
    b = T(c["act_b"]).reshape(1).requires_grad_(True)
    s, qr = torch.exp2(ls), torch.exp2(lq)
    hi = b + qr - s
    s_g, zp_g, lo_g, hi_g = leaf(s), leaf(b), leaf(b), leaf(hi)
    x_g = leaf(T(c["x"]))
    y = ops.fake_quant_per_tensor(x_g, s_g, zp_g, lo_g, hi_g, method, r_sign=sign8(c))
    y.backward(T(c["g"]).to(DEV))
    assert bit_equal(y.detach().cpu().numpy(), c["y"])
    x, g = T(c["x"]), T(c["g"])
    delta = None
    if method == "AEWGS":
        # the estimator's input gradient depends on group means (over dim 0 for the [1]-shaped scale): fp64 sums
        # here, fp32 in the reference -- bounded by the propagated summation slack of the three means
        sd, bd, hd = s.detach(), b.detach(), hi.detach()
        v = (torch.clamp(x, bd, hd) - bd) / sd
        e = torch.round(v) - v
        co = x.shape[0]
        mean64 = lambda t: (t.double().sum(0, keepdim=True).float() / float(co))  # noqa: E731
        num, e2, me = mean64((g * sd).sign() * e), mean64(e * e), mean64(e)
        den = (e2 - me * me).clamp_min(1e-3)
        delta = num / den
        ddelta = 1e-6 * (mean64(e.abs()) / den + num.abs() * (e2 + 2 * me.abs() * mean64(e.abs())) / den ** 2)
        tol = (g.abs() * (e.abs() * ddelta + 1e-6)).numpy()
        assert np.all(np.abs(x_g.grad.cpu().numpy() - c["gx"]) <= tol + 1e-30), "AEWGS gx vs the reference"
        # ... and what the reduced gradients inherit from it (tests/aewgs_bound.py::aewgs_slack): d/ds sums (gv / s) * v over
        # every element, d/dhi (d/dlo) sums gv / s over the elements clipped above (below), d/dzp sums it over all of them
        tol64 = tol.astype(np.float64)
        sl_v, sl_all = float((tol64 * np.abs(v.numpy())).sum()), float(tol64.sum())
        sl_over = float((tol64 * (x > hd).numpy()).sum())
    else:
        assert value_equal(x_g.grad.cpu().numpy(), c["gx"])
    # chain the 4 kernel gradients through the scalar graph on the CPU (autograd, as in the layer)
    torch.autograd.backward([s, b, hi], [s_g.grad.cpu(), zp_g.grad.cpu() + lo_g.grad.cpu(), hi_g.grad.cpu()])
    cf = CF.per_tensor(x, g, r_from_sign(c["r"]), s.detach(), b.detach(), b.detach(), hi.detach(), method, delta)
    ln2s = math.log(2.0) * float(s.detach())
    ln2q = math.log(2.0) * float(qr.detach())
    # 1e-6 * sum|terms| for every estimator; AEWGS adds the propagated slack of its group means, term by term:
    # log_act_s: (d/ds - d/dhi) * s ln2 [hi = b + qr - s]; log_act_q: d/dhi * qr ln2; act_b: d/dzp + d/dlo + d/dhi <= 2 sum
    sl = (sl_v, sl_all, sl_over) if method == "AEWGS" else (0.0, 0.0, 0.0)
    assert_reduced(ls.grad, c["g_log_act_s"], (float(cf["abs_s"]) + float(cf["abs_g"])) * ln2s, "g_log_act_s",
                   slack=(sl[0] + sl[2]) * ln2s)
    assert_reduced(lq.grad, c["g_log_act_q"], float(cf["abs_g"]) * ln2q, "g_log_act_q", slack=sl[2] * ln2q)
    if c["signed"]:
        assert_reduced(b.grad, c["g_act_b"], float(cf["abs_g"]), "g_act_b", slack=2 * sl[1])
    # the layer entry points (mhaq_fq_act_fwd / _bwd: exp2 and the clamp bounds derived in the kernel) on the same
    # case, whenever the device's exp2 gives the reference's scale bits (always for integer log parameters)
    if method != "AEWGS":
        ls_d, lq_d, b_d = (leaf(t.detach()) for t in (ls, lq, b))
        x_d = leaf(x)
        y2, params = ops.fake_quant_act_layer(x_d, ls_d, lq_d, b_d, method, r_sign=sign8(c))
        y2.backward(g.to(DEV))
        same_scale = torch.equal(params.cpu()[[0, 3, 4]], torch.cat([s.detach(), hi.detach(), qr.detach()]))
        if float(c["log_act_s"]) == round(float(c["log_act_s"])) and float(c["log_act_q"]) == round(float(c["log_act_q"])):
            assert same_scale
        if same_scale:
            assert bit_equal(y2.detach().cpu().numpy(), c["y"])
            assert value_equal(x_d.grad.cpu().numpy(), c["gx"])
            assert_reduced(ls_d.grad.cpu(), c["g_log_act_s"], (float(cf["abs_s"]) + float(cf["abs_g"])) * ln2s,
                           "act_bwd g_log_act_s")
            assert_reduced(lq_d.grad.cpu(), c["g_log_act_q"], float(cf["abs_g"]) * ln2q, "act_bwd g_log_act_q")
            if c["signed"]:
                assert_reduced(b_d.grad.cpu(), c["g_act_b"], float(cf["abs_g"]), "act_bwd g_act_b")
    # eval mode: bit width and integrity flags
    ye, q, qstats, flags = ops.fake_quant_per_tensor_eval(T(c["x"]).to(DEV), s_g.detach(), zp_g.detach(),
                                                         lo_g.detach(), hi_g.detach(), want_q=True)
    assert bit_equal(ye.cpu().numpy(), c["y"])
    qn = q.cpu().numpy()
    assert np.array_equal(qn, np.rint(qn)), "rounding indices must be integer valued"
    if c["eval_raises"]:
        assert int(flags.item()) != 0
    else:
        assert int(flags.item()) == 0
        qs = qstats.cpu()                      # log2 on the host: device log2 may differ by 1 ulp
        bw = torch.log2(qs[1] - qs[0] + 1)
        assert bit_equal(bw.numpy().reshape(()), c["bw"].reshape(()))


# ------------------------------------------------------------------------------ K2 golden
@pytest.mark.parametrize("name", sorted(WGT))
def test_weight_matches_reference_golden(ops, name):
    c = WGT[name]
    pc = bool(c["per_channel"])
    method = O.METHODS[int(c["method"])]
    ls = T(c["log_wght_s"]).requires_grad_(True)
    s = torch.exp2(ls)
    s_g, w_g = leaf(s), leaf(T(c["w"]))
    has_bias = "bias" in c
    if pc:
        wq, zp = ops.fake_quant_weight_pc(w_g, s_g, method, r_sign=sign8(c), zp_grad=has_bias)
    else:
        wq, zp = ops.fake_quant_weight_pt(w_g, s_g, method, r_sign=sign8(c))
    outs, grads = [wq], [T(c["G"]).to(DEV)]
    if has_bias:
        b_g = leaf(T(c["bias"]))
        bq = ops.fake_quant_per_element(b_g, s_g.ravel(), zp.ravel(), method, r_sign=sign8(c, "rb"))
        outs.append(bq)
        grads.append(T(c["Gb"]).to(DEV))
    torch.autograd.backward(outs, grads)
    assert bit_equal(wq.detach().cpu().numpy(), c["wq"])
    assert bit_equal(zp.detach().cpu().numpy().reshape(c["zp"].shape), c["zp"])
    # yardsticks from the closed form (per channel, or whole tensor for per-tensor)
    w, G, r = T(c["w"]), T(c["G"]), r_from_sign(c["r"])
    if pc:
        cf = CF.per_channel(w, G, r, s.detach().reshape(-1), method)
        abs_g = cf["abs_g"].reshape([-1] + [1] * (w.dim() - 1)).numpy()
        abs_s = cf["abs_s"].numpy()
    else:
        cf = CF.per_channel(w.reshape(1, -1), G.reshape(1, -1), r.reshape(1, -1), s.detach().reshape(1),
                            "STE" if method == "AEWGS" else method)
        abs_g = float(cf["abs_g"])
        abs_s = float(cf["abs_s"])
    if has_bias:
        abs_g = abs_g + np.abs(c["Gb"]).reshape(abs_g.shape) * 2
        assert bit_equal(bq.detach().cpu().numpy(), c["bq"])
        assert np.allclose(b_g.grad.cpu().numpy(), c["gbias"], rtol=1e-6, atol=1e-7)
    # gw: for STE / LSQ / EWGS the elementwise part (G*s [+ estimator])/s is exact -- every element that is not a minimum of its
    # group equals the reference's value; the minima carry the tie-split share of the (reduced) zero-point gradient
    # (EWGS adds -|G*s| * e * 0.01 per element, gdnsq.py:96-100: elementwise too)
    if method in ("STE", "LSQ", "EWGS"):
        assert exact_off_extremes(w_g.grad.cpu().numpy(), c["gw"], c["w"], pc), "gw off the minima"
    assert_reduced(w_g.grad.cpu().numpy(), c["gw"], abs_g + np.abs(c["gw"]), "gw")
    ls_grad = torch.autograd.grad(s, ls, s_g.grad.cpu().reshape(s.shape))[0]
    yard = (abs_s * math.log(2.0) * s.detach().reshape(-1).numpy()).reshape(c["g_log_wght_s"].shape)
    if has_bias:
        yard = yard + (np.abs(c["Gb"]) * np.abs(c["bq"]) * 4).reshape(yard.shape)
    assert_reduced(ls_grad.numpy(), c["g_log_wght_s"], yard * 2, "g_log_wght_s")


# ------------------------------------------------------------------------------ vs oracle, seeded
def _oracle_per_tensor(x, g, r, s, zp, lo, hi, method):
    xs = x.clone().requires_grad_(True)
    P = [t.clone().reshape(1).requires_grad_(True) for t in (s, zp, lo, hi)]
    y = O.dequantize(O.quantize(xs, P[0], P[1], P[2], P[3], method, r), P[0], P[1])
    y.backward(g)
    return y.detach(), xs.grad, [p.grad for p in P]


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("shape,scale_kind", [((8, 64, 56, 56), "w4"), ((4, 50, 24, 24), "calib10"),
                                              ((3, 5, 7, 9), "w4"), ((1, 1, 1, 3), "w4"),
                                              ((128, 16, 32, 32), "pow2"), ((2, 4099), "w4")])
def test_per_tensor_matches_oracle(ops, method, shape, scale_kind):
    gen = torch.Generator().manual_seed(len(shape) * 1000 + shape[-1])
    x = torch.randn(*shape, generator=gen) * 2
    g = torch.randn(*shape, generator=gen)
    r = torch.randint(0, 2, shape, generator=gen).float() - 0.5
    rng_ = float(x.max() - x.min())
    if scale_kind == "w4":
        s = torch.tensor(rng_ * 0.7 / 15); zp = x.min() * 0.7; hi = zp + 16 * s - s
    elif scale_kind == "calib10":
        s = torch.tensor(rng_ / 1023); zp = x.min().clone(); hi = zp + 1024 * s - s
    else:
        s = torch.tensor(2.0 ** -3); zp = torch.tensor(-2.0); hi = zp + 32 * s - s
    lo = zp.clone()
    y_ref, gx_ref, pg_ref = _oracle_per_tensor(x, g, r, s, zp, lo, hi, method)
    P = [leaf(t.reshape(1)) for t in (s, zp, lo, hi)]
    x_g = leaf(x)
    y = ops.fake_quant_per_tensor(x_g, *P, method, r_sign=(r * 2).to(torch.int8).to(DEV))
    y.backward(g.to(DEV))
    assert bit_equal(y.detach().cpu().numpy(), y_ref.numpy())
    if method == "AEWGS":
        # Group statistics are means over dim 0 of fp32 terms (gdnsq.py:150-152).  The kernel sums
        # them in fp64 and rounds once; torch sums in fp32 in an order that differs between its CPU
        # and GPU back ends.  delta = num / max(e2 - me^2, 1e-3) amplifies that last-bit difference
        # where e2 - me^2 cancels, so: tight check against the closed form evaluated with
        # fp64-summed means (the kernel's specification), loose check against the eager oracle.
        v = (torch.clamp(x, lo, hi) - zp) / s
        e = torch.round(v) - v
        co = shape[0]
        mean64 = lambda t: (t.double().sum(0, keepdim=True).float() / float(co))  # noqa: E731
        num, e2, me = mean64((g * s).sign() * e), mean64(e * e), mean64(e)
        delta = num / (e2 - me * me).clamp_min(1e-3)
        cf = CF.per_tensor(x, g, r, s, zp, lo, hi, "AEWGS", delta)
        assert np.allclose(x_g.grad.cpu().numpy(), cf["gx"].numpy(), rtol=1e-6, atol=1e-7 * float(g.abs().max()))
        assert_reduced(float(P[0].grad), float(cf["g_s"]), float(cf["abs_s"]), "AEWGS g_s vs closed form")
        # against the reference-order eager chain: its fp32 group means differ from the fp64 ones in the last
        # bits; propagated through delta that is at most |g| * |e| * ddelta per element
        den = (e2 - me * me).clamp_min(1e-3)
        ddelta = 1e-6 * (mean64(e.abs()) / den + num.abs() * (e2 + 2 * me.abs() * mean64(e.abs())) / den ** 2)
        tol = (g.abs() * (e.abs() * ddelta + 1e-6)).numpy()
        assert np.all(np.abs(x_g.grad.cpu().numpy() - gx_ref.numpy()) <= tol + 1e-30)
        return
    assert value_equal(x_g.grad.cpu().numpy(), gx_ref.numpy())
    cf = CF.per_tensor(x, g, r, s, zp, lo, hi, method)
    for i, (key, yard) in enumerate((("g_s", "abs_s"), ("g_zp", "abs_g"), ("g_lo", "abs_g"), ("g_hi", "abs_g"))):
        got = float(P[i].grad)
        assert_reduced(got, float(pg_ref[i]), float(cf[yard]), f"{key} vs reference-order eager")
        assert_reduced(got, float(cf[key]), float(cf[yard]), f"{key} vs fp64 closed form", rel=1e-7)


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("shape", [(64, 64, 3, 3), (512, 512, 3, 3), (50, 50, 3, 3), (16, 3, 7, 7), (10, 64), (5, 1, 1, 1),
                                   (20000, 7)])
@pytest.mark.parametrize("per_channel", [True, False])
def test_weight_matches_oracle(ops, method, shape, per_channel):
    gen = torch.Generator().manual_seed(shape[0] * 7 + len(shape))
    fan_in = int(np.prod(shape[1:]))
    w = torch.randn(*shape, generator=gen) * math.sqrt(2.0 / fan_in)
    G = torch.randn(*shape, generator=gen)
    r = torch.randint(0, 2, shape, generator=gen).float() - 0.5
    dims = tuple(range(1, len(shape)))
    if per_channel:
        span = (w.amax(dims) - w.amin(dims)).clamp_min(1e-3)
        ls = torch.log2(span / 15.0).reshape([shape[0]] + [1] * (len(shape) - 1))
    else:
        ls = torch.log2((w.max() - w.min()) / 15.0).reshape(1)
    s = torch.exp2(ls)
    ws, ss = w.clone().requires_grad_(True), s.clone().requires_grad_(True)
    zp_ref = O.weight_zero_point(ws, per_channel)
    wq_ref = O.dequantize(O.quantize(ws, ss, zp_ref, -math.inf, math.inf, method, r), ss, zp_ref)
    wq_ref.backward(G)
    w_g, s_g = leaf(w), leaf(s)
    fn = ops.fake_quant_weight_pc if per_channel else ops.fake_quant_weight_pt
    wq, zp = fn(w_g, s_g, method, r_sign=(r * 2).to(torch.int8).to(DEV))
    wq.backward(G.to(DEV))
    assert bit_equal(wq.detach().cpu().numpy(), wq_ref.detach().numpy())
    assert bit_equal(zp.detach().cpu().numpy().reshape(-1), zp_ref.detach().numpy().reshape(-1))
    if per_channel:
        cf = CF.per_channel(w, G, r, s.reshape(-1), method)
        abs_g = cf["abs_g"].reshape([-1] + [1] * (len(shape) - 1)).numpy()
        abs_s = cf["abs_s"].numpy().reshape(s.shape)
        # the fp64 closed form is this kernel's exact specification
        assert_reduced(s_g.grad.cpu().numpy(), cf["g_s"].numpy().reshape(s.shape), abs_s, "g_s vs closed form",
                       rel=1e-6 if method == "AEWGS" else 1e-7)
    else:
        cf = CF.per_channel(w.reshape(1, -1), G.reshape(1, -1), r.reshape(1, -1), s.reshape(1),
                            "STE" if method == "AEWGS" else method)
        abs_g, abs_s = float(cf["abs_g"]), float(cf["abs_s"])
    if method != "AEWGS":        # (G*s [+ EWGS term])/s is elementwise: exact wherever no reduced share is added
        assert exact_off_extremes(w_g.grad.cpu().numpy(), ws.grad.numpy(), w.numpy(), per_channel), "gw off the minima"
    assert_reduced(w_g.grad.cpu().numpy(), ws.grad.numpy(), abs_g + ws.grad.abs().numpy(), "gw")
    assert_reduced(s_g.grad.cpu().numpy(), ss.grad.numpy(), abs_s * 2, "g_s vs reference-order eager")


# ------------------------------------------------------------------------------ in-kernel Philox
def test_inkernel_philox_stream(ops):
    n = 3 * 4096 + 517
    seed, offset = 0x1234ABCD5678, 42
    r8 = ops.fill_r(n, seed, offset, DEV)
    rf = r8.float().cpu()
    assert set(np.unique(rf.numpy())) == {-1.0, 1.0}
    big = ops.fill_r(1 << 22, seed, offset, DEV).float()
    assert abs(float(big.mean())) < 4.0 / math.sqrt(1 << 22)          # fair
    other = ops.fill_r(1 << 22, seed, offset + 1, DEV).float()
    assert abs(float((big * other).mean())) < 4.0 / math.sqrt(1 << 22)  # streams independent
    assert abs(float((big[1:] * big[:-1]).mean())) < 4.0 / math.sqrt(1 << 22)  # no lag-1 correlation
    # a backward with in-kernel signs == the same backward with the materialised stream
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(n, generator=gen) * 2
    g = torch.randn(n, generator=gen)
    P = [torch.tensor([v], device=DEV) for v in (0.21, -1.5, -1.5, 1.4)]
    ops.manual_seed(seed)
    xg = leaf(x)
    Pg = [p.clone().requires_grad_(True) for p in P]
    y = ops.fake_quant_per_tensor(xg, *Pg, "STE")
    y.backward(g.to(DEV))                       # first call after manual_seed -> offset 1
    r_used = ops.fill_r(n, seed, 1, DEV)
    xg2 = leaf(x)
    Pg2 = [p.clone().requires_grad_(True) for p in P]
    y2 = ops.fake_quant_per_tensor(xg2, *Pg2, "STE", r_sign=r_used)
    y2.backward(g.to(DEV))
    assert bit_equal(xg.grad.cpu().numpy(), xg2.grad.cpu().numpy())
    for a, b2 in zip(Pg, Pg2):
        assert bit_equal(a.grad.cpu().numpy(), b2.grad.cpu().numpy())
    # and the per-channel kernel draws the same stream definition
    w = torch.randn(8, 4, 3, 3, generator=gen)
    G = torch.randn(8, 4, 3, 3, generator=gen)
    s = torch.full((8, 1, 1, 1), 0.05)
    ops.manual_seed(seed)
    wg, sg = leaf(w), leaf(s)
    ops.fake_quant_weight_pc(wg, sg, "STE")[0].backward(G.to(DEV))
    wg2, sg2 = leaf(w), leaf(s)
    ops.fake_quant_weight_pc(wg2, sg2, "STE", r_sign=ops.fill_r(w.numel(), seed, 1, DEV).view(w.shape))[0].backward(G.to(DEV))
    assert bit_equal(sg.grad.cpu().numpy(), sg2.grad.cpu().numpy())


# ------------------------------------------------------------------------------ views / alignment
def test_unaligned_and_noncontiguous_inputs(ops):
    gen = torch.Generator().manual_seed(9)
    base = torch.randn(4 * 1031 + 3, generator=gen)
    P = [torch.tensor([v]) for v in (0.11, -1.0, -1.0, 0.9)]
    x = base[1:]                       # 4-byte aligned only
    r = torch.randint(0, 2, x.shape, generator=gen).float() - 0.5
    g = torch.randn(x.shape, generator=gen)
    y_ref, gx_ref, _ = _oracle_per_tensor(x, g, r, *P, "STE")
    xg = base.to(DEV)[1:].detach().requires_grad_(True)
    Pg = [leaf(p) for p in P]
    y = ops.fake_quant_per_tensor(xg, *Pg, "STE", r_sign=(r * 2).to(torch.int8).to(DEV))
    y.backward(g.to(DEV))
    assert bit_equal(y.detach().cpu().numpy(), y_ref.numpy())
    assert value_equal(xg.grad.cpu().numpy(), gx_ref.numpy())
    xt = torch.randn(6, 5, generator=gen)
    yt = ops.fake_quant_per_tensor(xt.to(DEV).t(), *[p.to(DEV) for p in P], "LSQ")
    yr, _, _ = _oracle_per_tensor(xt.t().contiguous(), torch.zeros(5, 6), None, *P, "LSQ")
    assert bit_equal(yt.cpu().numpy(), yr.numpy())


def test_empty_tensor(ops):
    P = [torch.tensor([v], device=DEV, requires_grad=True) for v in (0.1, 0.0, 0.0, 1.0)]
    x = torch.empty(0, 3, device=DEV, requires_grad=True)
    y = ops.fake_quant_per_tensor(x, *P, "STE")
    assert y.shape == (0, 3)
    y.sum().backward()
    assert all(float(p.grad) == 0.0 for p in P)


def test_cpu_tensor_fails_loudly(ops):
    import mhaq_amd._lib as L
    with pytest.raises(L.MhaqFqError):
        ops.fake_quant_per_tensor(torch.ones(4), 0.1, 0.0, 0.0, 1.0)


# ------------------------------------------------------------------------------ full BASELINE sizes
@pytest.mark.parametrize("shape", [(250, 64, 56, 56), (1000, 16, 32, 32), (24, 50, 24, 24), (24, 50, 180, 320)])
def test_full_size_activation_properties(ops, shape):
    """ResNet-18 layer1 input at batch 250 (50.2 M elements, 200 MB), ResNet-20 stage-1 at batch 1000, and
    config 5 (RFDN, batch 24) at the reference's training shape [24,50,24,24] and at the full-frame stress shape
    [24,50,180,320] (69.1 M elements, 276 MB per tensor: SURVEY.md 8d): size-independent properties + a direct
    oracle comparison on a 2 M-element window."""
    torch.manual_seed(0)
    x = torch.randn(*shape, device=DEV) * 2
    g = torch.randn(*shape, device=DEV)
    s = torch.tensor([2.0 ** -2], device=DEV, requires_grad=True)   # power of two: (g*s)/s == g exactly
    b = torch.tensor([-2.0], device=DEV, requires_grad=True)
    hi = (b + 16 * s - s).detach().requires_grad_(True)
    lo = b.detach().clone().requires_grad_(True)
    xg = x.clone().requires_grad_(True)
    from mhaq_amd import ops as _ops
    _ops.manual_seed(123)
    y = ops.fake_quant_per_tensor(xg, s, b, lo, hi, "STE")
    y.backward(g)
    # idempotence: re-quantizing the output is the identity
    y2 = ops.fake_quant_per_tensor(y.detach(), s.detach(), b.detach(), lo.detach(), hi.detach(), "STE")
    assert torch.equal(y2, y.detach())
    # rounding indices are integers in [0, 15]
    q = (y.detach() - b.detach()) / s.detach()
    assert torch.equal(q, q.round()) and float(q.min()) >= 0 and float(q.max()) <= 15
    # straight-through with inclusive bounds
    inside = (x >= lo.detach()) & (x <= hi.detach())
    assert torch.equal(xg.grad, torch.where(inside, g, torch.zeros_like(g)))
    # gradient mass conservation: sum(gx) + g_lo + g_hi == sum(g)   (fp64 yardstick)
    tot = g.double().sum()
    got = xg.grad.double().sum() + lo.grad.double() + hi.grad.double()
    assert abs(float(got - tot)) <= 1e-6 * float(g.double().abs().sum())
    assert abs(float(b.grad)) <= 1e-6 * float(g.double().abs().sum())          # sum g - sum g1 == 0 here
    assert abs(float(lo.grad) - float(g[x < lo.detach()].double().sum())) <= 1e-6 * float(g.double().abs().sum())
    # determinism: same seed/offset -> bit-identical reductions
    _ops.manual_seed(123)
    xg2 = x.clone().requires_grad_(True)
    P2 = [t.detach().clone().requires_grad_(True) for t in (s, b, lo, hi)]
    ops.fake_quant_per_tensor(xg2, *P2, "STE").backward(g)
    assert all(torch.equal(a.grad, b2.grad) for a, b2 in zip((s, b, lo, hi), P2))
    # direct oracle comparison on a window that starts mid-tensor
    n0, n1 = 12_345_678 % x.numel(), min(x.numel(), 12_345_678 % x.numel() + 2_000_000)
    xw, gw = x.flatten()[n0:n1].cpu(), g.flatten()[n0:n1].cpu()
    cf = CF.per_tensor(xw, gw, torch.zeros_like(xw), s.detach().cpu(), b.detach().cpu(), lo.detach().cpu(),
                       hi.detach().cpu(), "LSQ")
    assert bit_equal(y.detach().flatten()[n0:n1].cpu().numpy(), cf["y"].numpy())
    assert value_equal(xg.grad.flatten()[n0:n1].cpu().numpy(), cf["gx"].numpy())


def test_full_size_scale_gradient_vs_fp64(ops):
    """g_s over 50 M elements against an fp64 evaluation of the same fp32 terms on the GPU."""
    torch.manual_seed(1)
    shape = (250, 64, 56, 56)
    x = torch.randn(*shape, device=DEV) * 2
    g = torch.randn(*shape, device=DEV)
    s = torch.tensor([0.2371], device=DEV, requires_grad=True)
    b = torch.tensor([-1.9], device=DEV, requires_grad=True)
    hi = (b + 16 * s - s).detach().requires_grad_(True)
    lo = b.detach().clone().requires_grad_(True)
    y = ops.fake_quant_per_tensor(x, s, b, lo, hi, "LSQ")
    y.backward(g)
    sd, bd, lod, hid = (t.detach() for t in (s, b, lo, hi))
    v = (torch.clamp(x, lod, hid) - bd) / sd
    q = v + (torch.round(v) - v)
    gq = g * sd
    terms = (g * q + (-gq) * (v / sd)) + gq * (torch.round(v) - v)
    ref = float(terms.double().sum())
    yard = float((g * q).double().abs().sum() * 2)
    assert abs(float(s.grad) - ref) <= 1e-7 * yard
    assert bit_equal(y.detach().flatten()[:1000].cpu().numpy(), (q * sd + bd).flatten()[:1000].cpu().numpy())


def test_full_size_resnet18_weights(ops):
    """All 16 ResNet-18 per-channel weight tensors (10.99 M elements): forward exact vs torch on the
    GPU, zero point == row minimum, gw tie split conserves the zero-point gradient."""
    torch.manual_seed(2)
    shapes = [(64, 64, 3, 3)] * 4 + [(128, 64, 3, 3)] + [(128, 128, 3, 3)] * 3 + [(256, 128, 3, 3)] + \
             [(256, 256, 3, 3)] * 3 + [(512, 256, 3, 3)] + [(512, 512, 3, 3)] * 3
    total = 0
    for shp in shapes:
        w = (torch.randn(*shp, device=DEV) * math.sqrt(2.0 / (shp[1] * 9))).requires_grad_(True)
        span = (w.detach().amax((1, 2, 3)) - w.detach().amin((1, 2, 3)))
        s = (span / 15).reshape(-1, 1, 1, 1).requires_grad_(True)
        G = torch.randn(*shp, device=DEV)
        wq, zp = ops.fake_quant_weight_pc(w, s, "LSQ")
        wq.backward(G)
        wd, sd = w.detach(), s.detach()
        zr = wd.amin((1, 2, 3), keepdim=True)
        assert torch.equal(zp, zr)
        v = (wd - zr) / sd
        q = v + (torch.round(v) - v)
        assert torch.equal(wq.detach(), q * sd + zr)
        gvs = (G * sd) / sd
        gzp = (G.double() - gvs.double()).sum((1, 2, 3))
        extra = (w.grad.double() - gvs.double()).sum((1, 2, 3))       # what the tie split added
        assert torch.allclose(extra, gzp, rtol=0, atol=1e-6 * float(G.abs().sum((1, 2, 3)).max()))
        total += w.numel()
    assert total == 10_985_472


def test_full_size_rfdn_weights_lsq(ops):
    """Config 5 (config/gdnsq_config_rfdn_lsq_w2a2.yaml): all 33 wrapped RFDN convolutions, per-channel LSQ, W2
    grid, through the layer entry point (mhaq_fq_wlayer_fwd / _bwd incl. the regulariser input) against the
    eager oracle on the CPU with the same upstream gradients: wq / zp / lwq bit-exact, gW exact off the row
    extremes, d/dlog_wght_s within 1e-6 * sum|terms|."""
    per = [(50, 50, 3, 3)] * 3 + [(25, 50, 3, 3)] + [(12, 12, 3, 3)] * 4
    shapes = per * 4 + [(50, 50, 3, 3)]
    assert len(shapes) == 33 and sum(int(np.prod(s)) for s in shapes) == 358_236
    gen = torch.Generator().manual_seed(55)
    for k, shp in enumerate(shapes):
        w = torch.randn(*shp, generator=gen) * math.sqrt(2.0 / (shp[1] * 9))
        G = torch.randn(*shp, generator=gen)
        h = torch.randn(shp[0], generator=gen) * 0.01
        span = w.amax((1, 2, 3)) - w.amin((1, 2, 3))
        ls0 = torch.log2(span / 3.0).reshape(-1, 1, 1, 1) + 0.2 * torch.randn(shp[0], 1, 1, 1, generator=gen)
        # the scale bits the device derives from log_wght_s drive both sides
        wg, lsg = leaf(w), leaf(ls0)
        wq, zp, s, lwq = ops.fake_quant_weight_layer(wg, lsg, "LSQ")
        torch.autograd.backward([wq, lwq], [G.to(DEV), h.to(DEV)])
        sd = s.detach().cpu()
        wr = w.clone().requires_grad_(True)
        sr = sd.clone().requires_grad_(True)
        zr = O.weight_zero_point(wr, True)
        wq_r = O.dequantize(O.quantize(wr, sr, zr, -math.inf, math.inf, "LSQ"), sr, zr)
        lwq_r = torch.log2(wr.amax((1, 2, 3)) - wr.amin((1, 2, 3)) + sr.ravel())
        torch.autograd.backward([wq_r, lwq_r], [G, h])
        assert bit_equal(wq.detach().cpu().numpy(), wq_r.detach().numpy()), k
        assert bit_equal(zp.detach().cpu().numpy().reshape(-1), zr.detach().numpy().reshape(-1)), k
        # log2 on the device vs on the host may differ in the last bit; the forward's own value is checked
        # bit-exactly against the device eager chain in test_gpu_fused_layers.py
        assert np.allclose(lwq.detach().cpu().numpy(), lwq_r.detach().numpy(), rtol=2e-7, atol=1e-6), k
        assert exact_off_extremes(wg.grad.cpu().numpy(), wr.grad.numpy(), w.numpy(), True, also_max=True), k
        cf = CF.per_channel(w, G, None, sd.reshape(-1), "LSQ")
        u = (span + sd.reshape(-1)).numpy()
        t = np.abs(h.numpy()) / (u * math.log(2.0))
        abs_g = (cf["abs_g"].numpy() + 4 * t).reshape(-1, 1, 1, 1)
        assert_reduced(wg.grad.cpu().numpy(), wr.grad.numpy(), abs_g + np.abs(wr.grad.numpy()), f"gw[{k}]")
        # d/dlog_s = d/ds * s * ln2 (exp2 backward)
        exp_ls = sr.grad.numpy().reshape(-1) * sd.numpy().reshape(-1) * math.log(2.0)
        yard = (cf["abs_s"].numpy() + 4 * t) * math.log(2.0) * sd.numpy().reshape(-1) * 2
        assert_reduced(lsg.grad.cpu().numpy().reshape(-1), exp_ls, yard, f"g_log_wght_s[{k}]")


# ------------------------------------------------------------------------------ Markstein quotients
def test_backward_quotients_equal_ieee_division_for_many_scales(ops):
    """The backward replaces two IEEE divisions by the wave-uniform scale with FMA corrections of a
    reciprocal estimate.  For 300 random scales over 12 decades, including scales with
    an all-ones significand (the documented fallback) and near-power-of-two ones, gx must equal torch's
    (g*s)/s and the recomputed indices must equal the forward's, bit for bit, on 1 M wide-range elements."""
    gen = torch.Generator(device=DEV).manual_seed(2024)
    n = 1 << 20
    mag = torch.exp2(torch.randint(-20, 20, (n,), device=DEV, generator=gen).float())
    x = torch.randn(n, device=DEV, generator=gen) * 3
    g = torch.randn(n, device=DEV, generator=gen) * mag
    scales = torch.exp2(torch.rand(296, generator=torch.Generator().manual_seed(1)) * 40 - 30).tolist()
    scales += [float(np.float32(np.nextafter(np.float32(2.0), np.float32(0.0)))) * 0.25,   # all-ones significand
               float(np.nextafter(np.float32(0.125), np.float32(1.0))), 0.1, 3.0]
    bad = 0
    for s0 in scales:
        s = torch.tensor([s0], device=DEV, requires_grad=True)
        zp = torch.tensor([-5.0 * s0], device=DEV)
        lo = zp.clone()
        hi = zp + 60.0 * s0
        xs = (x * s0 * 8).requires_grad_(True)
        y = ops.fake_quant_per_tensor(xs, s, zp, lo, hi, "LSQ")
        y.backward(g)
        sd = s.detach()
        inside = (xs.detach() >= lo) & (xs.detach() <= hi)
        want = torch.where(inside, (g * sd) / sd, torch.zeros_like(g))
        v = (torch.clamp(xs.detach(), lo, hi) - zp) / sd
        q = v + (torch.round(v) - v)
        bad += int((xs.grad != want).sum()) + int((y.detach() != q * sd + zp).sum())
        # LSQ scale gradient recomputes q and the noise in the backward: compare with torch in fp64
        ref = ((g.double() * (q - v).double()) + ((g * sd) * (q - v)).double()).sum()
        yard = (g.abs().double() * (q - v).abs().double()).sum() * 2 + 1e-30
        # 1 M terms over 12 decades: per-lane fp32 partials of <= 17 terms, fp32 DPP wave sums (1088 terms), fp64 above;
        # each rounding costs <= 2^-24 of the running partial, i.e. of the few dominant terms -> ~1e-7 * sum|terms|
        assert abs(float(s.grad) - float(ref)) <= 1e-6 * float(yard), s0
    assert bad == 0


def test_more_than_2_31_elements(ops):
    """Maximum-size edge: a tensor past the int32 element range with a ragged tail (n % 4 == 3), through
    both fused kernels.  Checked window by window on the device against the eager formula (bit-exact y and
    gx, windows straddling element 2^31 and the tail included) and the reduced gradients against fp64 sums."""
    n = (1 << 31) + 4099
    if torch.cuda.mem_get_info()[0] < 48e9:
        pytest.skip("needs ~40 GB of free HBM")
    gen = torch.Generator(device=DEV).manual_seed(5)
    x = torch.empty(n, device=DEV).normal_(generator=gen).mul_(2)
    g = torch.empty(n, device=DEV).normal_(generator=gen)
    s = torch.tensor([0.25], device=DEV, requires_grad=True)
    b = torch.tensor([-2.0], device=DEV, requires_grad=True)
    lo = b.detach().clone().requires_grad_(True)
    hi = (b + 16 * s - s).detach().requires_grad_(True)
    xg = x.requires_grad_(True)
    y = ops.fake_quant_per_tensor(xg, s, b, lo, hi, "LSQ")
    y.backward(g)
    gx, y = xg.grad, y.detach()
    sd, bd, lod, hid = (t.detach() for t in (s, b, lo, hi))
    x = x.detach()

    def eager(sl):
        v = (torch.clamp(x[sl], lod, hid) - bd) / sd
        e = torch.round(v) - v
        return (v + e) * sd + bd, torch.where((x[sl] >= lod) & (x[sl] <= hid), g[sl], torch.zeros_like(g[sl])), e

    for a in (0, (1 << 31) - 1_000_003, n - 2_000_001):          # head, across 2^31, ragged tail
        sl = slice(a, min(n, a + 2_000_001))
        ye, gxe, _ = eager(sl)
        assert torch.equal(y[sl], ye) and torch.equal(gx[sl], gxe)
    # reduced gradients: fp64 sums of the fp32 terms, window by window
    ref_s = ref_lo = ref_hi = yard = 0.0
    for a in range(0, n, 1 << 28):
        sl = slice(a, min(n, a + (1 << 28)))
        _, _, e = eager(sl)
        ref_s += float((g[sl] * e + (g[sl] * sd) * e).double().sum())      # g*n + noise term (LSQ: gq*n)
        ref_lo += float(g[sl][x[sl] < lod].double().sum())
        ref_hi += float(g[sl][x[sl] > hid].double().sum())
        yard += float(g[sl].double().abs().sum())
    assert abs(float(s.grad) - ref_s) <= 1e-6 * yard
    assert abs(float(lo.grad) - ref_lo) <= 1e-6 * yard
    assert abs(float(hi.grad) - ref_hi) <= 1e-6 * yard


@pytest.mark.parametrize("n", [1, 5, 1024, 4099, 3 * 4096 + 7, 1 << 20])
def test_eval_flag_word_equals_the_three_reference_asserts(ops, n):
    """The eval-mode forward reports gdnsq.py:211-217 as a flag word.  The two range asserts come from the running
    min / max of q and integrality from q == rne(q): the word must equal the reference's three torch.any / torch.all
    expressions for finite data, for NaN and infinite inputs, for a quantizer whose bounds do not bracket the data
    (lo > hi) and for a bound range that excludes part of q -- at sizes that cover the tail, one block and many blocks."""
    gen = torch.Generator().manual_seed(n)
    base = (torch.randn(n, generator=gen) * 3).to(DEV)
    cases = []
    cases.append((base.clone(), 0.25, -1.0, -2.0, 2.0))                       # ordinary clamp
    x = base.clone(); x[n // 2] = float("nan")
    cases.append((x, 0.25, -1.0, -2.0, 2.0))                                  # NaN -> not integer
    x = base.clone(); x[0] = float("inf"); x[-1] = float("-inf")
    cases.append((x, 0.25, -1.0, -2.0, 2.0))                                  # infinities are clamped away
    cases.append((base.clone(), 0.3, 0.5, 1.0, -1.0))                         # lo > hi: every q from hi
    cases.append((base.clone(), 0.25, 0.0, -float("inf"), float("inf")))      # no clamp (weights)
    cases.append((base.clone() * 1e30, 1e-3, 0.0, -float("inf"), float("inf")))   # q overflows to inf: inf - inf = NaN
    for x, s, zp, lo, hi in cases:
        y, q, qstats, flags = ops.fake_quant_per_tensor_eval(x, s, zp, lo, hi, want_q=True)
        st, zt, lt, ht = (torch.tensor(v, device=DEV) for v in (s, zp, lo, hi))
        v = (torch.clamp(x, min=lt, max=ht) - zt) / st
        qr = v + (torch.round(v) - v)
        assert torch.equal(q.view(torch.int32), qr.view(torch.int32))
        want = 0
        if bool(torch.any(qr < torch.floor((lt - zt) / st))):
            want |= 1
        if bool(torch.any(qr > torch.ceil((ht - zt) / st))):
            want |= 2
        if not bool(torch.all((qr == qr.floor()) | (qr == qr.ceil()))):
            want |= 4
        assert int(flags.item()) == want, (n, s, zp, lo, hi, int(flags.item()), want)
        fin = qr[~qr.isnan()]
        if fin.numel():
            assert float(qstats[0]) == float(fin.min()) and float(qstats[1]) == float(fin.max())


def P(v):
    return torch.tensor([float(v)], device=DEV, requires_grad=True)


@pytest.mark.parametrize("method", ["STE", "LSQ", "EWGS"])
def test_hip_path_matches_the_plain_c_oracle(ops, method):
    """The torch-free checker (oracle/fq_ref.c, pinned by the reference's vectors in tests/test_oracle_c_golden.py):
    NoisyAct from its parameters and the per-channel / per-tensor weight path, forward and backward, on seeded
    tensors with clipping on both sides, ragged sizes and a tied minimum -- elementwise bit for bit, reduced gradients
    within 1e-6 of the sum of |terms|."""
    from oracle import fq_c
    gen = torch.Generator().manual_seed(17)
    x = torch.randn(3, 7, 5, 5, generator=gen) * 2
    g = torch.randn(3, 7, 5, 5, generator=gen)
    r = (torch.randint(0, 2, x.shape, generator=gen) * 2 - 1).to(torch.int8)
    ls, lq, b = P(-2.3), P(1.7), P(-1.4)
    xd = x.to(DEV).requires_grad_(True)
    y, params = ops.fake_quant_act_layer(xd, ls, lq, b, method, r_sign=r.to(DEV).reshape(-1))
    y.backward(g.to(DEV))
    s, qr = float(params[0]), float(params[4])
    c = fq_c.act(x.numpy(), g.numpy(), r.numpy(), s, qr, float(b.detach()), method)
    assert torch.equal(y.detach().cpu(), torch.from_numpy(c["y"]))
    assert torch.equal(xd.grad.cpu(), torch.from_numpy(c["gx"]))
    yard = 1e-6 * float(g.abs().sum()) * (qr / s + 1) * s + 1e-9
    assert abs(float(ls.grad) - c["g_log_act_s"]) <= yard and abs(float(lq.grad) - c["g_log_act_q"]) <= yard
    assert abs(float(b.grad) - c["g_act_b"]) <= 1e-6 * float(g.abs().sum()) * 2 + 1e-9
    for per_channel in (True, False):
        w = torch.randn(6, 5, 3, 3, generator=gen) * 0.1
        w[1, 0, 0, 0] = w[1].min()                      # a tied minimum: the amin backward splits its share
        w[1, 2, 1, 1] = w[1].min()
        G = torch.randn(6, 5, 3, 3, generator=gen)
        rw = (torch.randint(0, 2, w.shape, generator=gen) * 2 - 1).to(torch.int8)
        wd = w.to(DEV).requires_grad_(True)
        if per_channel:
            lws = (torch.full((6, 1, 1, 1), -5.0) + torch.randn(6, 1, 1, 1, generator=gen) * 0.3).to(DEV).requires_grad_(True)
            wq, zp, sc, lwq = ops.fake_quant_weight_layer(wd, lws, method, r_sign=rw.to(DEV).reshape(-1))
        else:
            lws = P(-5.2)
            wq, zp, sc, lwq = ops.fake_quant_weight_layer_pt(wd, lws, method, r_sign=rw.to(DEV).reshape(-1))
        wq.backward(G.to(DEV))
        cw = fq_c.weight(w.numpy(), G.numpy(), rw.numpy(), sc.detach().cpu().numpy().reshape(-1), per_channel, method)
        assert torch.equal(wq.detach().cpu(), torch.from_numpy(cw["wq"]))
        assert exact_off_extremes(wd.grad.cpu().numpy(), cw["gw"], w.numpy(), per_channel)
        # at the tied minima gW also carries a share of the REDUCED zero-point gradient sum(G - gv/s): 1e-6 of its terms
        grp = (G.abs().reshape(6, -1).sum(1).reshape(6, 1, 1, 1) if per_channel else G.abs().sum()).numpy() * 2
        assert bool((np.abs(wd.grad.cpu().numpy() - cw["gw"]) <= 1e-6 * grp + 1e-30).all())
        s_np = sc.detach().cpu().numpy().reshape(-1)
        qmax = float((w.max() - w.min()) / s_np.min()) + 1
        yard_w = 1e-6 * (G.abs().reshape(6, -1).sum(1).numpy() if per_channel else np.array([float(G.abs().sum())])) * s_np * qmax
        assert bool((np.abs(lws.grad.cpu().numpy().reshape(-1) - cw["g_log_wght_s"]) <= yard_w + 1e-9).all())


@pytest.mark.parametrize("per_channel", [True, False])
def test_hip_aewgs_weight_path_matches_the_plain_c_oracle(ops, per_channel):
    """AEWGS (gdnsq.py:113-147) against the torch-free checker: per-channel statistics, and the [1]-shaped-scale quirk
    (statistics per position over dim 0) for a per-tensor scale.  Forward bit for bit; the input gradient carries
    delta = num / max(e2 - me^2, 1e-3), whose three means the HIP path and the C oracle both take in fp64 -- so here,
    unlike against the reference's fp32 means, the agreement is at the last-bit level of delta."""
    from oracle import fq_c
    gen = torch.Generator().manual_seed(23)
    w = torch.randn(12, 6, 3, 3, generator=gen) * 0.1
    G = torch.randn(12, 6, 3, 3, generator=gen)
    rw = (torch.randint(0, 2, w.shape, generator=gen) * 2 - 1).to(torch.int8)
    wd = w.to(DEV).requires_grad_(True)
    if per_channel:
        lws = (torch.full((12, 1, 1, 1), -5.0) + torch.randn(12, 1, 1, 1, generator=gen) * 0.3).to(DEV).requires_grad_(True)
        wq, zp, sc, lwq = ops.fake_quant_weight_layer(wd, lws, "AEWGS", r_sign=rw.to(DEV).reshape(-1))
    else:
        lws = P(-5.2)
        sc = torch.exp2(lws)
        wq, zp = ops.fake_quant_weight_pt(wd, sc, "AEWGS", r_sign=rw.to(DEV).reshape(-1))
    wq.backward(G.to(DEV))
    s_np = sc.detach().cpu().numpy().reshape(-1)
    cw = fq_c.weight(w.numpy(), G.numpy(), rw.numpy(), s_np, per_channel, "AEWGS")
    assert torch.equal(wq.detach().cpu(), torch.from_numpy(cw["wq"]))
    Gabs = G.abs().numpy()
    mask = (w != (w.amin((1, 2, 3), keepdim=True) if per_channel else w.min())).numpy()     # off the tied minima
    err = np.abs(wd.grad.cpu().numpy() - cw["gw"])
    assert bool((err[mask] <= 2e-6 * Gabs[mask] + 1e-9).all()), float((err[mask] / (Gabs[mask] + 1e-12)).max())
    # at the tied minima gW also carries a share of the REDUCED zero-point gradient sum(G - gv/s): 1e-6 of its terms
    grp = (Gabs.reshape(12, -1).sum(1).reshape(12, 1, 1, 1) if per_channel else Gabs.sum()) * 2
    assert bool((err <= 2e-6 * Gabs + 1e-6 * grp + 1e-30).all())
    qmax = float((w.max() - w.min()) / s_np.min()) + 1
    yard = 1e-6 * (Gabs.reshape(12, -1).sum(1) if per_channel else np.array([float(Gabs.sum())])) * s_np * qmax
    assert bool((np.abs(lws.grad.cpu().numpy().reshape(-1) - cw["g_log_wght_s"]) <= yard + 1e-9).all())


def test_fill_r_is_the_documented_philox_stream(ops):
    """mhaq_fq_fill_r (and with it every kernel's in-kernel signs, which other tests hold equal to it) against the numpy
    restatement of the stream layout documented in include/mhaq_fq.h: ragged sizes, a 64-bit seed and offset."""
    from tests.philox_ref import signs
    for n, seed, off in ((1, 0, 0), (4099, 1234, 1), (1 << 16, 2024 ^ 0x9E3779B97F4A7C15, 7), (70001, (1 << 63) + 5, (1 << 40) + 3)):
        got = ops.fill_r(n, seed, off, DEV).cpu().numpy()
        assert np.array_equal(got, signs(n, seed, off)), (n, seed, off)


def test_per_channel_forward_equals_ieee_division_for_many_scales(ops):
    """The per-channel forward computes q through the backward's exact-quotient core (reciprocal estimate + two FMA
    corrections, fq_common.hpp quant_core_w) instead of the IEEE division: for 300 random per-channel scales over 12
    decades -- incl. an all-ones significand (the documented fallback to the division) and near-power-of-two ones -- on
    rows of both kernel forms (register-resident: 2048 floats; staged: 450 floats, not a multiple of 4), wq must equal
    torch's round((w - min) / s) * s + min bit for bit, and so must the zero points."""
    gen = torch.Generator().manual_seed(77)
    scales = torch.exp2(torch.rand(296, generator=gen) * 40 - 30).tolist()
    scales += [float(np.float32(np.nextafter(np.float32(2.0), np.float32(0.0)))) * 0.25,   # all-ones significand
               float(np.nextafter(np.float32(0.125), np.float32(1.0))), 0.1, 3.0]
    s = torch.tensor(scales, dtype=torch.float32)
    for row in (2048, 450):
        w = torch.randn(len(scales), row, generator=gen) * s[:, None] * 6     # ~ +-20 quantization steps, many .5 ties apart
        w[:, 3] = w[:, 0] + 2.5 * s                                            # exact half-way points (round-half-even)
        wd, sd = w.to(DEV), s.to(DEV)
        wq, zp = ops.fake_quant_weight_pc(wd, sd, "LSQ")[:2]
        mn = wd.amin(1, keepdim=True)
        v = (wd - mn) / sd[:, None]
        want = (v + (torch.round(v) - v)) * sd[:, None] + mn
        assert torch.equal(zp.reshape(-1), mn.reshape(-1))
        assert torch.equal(wq.detach(), want), (row, int((wq.detach() != want).sum()))
