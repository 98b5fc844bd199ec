"""GPU (-m gpu): PER_TENSOR weight layers beyond one workgroup, PER_TENSOR AEWGS layers and PER_CHANNEL layers with a
quantized bias through their fused layer ops (mhaq_fq_wlayer_ptl_fwd / _bwd; mhaq_fq_wlayer_* with a differentiable
s / zp) -- SURVEY.md 8(f) rank 1 off the BASELINE shapes: the weight quantizer AND the regulariser input
log2(max - min + s) of ModelHelper.get_model_values (utils/model_helper.py:24-45) with its amin / amax backward, no
torch amin / amax sweep left.  Checker: the eager oracle executed on the same device with explicit random signs."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fq_closed_form as CF  # noqa: E402
from oracle import fq_eager as O  # noqa: E402
from tests.golden_util import bit_equal, exact_off_extremes  # noqa: E402

DEV = "cuda:0"
LARGE = [(512, 512, 3, 3), (128, 64, 3, 3), (1000, 512), (70, 1001), (3, 40000)]


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from mhaq_amd import _lib, ops
    _lib.lib()
    return ops


def _case(shape, seed, ties=True):
    gen = torch.Generator().manual_seed(seed)
    fan = int(np.prod(shape[1:]))
    w = torch.randn(*shape, generator=gen) * math.sqrt(2.0 / fan)
    if ties:
        w.flatten()[[1, 5, 4097]] = w.min() - 0.01            # tied global minima and maxima
        w.flatten()[[2, 3, 7, 9001]] = w.max() + 0.02
    G = torch.randn(*shape, generator=gen)
    h = torch.randn(1, generator=gen)
    r = torch.randint(0, 2, shape, generator=gen).float() - 0.5
    return tuple(t.to(DEV) for t in (w, G, h, r))


@pytest.mark.parametrize("method", ["STE", "LSQ", "EWGS", "AEWGS"])
@pytest.mark.parametrize("shape", LARGE)
def test_large_per_tensor_layer_matches_the_eager_oracle(ops, method, shape):
    w, G, h, r = _case(shape, shape[0] * 5 + len(shape))
    ls0 = (torch.log2((w.max() - w.min()) / 15.0).reshape(1) + 0.137)
    assert method == "AEWGS" or not ops.small_pt_layer_supported(w, method)
    wr, lsr = w.clone().requires_grad_(True), ls0.clone().requires_grad_(True)
    wq_r, _, zp_r = O.weight_fake_quant(wr, lsr, False, method, r=r)
    lwq_r = torch.log2(wr.amax() - wr.amin() + torch.exp2(lsr.ravel()))
    ((wq_r * G).sum() + (lwq_r * h).sum()).backward()
    wg, lsg = w.clone().requires_grad_(True), ls0.clone().requires_grad_(True)
    wq, zp, s, lwq = ops.fake_quant_weight_layer_ptl(wg, lsg, method, r_sign=(r * 2).to(torch.int8))
    ((wq * G).sum() + (lwq * h).sum()).backward()
    assert torch.equal(s, torch.exp2(ls0)) and torch.equal(zp, zp_r.detach()) and zp.dim() == 0
    assert bit_equal(wq.detach().cpu().numpy(), wq_r.detach().cpu().numpy())
    assert bit_equal(lwq.detach().cpu().numpy(), lwq_r.detach().cpu().numpy())
    gw, gw_r = wg.grad.cpu().numpy(), wr.grad.cpu().numpy()
    if method != "AEWGS":
        # elementwise part (G*s [+ estimator]) / s: exact wherever no reduced share lands (off the tied extremes)
        assert exact_off_extremes(gw, gw_r, w.cpu().numpy(), False, also_max=True), "gw off the extremes"
    cf = CF.per_channel(w.reshape(1, -1).cpu(), G.reshape(1, -1).cpu(), r.reshape(1, -1).cpu(), s.cpu(),
                        "STE" if method == "AEWGS" else method)
    abs_g = float(cf["abs_g"]) + abs(float(h)) * 4
    if method == "AEWGS":
        # per-position statistics (means over dim 0, fp64 here / fp32 in torch) feed delta: the propagated slack of
        # the three means, as in tests/test_gpu_parity.py
        v = (w - zp) / s
        e = torch.round(v) - v
        co = shape[0]
        mean64 = lambda t: (t.double().sum(0, keepdim=True).float() / float(co))  # noqa: E731
        num, e2, me = mean64((G * s).sign() * e), mean64(e * e), mean64(e)
        den = (e2 - me * me).clamp_min(1e-3)
        ddelta = 1e-6 * (mean64(e.abs()) / den + num.abs() * (e2 + 2 * me.abs() * mean64(e.abs())) / den ** 2)
        tol = (G.abs() * (e.abs() * ddelta + 1e-6)).cpu().numpy()
        mask = np.ones(shape, dtype=bool)
        wn = w.cpu().numpy()
        mask &= (wn != wn.min()) & (wn != wn.max())
        assert np.all(np.abs(gw - gw_r)[mask] <= tol[mask] + 1e-30), "AEWGS gw off the extremes"
        abs_g = abs_g + float(tol.sum())
    err = np.abs(gw - gw_r)
    assert np.all(err <= 1e-6 * (abs_g + np.abs(gw_r))), err.max()          # the tie-split shares at the extremes
    yard = (float(cf["abs_s"]) + abs(float(h)) * 4) * math.log(2.0) * float(s) * 2
    if method == "AEWGS":
        yard = yard * 4
    assert abs(float(lsg.grad) - float(lsr.grad)) <= 1e-6 * yard + 1e-9


@pytest.mark.parametrize("method", ["STE", "LSQ", "EWGS", "AEWGS"])
def test_streaming_layer_equals_the_unfused_per_tensor_op_bit_for_bit(ops, method):
    """Without a regulariser gradient the streaming layer op is the unfused per-tensor op (minmax + pt_fwd + pt_bwd +
    tie scatter, ops.fake_quant_weight_pt: the path held to the reference's golden vectors) launch for launch: same wq,
    same gW, and dL/dlog_s = (dL/ds * s) * ln2 of the same dL/ds.  Power-of-two scale: exp2 is exact on both sides."""
    w, G, h, r = _case((128, 64, 3, 3), 11)
    r8 = (r * 2).to(torch.int8)
    ls0 = torch.tensor([-6.0], device=DEV)
    w1, l1 = w.clone().requires_grad_(True), ls0.clone().requires_grad_(True)
    wq1, zp1, s1, _ = ops.fake_quant_weight_layer_ptl(w1, l1, method, r_sign=r8)
    (wq1 * G).sum().backward()
    w2, l2 = w.clone().requires_grad_(True), ls0.clone().requires_grad_(True)
    wq2, zp2 = ops.fake_quant_weight_pt(w2, torch.exp2(l2), method, r_sign=r8)
    (wq2 * G).sum().backward()
    assert torch.equal(wq1, wq2) and torch.equal(zp1, zp2)
    assert torch.equal(w1.grad, w2.grad)
    assert torch.equal(l1.grad, l2.grad)


def test_large_per_tensor_layer_on_a_channels_last_weight(ops):
    w, G, h, r = _case((128, 64, 3, 3), 13)
    ls0 = torch.tensor([-5.0], device=DEV)
    wa, la = w.clone().requires_grad_(True), ls0.clone().requires_grad_(True)
    wqa, zpa, sa, lwqa = ops.fake_quant_weight_layer_ptl(wa, la, "LSQ")
    ((wqa * G).sum() + (lwqa * h).sum()).backward()
    wc = w.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    lc = ls0.clone().requires_grad_(True)
    wqc, zpc, sc, lwqc = ops.fake_quant_weight_layer_ptl(wc, lc, "LSQ")
    assert wqc.is_contiguous(memory_format=torch.channels_last)
    ((wqc * G).sum() + (lwqc * h).sum()).backward()
    assert torch.equal(wqc, wqa) and torch.equal(zpc, zpa) and torch.equal(lwqc, lwqa)
    off = (w != w.min()) & (w != w.max())
    assert torch.equal(wc.grad[off], wa.grad[off]) and wc.grad.is_contiguous(memory_format=torch.channels_last)
    assert torch.allclose(wc.grad, wa.grad, rtol=1e-5, atol=1e-6)          # tie shares: same terms, another order
    assert torch.allclose(lc.grad, la.grad, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("per_channel_with_quantized_bias", [False, True])
def test_no_layer_of_the_product_leaves_its_regulariser_input_to_torch(ops, per_channel_with_quantized_bias):
    """get_model_values (wrap.py) takes log2(max - min + s) from every mhaq_amd layer's own forward -- small and
    ResNet-18-sized PER_TENSOR layers, PER_CHANNEL layers with a quantized bias -- and the model-level gradient (task loss
    + a hinge on lwq - lws, as PotentialLoss applies it) equals the one the oracle's layers give.  LSQ: nothing random."""
    import mhaq_amd as M
    from mhaq_amd import wrap
    from oracle import ref_layers as R
    torch.manual_seed(5)
    qb = per_channel_with_quantized_bias
    qs = M.QScheme.PER_CHANNEL if qb else M.QScheme.PER_TENSOR
    shapes = [(8, 3, 3), (96, 8, 3), (128, 96, 3)]           # last one: 110 592 weights > one workgroup's 64 K
    hip = torch.nn.Sequential(*[M.NoisyConv2d(ci, co, k, padding=1, bias=True, qscheme=qs, log_s_init=-6,
                                              quant_bias=qb, qnmethod=M.QNMethod.LSQ)
                                for co, ci, k in shapes]).to(DEV)
    ref = torch.nn.Sequential(*[R.NoisyConv2d(ci, co, k, padding=1, bias=True, qscheme=qs.value, log_s_init=-6,
                                              quant_bias=qb, qnmethod="LSQ")
                                for co, ci, k in shapes]).to(DEV)
    with torch.no_grad():
        for a, b in zip(hip, ref):
            b.weight.copy_(a.weight)
            b.bias.copy_(a.bias)
            b.log_wght_s.copy_(a.log_wght_s)
    x = torch.randn(2, 3, 10, 10, device=DEV)
    outs = []
    for net in (hip, ref):
        y = net(x)
        if net is hip:
            assert all(m.regulariser_input() is not None for m in hip), "a layer left lwq to torch amin / amax"
        lws, lwq = _weight_values(wrap, net, qs)
        loss = y.square().mean() + (lwq - lws).clamp_min(3.0).sum() * 0.1
        loss.backward()
        outs.append((y.detach(), lwq.detach(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}))
    (y1, q1, g1), (y2, q2, g2) = outs
    assert torch.equal(q1, q2)
    assert torch.allclose(y1, y2, rtol=1e-5, atol=1e-6)
    assert set(g1) == set(g2)
    for n in g1:
        a, b = g1[n], g2[n]
        assert torch.allclose(a, b, rtol=2e-4, atol=1e-5 * float(b.abs().max()) + 1e-9), (n, float((a - b).abs().max()))


def _weight_values(wrap, net, qs):
    """The weight half of get_model_values (these toy nets have no NoisyAct: torch.cat of nothing would raise)."""
    import mhaq_amd as M
    lws, lwq = [], []
    for m in net:
        fused = m.regulariser_input() if hasattr(m, "regulariser_input") else None
        if qs == M.QScheme.PER_CHANNEL:
            lws.append(m.log_wght_s.ravel())
            lwq.append(fused if fused is not None else torch.log2(
                m.weight.amax((1, 2, 3)) - m.weight.amin((1, 2, 3)) + torch.exp2(m.log_wght_s.ravel())))
        else:
            lws.append(m.log_wght_s.ravel())
            lwq.append(fused if fused is not None else torch.log2(
                m.weight.amax() - m.weight.amin() + torch.exp2(m.log_wght_s.ravel())))
    return torch.cat(lws), torch.cat(lwq)


def test_regulariser_inputs_are_published_for_every_weight_route(ops):
    import mhaq_amd as M
    x = torch.randn(2, 8, 6, 6, device=DEV)
    for kw in (dict(qscheme=M.QScheme.PER_TENSOR, qnmethod=M.QNMethod.AEWGS),            # the reference's defaults
               dict(qscheme=M.QScheme.PER_TENSOR, qnmethod=M.QNMethod.STE),
               dict(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.AEWGS, quant_bias=True),
               dict(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.LSQ)):
        conv = M.NoisyConv2d(8, 16, 3, log_s_init=-6, **kw).to(DEV)
        conv(x).sum().backward()
        lwq = conv.regulariser_input()
        assert lwq is not None and lwq.requires_grad
        if kw["qscheme"] == M.QScheme.PER_CHANNEL:
            want = torch.log2(conv.weight.amax((1, 2, 3)) - conv.weight.amin((1, 2, 3))
                              + torch.exp2(conv.log_wght_s.ravel()))
        else:
            want = torch.log2(conv.weight.amax() - conv.weight.amin() + torch.exp2(conv.log_wght_s)).reshape(1)
        assert torch.equal(lwq.detach(), want.detach())
        assert conv.weight.grad is not None and torch.isfinite(conv.weight.grad).all()
        assert torch.isfinite(conv.log_wght_s.grad).all()


# ------------------------------------------------------------------ the reference's own per-tensor vectors
from tests.golden_util import T, load_cases, r_from_sign  # noqa: E402

WGT_PT = {k: v for k, v in {**load_cases("weight_cases.npz"), **load_cases("ewgs_weight_cases.npz")}.items()
          if not bool(v["per_channel"])}


@pytest.mark.parametrize("name", sorted(WGT_PT))
def test_streaming_per_tensor_layer_matches_the_reference_golden(ops, name):
    """Every PER_TENSOR weight case recorded from the reference (STE / LSQ / AEWGS / EWGS, tied minima, Linear, the wide one)
    through the streaming layer op: wq and zp bit for bit; gW value-equal off the tied minima for STE / LSQ / EWGS and within the
    propagated slack of the per-position means for AEWGS; dL/dlog_wght_s within 1e-6 of its terms.  (Integer-valued
    log scales in the fixtures' per-tensor cases would make exp2 exact; where the device's exp2 differs from the host's by an
    ulp the case is checked with the device's scale bits through the oracle instead.)"""
    c = WGT_PT[name]
    method = O.METHODS[int(c["method"])]
    w, G, r = T(c["w"]), T(c["G"]), r_from_sign(c["r"])
    ls = T(c["log_wght_s"]).reshape(1)
    wd, lsd = w.to(DEV).requires_grad_(True), ls.to(DEV).requires_grad_(True)
    wq, zp, s, lwq = ops.fake_quant_weight_layer_ptl(wd, lsd, method, r_sign=torch.from_numpy(c["r"].astype(np.int8)).to(DEV))
    wq.backward(G.to(DEV))
    s_host = torch.exp2(ls)
    if torch.equal(s.cpu(), s_host):
        ref_wq, ref_zp, ref_gw, ref_gls = c["wq"], c["zp"], c["gw"], c["g_log_wght_s"]
    else:       # one ulp between the host's and the device's exp2: compare with the oracle at the device's scale bits
        wr = w.clone().requires_grad_(True)
        sr = s.detach().cpu().clone().requires_grad_(True)
        zpr = O.weight_zero_point(wr, False)
        wq_r = O.dequantize(O.quantize(wr, sr, zpr, -math.inf, math.inf, method, r), sr, zpr)
        wq_r.backward(G)
        ref_wq, ref_zp, ref_gw = wq_r.detach().numpy(), zpr.detach().numpy(), wr.grad.numpy()
        ref_gls = (sr.grad * sr.detach() * math.log(2.0)).numpy()
    assert bit_equal(wq.detach().cpu().numpy(), ref_wq)
    assert bit_equal(zp.detach().cpu().numpy().reshape(np.shape(ref_zp)), ref_zp)
    gw = wd.grad.cpu().numpy()
    cf = CF.per_channel(w.reshape(1, -1), G.reshape(1, -1), r.reshape(1, -1), s.detach().cpu().reshape(1),
                        "STE" if method == "AEWGS" else method)
    abs_g, abs_s = float(cf["abs_g"]), float(cf["abs_s"])
    if method in ("STE", "LSQ", "EWGS"):
        assert exact_off_extremes(gw, ref_gw, c["w"], False), "gw off the minima"
        assert np.all(np.abs(gw - ref_gw) <= 1e-6 * (abs_g + np.abs(ref_gw)))
        rel, slack = 1e-6, 0.0
    else:
        sd = s.detach().cpu()
        v = (w - float(ref_zp)) / sd
        e = torch.round(v) - v
        co = w.shape[0]
        mean64 = lambda t: (t.double().sum(0, keepdim=True).float() / float(co))  # noqa: E731
        num, e2, me = mean64((G * sd).sign() * e), mean64(e * e), mean64(e)
        den = (e2 - me * me).clamp_min(1e-3)
        ddelta = 1e-6 * (mean64(e.abs()) / den + num.abs() * (e2 + 2 * me.abs() * mean64(e.abs())) / den ** 2)
        tol = (G.abs() * (e.abs() * ddelta + 1e-6)).numpy()
        off = np.asarray(c["w"]) != np.asarray(c["w"]).min()
        assert np.all(np.abs(gw - ref_gw)[off] <= tol[off] + 1e-30), "AEWGS gw off the minima"
        assert np.all(np.abs(gw - ref_gw) <= tol + 1e-6 * (abs_g + float(tol.sum()) + np.abs(ref_gw)))
        rel = 1e-6
        # d/dlog_s = d/ds * s ln2 sums (gv / s) * v over the tensor: it inherits sum tol * |v| (tests/aewgs_bound.py)
        slack = float((tol.astype(np.float64) * np.abs(v.numpy())).sum()) * float(s) * math.log(2.0)
    yard = abs_s * math.log(2.0) * float(s) * 2
    assert abs(float(lsd.grad) - float(np.asarray(ref_gls).reshape(-1)[0])) <= rel * yard + slack + 1e-12
