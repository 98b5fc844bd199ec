"""GPU (-m gpu): the product's MODULES on the cases recorded from the reference's modules.

tests/test_gpu_parity.py holds the OPS to the golden vectors (scales injected with the host's bits).  Here the drop-in objects
themselves -- mhaq_amd.NoisyAct / NoisyConv2d / NoisyLinear, constructed with the reference's arguments and given the
fixture's parameters -- run their own forward (exp2 of the log-parameters, zero point, clamp bounds: all inside the fused
layer kernels) and backward (in-kernel Philox signs), and are compared with what the reference's modules produced
(tests/golden/*.npz, oracle/gen_golden.py: /root/reference/src/quantization/gdnsq/layers/*.py run unchanged):

  * forward outputs bit for bit, input / weight gradients value for value off the tied extremes (AEWGS: within the propagated
    slack of its fp64-vs-fp32 group means), whenever the device's exp2 gives the scale bits the reference's host exp2 gave
    (always for integer log-parameters; `same_bits` below) -- otherwise against the eager oracle evaluated at the DEVICE's
    scale bits (a one-ulp different scale is a different quantizer);
  * the learnable-scale gradients within 1e-6 * sum|terms| AFTER accounting for the random +-0.5 signs: the module draws its
    own stream, the fixture carries the reference's draw; the stream the module used is replayed with mhaq_fq_fill_r and the
    difference of the two noise terms, 3^-1/2 * sum gq * (r_module - r_fixture), is added to the reference's value (LSQ has no
    random term and is compared directly).

11 of the 22 activation cases and 17 of the 31 weight cases have integer log-parameters, i.e. are compared with the reference's
recorded vectors on every device; the others are whenever the device's exp2 rounds like the host's.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fq_closed_form as CF  # noqa: E402
from oracle import fq_eager as O  # noqa: E402
from tests.aewgs_bound import aewgs_gx_bound, aewgs_slack, aewgs_weight_slacks, within  # noqa: E402
from tests.golden_util import T, bit_equal, exact_off_extremes, load_cases, r_from_sign, value_equal  # noqa: E402

ACT = {**load_cases("act_cases.npz"), **load_cases("ewgs_act_cases.npz")}
WGT = {**load_cases("weight_cases.npz"), **load_cases("ewgs_weight_cases.npz")}
DEV = "cuda:0"
SEED = 20251005
INV_SQRT3 = 3.0 ** -0.5
LN2 = math.log(2.0)


@pytest.fixture(scope="module")
def M():
    import mhaq_amd as M
    return M


def _replayed_signs(n, offset):
    """The +-0.5 tensor the module's backward drew: stream (SEED, offset) of this process (rank 0)."""
    from mhaq_amd import ops
    return ops.fill_r(n, SEED, offset, DEV).float().cpu() * 0.5


def _reduced(got, ref, yard, what, rel=1e-6, slack=0.0):
    """|got - ref| <= rel * sum|terms| + slack; slack = the propagated AEWGS term (tests/aewgs_bound.py), derived at the call"""
    got, ref, yard = (np.asarray(a, dtype=np.float64) for a in (got, ref, yard))
    err = np.abs(got - ref)
    assert np.all(err <= rel * yard + slack + 1e-30), f"{what}: err {err.max():.3e} > {rel:g} * sum|terms| {yard.max():.3e}"


# ------------------------------------------------------------------------------ NoisyAct
@pytest.mark.parametrize("name", sorted(ACT))
def test_noisy_act_module_matches_the_reference_module(M, name):
    from mhaq_amd import ops
    c = ACT[name]
    method = O.METHODS[int(c["method"])]
    signed = bool(c["signed"])
    act = M.NoisyAct(signed=signed, qnmethod=M.QNMethod[method])
    with torch.no_grad():
        act.log_act_s.fill_(float(c["log_act_s"]))
        act.log_act_q.fill_(float(c["log_act_q"]))
        act.act_b.fill_(float(c["act_b"]))
    act = act.to(DEV).train()
    assert act.act_b.requires_grad == signed                          # gdnsq_act.py:28-31
    x, g = T(c["x"]), T(c["g"])
    xg = x.to(DEV).requires_grad_(True)
    ops.manual_seed(SEED)
    y = act(xg)
    drawn = ops.rng.drawn()
    y.backward(g.to(DEV))
    assert ops.rng.drawn() - drawn == (0 if method == "LSQ" else 1)
    # the reference's scalar chain on the host (gdnsq_act.py:42-47) against what the kernel derived on the device
    ls, lq, b = (T(c[k]).reshape(1) for k in ("log_act_s", "log_act_q", "act_b"))
    s_h, qr_h = torch.exp2(ls), torch.exp2(lq)
    hi_h = b + qr_h - s_h
    s_d, hi_d = act.Q.scale.detach().cpu().reshape(1), act.Q.max_val.detach().cpu().reshape(1)
    same_bits = torch.equal(s_d, s_h) and torch.equal(hi_d, hi_h)
    if float(ls) == round(float(ls)) and float(lq) == round(float(lq)):
        assert same_bits
    r_fix = r_from_sign(c["r"])
    r_mod = _replayed_signs(x.numel(), drawn + 1).reshape(x.shape) if method != "LSQ" else r_fix
    if same_bits:
        y_ref, gx_ref = c["y"], c["gx"]
        gls_ref, glq_ref, gb_ref = (float(np.asarray(c[k]).reshape(-1)[0]) for k in ("g_log_act_s", "g_log_act_q", "g_act_b"))
        # the reference's draw -> the module's draw in the scale gradient's noise term (gdnsq.py:53-55; s * ln2: exp2 backward)
        if method != "LSQ":
            gls_ref += float(((INV_SQRT3 * (g * s_h)).double() * (r_mod - r_fix).double()).sum()) * float(s_h) * LN2
        s_e, hi_e, qr_e = s_h, hi_h, qr_h
    else:
        # the device's exp2 is an ulp away from the host's: the same case through the eager oracle at the device's bits
        qr_e = (hi_d - b) + s_d                     # hi = (b + qr) - s; only used for yardsticks
        xs = x.clone().requires_grad_(True)
        P = [t.clone().requires_grad_(True) for t in (s_d, b, b, hi_d)]
        y_o = O.dequantize(O.quantize(xs, P[0], P[1], P[2], P[3], method, r_mod), P[0], P[1])
        y_o.backward(g)
        y_ref, gx_ref = y_o.detach().numpy(), xs.grad.numpy()
        gs, gzp, glo, ghi = (float(p.grad) for p in P)
        qr_d = float(torch.exp2(lq))                # chain back through hi = b + qr - s, s = 2^ls, qr = 2^lq
        gls_ref, glq_ref, gb_ref = (gs - ghi) * float(s_d) * LN2, ghi * qr_d * LN2, gzp + glo + ghi
        s_e, hi_e = s_d, hi_d
    assert bit_equal(y.detach().cpu().numpy(), y_ref)
    if method == "AEWGS":
        v = (torch.clamp(x, b, hi_e) - b) / s_e
        assert within(xg.grad.cpu(), torch.from_numpy(np.asarray(gx_ref)), aewgs_gx_bound(v, g, (0,)))
    else:
        assert value_equal(xg.grad.cpu().numpy(), gx_ref)
    cf = CF.per_tensor(x, g, r_mod, s_e, b, b, hi_e, "STE" if method == "AEWGS" else method)
    # 1e-6 * sum|terms|; AEWGS + the propagated slack of its group means (aewgs_slack): log_act_s = (d/ds - d/dhi) s ln2
    # with d/ds summing (gv / s) v and d/dhi summing gv / s above hi; log_act_q = d/dhi qr ln2; act_b = d/dzp + d/dlo + d/dhi
    sl_v = sl_all = sl_over = 0.0
    if method == "AEWGS":
        sl_v, sl_all = float(aewgs_slack(v, g, (0,), weight=v)), float(aewgs_slack(v, g, (0,)))
        sl_over = float(aewgs_slack(v, g, (0,), where=(x > hi_e)))
    yard_s = (float(cf["abs_s"]) + float(cf["abs_g"])) * LN2 * float(s_e)
    _reduced(float(act.log_act_s.grad), gls_ref, yard_s, "g_log_act_s", slack=(sl_v + sl_over) * LN2 * float(s_e))
    _reduced(float(act.log_act_q.grad), glq_ref, float(cf["abs_g"]) * LN2 * float(qr_e), "g_log_act_q",
             slack=sl_over * LN2 * float(qr_e))
    if signed:
        _reduced(float(act.act_b.grad), gb_ref, float(cf["abs_g"]), "g_act_b", slack=2 * sl_all)
    else:
        assert act.act_b.grad is None
    # eval mode: the same forward, the bit width of gdnsq_act.py:51-54 and the asserts of gdnsq.py:211-217 (lazily)
    act.eval()
    with torch.no_grad():
        ye = act(x.to(DEV))
    if c["eval_raises"]:
        with pytest.raises(AssertionError):
            act.Q.check_integrity()
    else:
        act.Q.check_integrity()
        if same_bits:
            assert bit_equal(ye.cpu().numpy(), c["y_eval"])
            assert abs(float(act.bw) - float(np.asarray(c["bw"]).reshape(-1)[0])) <= 1e-6


# ------------------------------------------------------------------------------ NoisyConv2d / NoisyLinear
def _build_weight_module(M, c):
    w = T(c["w"])
    method = O.METHODS[int(c["method"])]
    pc = bool(c["per_channel"])
    qs = M.QScheme.PER_CHANNEL if pc else M.QScheme.PER_TENSOR
    qn = M.QNMethod[method]
    if w.dim() == 2:
        m = M.NoisyLinear(w.shape[1], w.shape[0], bias=False, qscheme=qs, qnmethod=qn)
    else:
        m = M.NoisyConv2d(w.shape[1], w.shape[0], tuple(w.shape[2:]), bias="bias" in c, qscheme=qs, qnmethod=qn,
                          quant_bias="bias" in c)
    with torch.no_grad():
        m.weight.copy_(w)
        m.log_wght_s.copy_(T(c["log_wght_s"]).reshape(m.log_wght_s.shape))
        if "bias" in c:
            m.bias.copy_(T(c["bias"]))
    return m.to(DEV).train(), method, pc


@pytest.mark.parametrize("name", sorted(WGT))
def test_weight_modules_match_the_reference_modules(M, name):
    from mhaq_amd import ops
    c = WGT[name]
    m, method, pc = _build_weight_module(M, c)
    w, G = T(c["w"]), T(c["G"])
    has_bias = "bias" in c
    captured = {}
    ops.manual_seed(SEED)
    if has_bias:                       # as oracle/gen_golden.py records it: what the convolution would be handed
        def conv_forward(inp, weight, b):
            captured["w"], captured["b"] = weight, b
            return weight
        m._conv_forward = conv_forward
        m(torch.zeros(1, w.shape[1], 8, 8, device=DEV))
        wq, bq = captured["w"], captured["b"]
    else:
        wq, _, _ = m._quantized_weight()
    drawn = ops.rng.drawn()
    if has_bias:
        torch.autograd.backward([wq, bq], [G.to(DEV), T(c["Gb"]).to(DEV)])
    else:
        wq.backward(G.to(DEV))
    random_estimator = method != "LSQ"
    assert ops.rng.drawn() - drawn == (0 if not random_estimator else (2 if has_bias else 1))
    s_h = torch.exp2(T(c["log_wght_s"]))                                   # gdnsq_conv2d.py:72 on the host
    s_d = m.Q.scale.detach().cpu().reshape(s_h.shape)
    same_bits = torch.equal(s_d, s_h)
    lsv = np.asarray(c["log_wght_s"]).reshape(-1)
    if np.all(lsv == np.round(lsv)):
        assert same_bits
    r_fix = r_from_sign(c["r"])
    # one stream per backward op; with a quantized bias the order of the two draws is autograd's: not replayed
    replay = random_estimator and not has_bias
    r_mod = _replayed_signs(w.numel(), drawn + 1).reshape(w.shape) if replay else r_fix
    co = w.shape[0]
    if same_bits:
        wq_ref, zp_ref, gw_ref = c["wq"], c["zp"], c["gw"]
        gls_ref = np.asarray(c["g_log_wght_s"], dtype=np.float64).reshape(-1).copy()
        if replay:
            sv = s_h.reshape(-1, *([1] * (w.dim() - 1))) if pc else s_h.reshape(())
            t = ((INV_SQRT3 * (G * sv)).double() * (r_mod - r_fix).double())
            t = t.reshape(co, -1).sum(1) if pc else t.sum().reshape(1)
            gls_ref += (t * s_h.reshape(-1).double() * LN2).numpy()
        s_e = s_h
    else:
        wr = w.clone().requires_grad_(True)
        sr = s_d.clone().reshape([co] + [1] * (w.dim() - 1) if pc else [1]).requires_grad_(True)
        zpr = O.weight_zero_point(wr, pc)
        wq_o = O.dequantize(O.quantize(wr, sr, zpr, -math.inf, math.inf, method, r_mod), sr, zpr)
        outs, grads = [wq_o], [G]
        if has_bias:
            br = T(c["bias"]).requires_grad_(True)
            bq_o = O.dequantize(O.quantize(br, sr.ravel(), zpr.ravel(), -math.inf, math.inf, method, r_from_sign(c["rb"])),
                                sr.ravel(), zpr.ravel())
            outs.append(bq_o)
            grads.append(T(c["Gb"]))
        torch.autograd.backward(outs, grads)
        wq_ref, zp_ref, gw_ref = wq_o.detach().numpy(), zpr.detach().numpy(), wr.grad.numpy()
        gls_ref = (sr.grad.reshape(-1) * s_d.reshape(-1) * LN2).double().numpy()
        s_e = s_d
    assert bit_equal(wq.detach().cpu().numpy(), wq_ref)
    assert bit_equal(m.Q.zero_point.detach().cpu().numpy().reshape(np.shape(zp_ref)), zp_ref)
    gw = m.weight.grad.cpu().numpy()
    # yardsticks from the closed form (per channel, or the whole tensor for a per-tensor scale)
    if pc:
        cf = CF.per_channel(w, G, r_mod, s_e.reshape(-1), "STE" if method == "AEWGS" else method)
        abs_g = cf["abs_g"].reshape([-1] + [1] * (w.dim() - 1)).numpy()
        abs_s = cf["abs_s"].numpy()
    else:
        cf = CF.per_channel(w.reshape(1, -1), G.reshape(1, -1), r_mod.reshape(1, -1), s_e.reshape(1),
                            "STE" if method == "AEWGS" else method)
        abs_g, abs_s = float(cf["abs_g"]), np.array([float(cf["abs_s"])])
    if has_bias:
        abs_g = abs_g + np.abs(c["Gb"]).reshape(abs_g.shape) * 2
        if same_bits:
            assert bit_equal(bq.detach().cpu().numpy(), c["bq"])
            assert np.allclose(m.bias.grad.cpu().numpy(), c["gbias"], rtol=1e-6, atol=1e-7)
    sl_gw, sl_ls = 0.0, 0.0
    if method != "AEWGS":
        assert exact_off_extremes(gw, gw_ref, c["w"], pc), "gw off the minima"
    else:       # fp64 group means here, fp32 in the reference: their propagated slack (tests/test_gpu_aewgs_apply_exact.py pins the rest)
        sl_gw, sl_ls = aewgs_weight_slacks(w, G, s_e, pc)
    _reduced(gw, gw_ref, abs_g + np.abs(gw_ref), "gw", slack=sl_gw)
    if replay or not random_estimator:
        yard = abs_s.reshape(-1) * LN2 * s_e.reshape(-1).numpy() * 2
        if has_bias:
            yard = yard + (np.abs(c["Gb"]) * np.abs(c["bq"]) * 4).reshape(yard.shape)
        _reduced(m.log_wght_s.grad.cpu().numpy().reshape(-1), gls_ref, yard, "g_log_wght_s", slack=sl_ls)
    # the regulariser input of ModelHelper.get_model_values (model_helper.py:24-44), published by the same launch
    lwq = m.regulariser_input()
    assert lwq is not None
    w2 = w.reshape(co, -1) if pc else w.reshape(1, -1)
    ref_lwq = torch.log2((w2.amax(1) - w2.amin(1)) + s_e.reshape(-1))
    assert torch.allclose(lwq.detach().cpu().reshape(-1), ref_lwq, rtol=0, atol=2e-6)


# ------------------------------------------------------------------------------ model level (round 6)
MODEL = load_cases("model_cases.npz")


@pytest.mark.parametrize("through", ["layer_ops", "model_wide_launch"])
@pytest.mark.parametrize("name", sorted(MODEL))
def test_model_level_vectors_through_the_hip_layers(M, name, through):
    """ModelHelper.get_model_values + PotentialLoss(NoPred) of the reference's 2-conv toy (utils/model_helper.py:13-76,
    gdnsq_loss.py:32-86,114-168; tests/golden/model_cases.npz) with the toy built from the PRODUCT's layers on the GPU:
    the regulariser inputs lwq = log2(max - min + s) come out of the layers' own fused weight launches (per layer, or the
    trainer's model-wide launch + grouped backward), the hinge out of the fused loss kernel, and the gradient of lwq reaches
    gW through the kernels' amin / amax tie split.  Rounds 1-5 held these vectors to the oracle's CPU layers only."""
    from mhaq_amd import wrap
    from mhaq_amd.loss import FusedPotentialLoss, FusedPotentialLossNoPred
    from mhaq_amd.multi import MultiTensorWeightQuant
    c = MODEL[name]
    pc = bool(c["per_channel"])
    qs = M.QScheme.PER_CHANNEL if pc else M.QScheme.PER_TENSOR
    net = torch.nn.Sequential(
        M.NoisyAct(signed=True), M.NoisyConv2d(3, 6, 3, padding=1, qscheme=qs, qnmethod=M.QNMethod.LSQ),
        torch.nn.ReLU(),
        M.NoisyAct(signed=False), M.NoisyConv2d(6, 4, 3, padding=1, qscheme=qs, qnmethod=M.QNMethod.LSQ)).to(DEV)
    convs = [m for m in net if isinstance(m, M.NoisyConv2d)]
    acts = [m for m in net if isinstance(m, M.NoisyAct)]
    with torch.no_grad():
        for i, m in enumerate(convs):
            m.weight.copy_(T(c[f"w{i}"], DEV))
            m.log_wght_s.copy_(T(c[f"log_wght_s{i}"], DEV).view_as(m.log_wght_s))
        for i, a in enumerate(acts):
            a.log_act_s.copy_(T(c[f"log_act_s{i}"], DEV))
            a.log_act_q.copy_(T(c[f"log_act_q{i}"], DEV))
    if through == "model_wide_launch":
        plan = MultiTensorWeightQuant(net, joint_backward=False, backward_group_elems=300)      # both layers: one backward group
        plan.run()
    wqs = [m._quantized_weight()[0] for m in convs]          # each layer's fused forward (or its slice of the plan's)
    las, laq, lws, lwq = wrap.get_model_values(net, qs)
    assert all(m.regulariser_input() is not None for m in convs)        # lwq is the kernels', not a torch amin / amax
    for got, key in ((las, "las"), (laq, "laq"), (lws, "lws")):
        assert bit_equal(got.detach().cpu().numpy(), c[key]), key       # the parameters themselves
    # lwq = log2((max - min) + 2^log_s): the device's exp2 / log2 against the host's -- one ulp of the result each
    from tests.golden_util import max_ulp
    assert max_ulp(lwq.detach().cpu().numpy(), c["lwq"]) <= 2, max_ulp(lwq.detach().cpu().numpy(), c["lwq"])
    if name.endswith("nopred"):
        L = FusedPotentialLossNoPred(None, p=1, a=int(c["a_bits"]), w=int(c["w_bits"])).to(DEV)
        L.t, L.loss_sum, L.cnt = float(c["t"]), float(c["loss_sum"]), int(c["cnt"])
        base = torch.tensor(float(c["base"]), device=DEV, requires_grad=True) * 1.0
        ploss = L((base, las, laq, lws, lwq))
    else:
        L = FusedPotentialLoss(torch.nn.MSELoss(), p=1, a=int(c["a_bits"]), w=int(c["w_bits"])).to(DEV)
        L.t, L.loss_sum, L.cnt = float(c["t"]), float(c["loss_sum"]), int(c["cnt"])
        prd = torch.linspace(-1, 1, 12).view(3, 4).to(DEV).requires_grad_(True)
        tgt = (torch.linspace(1, -1, 12).view(3, 4) * 0.5).to(DEV)
        ploss = L((prd, las, laq, lws, lwq), tgt)
    assert abs(float(ploss) - float(c["ploss"])) <= 1e-6 * (abs(float(c["ploss"])) + float(c["base"]))
    # the quantized weights take part with a zero upstream gradient: the backward launches run with G = 0 and g_lwq live
    (ploss + sum((wq * 0.0).sum() for wq in wqs)).backward()
    for i, m in enumerate(convs):
        gw, ref = m.weight.grad.cpu().numpy(), c[f"gw{i}"]
        # gW here is ONLY the amin / amax share of dL/dlwq: +-t / count at the extremes with t = g_lwq / ((max - min + s) ln2),
        # zero elsewhere.  One term per element: 1e-6 relative to the term itself (the device's exp2 / log2 / division)
        assert np.array_equal(gw != 0, ref != 0), f"gw{i}: the extremes' positions"
        assert np.all(np.abs(gw - ref) <= 1e-6 * np.abs(ref) + 1e-30), f"gw{i}"
        gs, rs = m.log_wght_s.grad.cpu().numpy().reshape(-1), c[f"g_log_wght_s{i}"].reshape(-1)
        # d/dlog_s = (dL/dlws + t * s) * ...: two terms per channel; yardstick = their magnitudes
        assert np.all(np.abs(gs - rs) <= 1e-6 * (np.abs(rs) + 1.0)), (f"g_log_wght_s{i}", float(np.abs(gs - rs).max()))
    for i, a in enumerate(acts):
        for p, key in ((a.log_act_s, f"g_log_act_s{i}"), (a.log_act_q, f"g_log_act_q{i}")):
            ref = float(np.asarray(c[key]).reshape(-1)[0])
            assert abs(float(p.grad) - ref) <= 1e-6 * (abs(ref) + 1.0), key
