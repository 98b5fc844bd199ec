"""GPU (-m gpu): the host-side mirror of the reference interface (Quantizer, NoisyAct, NoisyConv2d,
NoisyLinear, the QN* Functions, the wrapping rule, the QAT step) against the CPU oracle layers."""
import copy
import math
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fq_eager as O  # noqa: E402
from oracle.ref_layers import ORACLE_LAYERS  # noqa: E402
from oracle import ref_layers as RL  # noqa: E402
from oracle import fq_closed_form as CF  # noqa: E402
from tests.aewgs_bound import aewgs_gx_bound, aewgs_slack, aewgs_weight_slacks, within  # noqa: E402
from tests.golden_util import bit_equal, exact_off_extremes, value_equal  # noqa: E402
from tests.teacher_forced import Recorder  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def M():
    assert torch.cuda.is_available()
    import mhaq_amd
    from mhaq_amd import _lib
    _lib.lib()
    return mhaq_amd


def reduced_close(got, ref, yard, rel=1e-6, slack=0.0):
    """|got - ref| <= rel * sum|terms| (+ slack) elementwise (the bar of a REDUCED gradient: DESIGN.md section 2).  An AEWGS
    caller passes as `slack` what its elementwise terms inherit from the last-bit difference of three group means (fp64
    here, fp32 in torch) amplified by delta = num / max(e2 - me^2, 1e-3): tests/aewgs_bound.py::aewgs_slack --
    tests/test_gpu_aewgs_apply_exact.py pins the arithmetic behind the means bit for bit."""
    got, ref, yard = (torch.as_tensor(t).detach().double().cpu().reshape(-1) for t in (got, ref, yard))
    slack = torch.as_tensor(slack).detach().double().cpu().reshape(-1)
    return bool(((got - ref).abs() <= rel * yard + slack + 1e-30).all())


def close(a, b, rtol=2e-5, atol=1e-6):
    return np.allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=rtol, atol=atol)


def sync_scale_params(dst, src):
    """Copy parameters CPU module -> GPU module (same names)."""
    src_params = dict(src.named_parameters())
    with torch.no_grad():
        for n, p in dst.named_parameters():
            p.copy_(src_params[n].reshape(p.shape))
    assert set(src_params) == {n for n, _ in dst.named_parameters()}


# ------------------------------------------------------------------ NoisyAct
@pytest.mark.parametrize("init_q", [2.0, 1.5])
@pytest.mark.parametrize("signed", [True, False])
def test_noisy_act_train_and_eval(M, signed, init_q):
    torch.manual_seed(3)
    ref = RL.NoisyAct(init_s=-3, init_q=init_q, signed=signed, qnmethod="LSQ")   # LSQ: no random draw
    act = M.NoisyAct(init_s=-3, init_q=init_q, signed=signed, qnmethod=M.QNMethod.LSQ).to(DEV)
    x = torch.relu(torch.randn(4, 8, 10, 10)) if not signed else torch.randn(4, 8, 10, 10) * 2
    g = torch.randn_like(x)
    xr, xg = x.clone().requires_grad_(True), x.clone().to(DEV).requires_grad_(True)
    yr = ref(xr); yr.backward(g)
    yg = act(xg); yg.backward(g.to(DEV))
    # log_s = -3 and log_q = 1.5: exp2 of a half-integer may differ by 1 ulp between host and device,
    # so compare with the device's own scale bits
    s = torch.exp2(act.log_act_s.detach()).cpu(); qr = torch.exp2(act.log_act_q.detach()).cpu()
    b = act.act_b.detach().cpu()
    y_exact = O.dequantize(O.quantize(x, s, b, b, b + qr - s, "LSQ"), s, b)
    assert bit_equal(yg.detach().cpu().numpy(), y_exact.numpy())
    same_quantizer = torch.equal(qr, torch.exp2(ref.log_act_q.detach()))      # host and device exp2 agree (always for 2.0)
    assert same_quantizer or init_q != 2.0
    cf = CF.per_tensor(x, g, torch.zeros_like(x), s, b, b, b + qr - s, "LSQ")
    ln2 = math.log(2.0)
    yard_s = (float(cf["abs_s"]) + float(cf["abs_g"])) * ln2 * float(s)
    yard_q = float(cf["abs_g"]) * ln2 * float(qr)
    if same_quantizer:
        assert bit_equal(yg.detach().cpu().numpy(), yr.detach().numpy())
        assert value_equal(xg.grad.cpu().numpy(), xr.grad.numpy())
        assert reduced_close(act.log_act_s.grad, ref.log_act_s.grad, yard_s)
        assert reduced_close(act.log_act_q.grad, ref.log_act_q.grad, yard_q)
        if signed:
            assert reduced_close(act.act_b.grad, ref.act_b.grad, float(cf["abs_g"]))
    else:
        # 2^1.5 differs by one ulp between the host's and the device's exp2: the CPU layer is then a DIFFERENT quantizer
        # (its upper bound sits one ulp away), so only closeness can be asked of it; the exact comparison above (y_exact)
        # and the closed form below use the device's own scale bits
        assert close(yg, yr) and close(xg.grad, xr.grad)
        assert close(act.log_act_s.grad, ref.log_act_s.grad, rtol=1e-4, atol=1e-4)
        assert close(act.log_act_q.grad, ref.log_act_q.grad, rtol=1e-4, atol=1e-4)
    # against the fp64 closed form with the device's scale bits, both parametrisations
    g_s, g_hi = float(cf["g_s"]), float(cf["g_hi"])
    assert reduced_close(act.log_act_s.grad, (g_s - g_hi) * float(s) * ln2, yard_s)
    assert reduced_close(act.log_act_q.grad, g_hi * float(qr) * ln2, yard_q)
    if signed:
        assert reduced_close(act.act_b.grad, float(cf["g_zp"]) + float(cf["g_lo"]) + g_hi, float(cf["abs_g"]))
    else:
        assert act.act_b.grad is None
    # eval: bit width + lazily checked integrity flags
    ref.eval(); act.eval()
    with torch.no_grad():
        ye_r, ye_g = ref(x), act(x.to(DEV))
    assert close(ye_g, ye_r)
    assert abs(float(act.bw) - float(ref.bw)) < 1e-5
    act.Q.check_integrity()     # must not raise


def test_eval_integrity_flags_raise_like_reference(M):
    act = M.NoisyAct().to(DEV).eval()
    x = torch.randn(16, device=DEV)
    x[3] = float("nan")
    with torch.no_grad():
        act(x)
    with pytest.raises(AssertionError):
        act.Q.check_integrity()


# ------------------------------------------------------------------ NoisyConv2d / NoisyLinear
def _spy_wq_grad(layer, store):
    """Record dL/dWq as the layer's own backward sees it (the convolution / matmul between the quantizer and the
    loss runs on different back ends for the CPU oracle and the GPU layer: its last bits are not the
    quantizer's business)."""
    inner = layer._conv_forward if hasattr(layer, "_conv_forward") else None

    def spy(inp, weight, bias):
        weight.register_hook(lambda g: store.__setitem__("G", g.detach().clone()))
        return inner(inp, weight, bias)
    layer._conv_forward = spy


def _assert_weight_grads(layer, ref_w, ref_ls, G_cpu, per_channel, method, r=None, rel=1e-6):
    """layer.weight.grad / layer.log_wght_s.grad against the oracle driven by the SAME upstream gradient:
    elementwise part of gW exact (STE/LSQ/EWGS), reduced parts within rel * sum|terms|."""
    w = ref_w.detach().clone().requires_grad_(True)
    ls = ref_ls.detach().clone().requires_grad_(True)
    wq_r = O.weight_fake_quant(w, ls, per_channel, method, r=r)[0]
    wq_r.backward(G_cpu)
    co = w.shape[0] if per_channel else 1
    s = torch.exp2(ls.detach()).reshape(co)
    cf = CF.per_channel(w.detach().reshape(co, -1), G_cpu.reshape(co, -1), None if r is None else r.reshape(co, -1), s,
                        "STE" if (method == "AEWGS" and not per_channel) else method)
    gw = layer.weight.grad.detach().cpu().numpy()
    if method != "AEWGS":
        assert exact_off_extremes(gw, w.grad.numpy(), w.detach().numpy(), per_channel)
    abs_g = cf["abs_g"].numpy().reshape([co] + [1] * (w.dim() - 1)) if per_channel else float(cf["abs_g"])
    sl_gw, sl_ls = 0.0, 0.0
    if method == "AEWGS":
        # the propagated slack of the three group means (tests/aewgs_bound.py::aewgs_weight_slacks), instead of the blanket
        # 4e-6 of rounds 2-5
        sl_gw, sl_ls = aewgs_weight_slacks(w, G_cpu, s, per_channel)
        sl_ls = sl_ls.reshape(ls.shape)
    assert np.all(np.abs(gw - w.grad.numpy()) <= rel * (abs_g + np.abs(w.grad.numpy())) + sl_gw)
    yard = (cf["abs_s"].numpy() * math.log(2.0) * s.numpy() * 2).reshape(ls.shape)
    err = np.abs(layer.log_wght_s.grad.detach().cpu().numpy().reshape(ls.shape) - ls.grad.numpy())
    assert np.all(err <= rel * yard + sl_ls + 1e-30), (float(err.max()), float(yard.max()))


@pytest.mark.parametrize("qscheme", [0, 1])
@pytest.mark.parametrize("method", ["LSQ", "AEWGS"])
def test_noisy_conv2d_matches_oracle_layer(M, qscheme, method):
    torch.manual_seed(11)
    ref = RL.NoisyConv2d(6, 8, 3, padding=1, qscheme=qscheme, log_s_init=-5, qnmethod=method)
    conv = M.NoisyConv2d(6, 8, 3, padding=1, qscheme=M.QScheme(qscheme), log_s_init=-5,
                         qnmethod=M.QNMethod[method]).to(DEV)
    sync_scale_params(conv, ref)
    x = torch.randn(2, 6, 9, 9)
    store, r = {}, None
    if method == "AEWGS":
        r = torch.randint(0, 2, ref.weight.shape).float() - 0.5
        wq_r = O.weight_fake_quant(ref.weight, ref.log_wght_s, bool(qscheme), method, r=r)[0]
        out_r = torch.nn.functional.conv2d(x, wq_r, ref.bias, padding=1)
        from mhaq_amd import ops
        s = torch.exp2(conv.log_wght_s)
        fn = ops.fake_quant_weight_pc if qscheme else ops.fake_quant_weight_pt
        wq_g, _ = fn(conv.weight, s, method, r_sign=(r * 2).to(torch.int8).to(DEV))
        wq_g.register_hook(lambda g: store.__setitem__("G", g.detach().clone()))
        out_g = torch.nn.functional.conv2d(x.to(DEV), wq_g, conv.bias, padding=1)
    else:
        _spy_wq_grad(conv, store)
        out_r, out_g = ref(x), conv(x.to(DEV))
    go = torch.randn_like(out_r)
    out_r.backward(go); out_g.backward(go.to(DEV))
    assert close(out_g, out_r, rtol=1e-4, atol=1e-5)             # through two different convolution back ends
    # the quantizer itself, driven by the upstream gradient the GPU layer actually received.  AEWGS: fp64 group
    # means here vs torch's fp32 (DESIGN.md "known deviations"), amplified by delta = num / max(e2 - me^2, 1e-3)
    _assert_weight_grads(conv, ref.weight, ref.log_wght_s, store["G"].cpu(), bool(qscheme), method, r)
    if method != "AEWGS":   # the layer's own forward ran: side consumers read Q.zero_point / Q.scale
        assert conv.Q.zero_point.shape == ((8, 1, 1, 1) if qscheme else ())
        assert conv.Q.scale.shape == conv.log_wght_s.shape
    if qscheme:
        assert conv.log_b_s.grad is None      # unused unless quant_bias (needs find_unused_parameters)


def test_noisy_conv2d_quant_bias(M):
    torch.manual_seed(12)
    ref = RL.NoisyConv2d(4, 6, 3, qscheme=1, log_s_init=-4, quant_bias=True, qnmethod="LSQ")
    conv = M.NoisyConv2d(4, 6, 3, qscheme=M.QScheme.PER_CHANNEL, log_s_init=-4, quant_bias=True,
                         qnmethod=M.QNMethod.LSQ).to(DEV)
    sync_scale_params(conv, ref)
    x = torch.randn(2, 4, 8, 8)
    store = {}
    inner = conv._conv_forward

    def spy(inp, weight, bias):
        weight.register_hook(lambda g: store.__setitem__("G", g.detach().clone()))
        bias.register_hook(lambda g: store.__setitem__("Gb", g.detach().clone()))
        return inner(inp, weight, bias)
    conv._conv_forward = spy
    out_r, out_g = ref(x), conv(x.to(DEV))
    go = torch.randn_like(out_r)
    out_r.backward(go); out_g.backward(go.to(DEV))
    assert close(out_g, out_r, rtol=1e-4, atol=1e-5)
    # oracle weight + bias quantizers driven by the GPU layer's own upstream gradients
    w = ref.weight.detach().clone().requires_grad_(True)
    ls = ref.log_wght_s.detach().clone().requires_grad_(True)
    bb = ref.bias.detach().clone().requires_grad_(True)
    wq_r = O.weight_fake_quant(w, ls, True, "LSQ")[0]
    bq_r = O.bias_fake_quant(bb, w, ls, "LSQ")
    torch.autograd.backward([wq_r, bq_r], [store["G"].cpu(), store["Gb"].cpu()])
    G, Gb = store["G"].cpu(), store["Gb"].cpu()
    s = torch.exp2(ls.detach()).reshape(-1)
    cf = CF.per_channel(w.detach(), G, None, s, "LSQ")
    assert np.allclose(conv.bias.grad.cpu().numpy(), bb.grad.numpy(), rtol=1e-6, atol=1e-7)
    gw = conv.weight.grad.cpu().numpy()
    assert exact_off_extremes(gw, w.grad.numpy(), w.detach().numpy(), True)
    abs_g = (cf["abs_g"] + 2 * Gb.abs()).numpy().reshape(-1, 1, 1, 1)
    assert np.all(np.abs(gw - w.grad.numpy()) <= 1e-6 * (abs_g + np.abs(w.grad.numpy())))
    qb = ((bb.detach() - w.detach().amin((1, 2, 3))) / s).abs() + 1          # the bias's rounding index magnitude
    yard = ((cf["abs_s"] + 4 * Gb.abs() * qb) * math.log(2.0) * s * 2).numpy().reshape(ls.shape)
    err = np.abs(conv.log_wght_s.grad.cpu().numpy() - ls.grad.numpy())
    assert np.all(err <= 1e-6 * yard + 1e-30), (float(err.max()), float(yard.max()))


@pytest.mark.parametrize("qscheme", [0, 1])
def test_noisy_linear(M, qscheme):
    torch.manual_seed(13)
    ref = RL.NoisyLinear(32, 10, qscheme=qscheme, log_s_init=-5, qnmethod="LSQ")
    lin = M.NoisyLinear(32, 10, qscheme=M.QScheme(qscheme), log_s_init=-5, qnmethod=M.QNMethod.LSQ).to(DEV)
    sync_scale_params(lin, ref)
    x = torch.randn(7, 32)
    out_r, out_g = ref(x), lin(x.to(DEV))
    out_r.sum().backward(); out_g.sum().backward()
    assert close(out_g, out_r, rtol=1e-4, atol=1e-5)
    # dL/dWq of sum(x @ Wq^T + b) is the column sums of x broadcast over the rows: exact inputs on both sides
    # would need the same matmul; drive the oracle with the analytic upstream gradient instead
    G = x.sum(0, keepdim=True).expand(10, 32).contiguous()
    ls_ref = ref.log_wght_s if not qscheme else ref.log_wght_s.reshape(10, 1)
    w = ref.weight.detach().clone().requires_grad_(True)
    ls = ls_ref.detach().clone().requires_grad_(True)
    O.weight_fake_quant(w, ls, bool(qscheme), "LSQ")[0].backward(G)
    co = 10 if qscheme else 1
    s = torch.exp2(ls.detach()).reshape(co)
    cf = CF.per_channel(w.detach().reshape(co, -1), G.reshape(co, -1), None, s, "LSQ")
    # the GPU layer's G comes out of a GEMM (x^T summed in another order): rel 1e-6 of the same yardsticks
    gw = lin.weight.grad.cpu().numpy()
    abs_g = cf["abs_g"].numpy().reshape(co, 1) if qscheme else float(cf["abs_g"])
    assert np.all(np.abs(gw - w.grad.numpy()) <= 2e-6 * (abs_g + np.abs(w.grad.numpy())))
    yard = (cf["abs_s"].numpy() * math.log(2.0) * s.numpy() * 2).reshape(ls.shape)
    err = np.abs(lin.log_wght_s.grad.cpu().numpy().reshape(ls.shape) - ls.grad.numpy())
    assert np.all(err <= 2e-6 * yard + 1e-30), (float(err.max()), float(yard.max()))


# ------------------------------------------------------------------ Quantizer facade + QN* Functions
@pytest.mark.parametrize("method", ["STE", "LSQ", "EWGS", "AEWGS"])
@pytest.mark.parametrize("kind", ["per_tensor", "per_channel"])
def test_quantizer_facade_matches_oracle(M, method, kind):
    from mhaq_amd import ops_generic as G
    torch.manual_seed(21)
    x = torch.randn(6, 4, 3, 3) * 0.5
    g = torch.randn_like(x)
    r = torch.randint(0, 2, x.shape).float() - 0.5
    if kind == "per_tensor":
        s0, zp0, lo0, hi0 = torch.tensor([0.07]), torch.tensor([-0.9]), torch.tensor([-0.9]), torch.tensor([0.8])
    else:
        s0 = (torch.rand(6, 1, 1, 1) * 0.05 + 0.02)
        zp0 = x.amin((1, 2, 3), keepdim=True)
        lo0, hi0 = -math.inf, math.inf
    # oracle
    xr = x.clone().requires_grad_(True)
    sr = s0.clone().requires_grad_(True)
    zr = zp0.clone().requires_grad_(True)
    qr_ = O.quantize(xr, sr, zr, lo0, hi0, method, r)
    yr = O.dequantize(qr_, sr, zr)
    yr.backward(g)
    # facade on the GPU
    mod = torch.nn.Identity().train()
    xg = x.clone().to(DEV).requires_grad_(True)
    sg = s0.clone().to(DEV).requires_grad_(True)
    zg = zp0.clone().to(DEV).requires_grad_(True)
    lo_g = lo0.to(DEV) if torch.is_tensor(lo0) else lo0
    hi_g = hi0.to(DEV) if torch.is_tensor(hi0) else hi0
    Q = M.Quantizer(mod, sg, zg, lo_g, hi_g, qnmethod=M.QNMethod[method])
    cls = G._BY_METHOD[M.QNMethod[method]]
    cls.r_sign = (r * 2).to(torch.int8).to(DEV)
    try:
        q = Q.quantize(xg)
        y = Q.dequantize(q)
        y.backward(g.to(DEV))
    finally:
        cls.r_sign = None
    assert bit_equal(q.detach().cpu().numpy(), qr_.detach().numpy())
    assert bit_equal(y.detach().cpu().numpy(), yr.detach().numpy())
    if method == "AEWGS":
        # group means: fp64 here, fp32 in torch -- the propagated slack of the three means (tests/aewgs_bound.py); the
        # [1]-shaped scale takes its means over dim 0 (reduce_to_shape), the [C,1,1,1] one over dims 1..3
        vq = ((torch.clamp(x, lo0, hi0) if kind == "per_tensor" else x) - zp0) / s0
        assert within(xg.grad, xr.grad, aewgs_gx_bound(vq, g, (0,) if kind == "per_tensor" else (1, 2, 3)))
    else:
        assert value_equal(xg.grad.cpu().numpy(), xr.grad.numpy())
    # reduced gradients: 1e-6 * sum|terms|; AEWGS + what the sums inherit from the group means' last bits (aewgs_slack:
    # d/ds sums (gv / s) * v, d/dzp sums gv / s -- over the tensor for the [1]-shaped scale, per channel for [C,1,1,1])
    sl_s = sl_z = 0.0
    if method == "AEWGS":
        sdims = None if kind == "per_tensor" else (1, 2, 3)
        sl_s = aewgs_slack(vq, g, (0,) if kind == "per_tensor" else (1, 2, 3), weight=vq, sum_dims=sdims)
        sl_z = aewgs_slack(vq, g, (0,) if kind == "per_tensor" else (1, 2, 3), sum_dims=sdims)
    if kind == "per_tensor":
        cf = CF.per_tensor(x, g, r, s0, zp0, lo0, hi0, "STE" if method == "AEWGS" else method)
        assert reduced_close(sg.grad, sr.grad, float(cf["abs_s"]), slack=sl_s)
        assert reduced_close(zg.grad, zr.grad, float(cf["abs_g"]), slack=sl_z)
    else:
        dims = (1, 2, 3)
        v = (x - zp0) / s0
        n = torch.round(v) - v
        gq = g * s0
        abs_s = ((g * (v + n)).abs().double().sum(dims) + (gq * (v / s0)).abs().double().sum(dims) * 2
                 + (gq * 0.5774).abs().double().sum(dims))
        abs_g = g.abs().double().sum(dims) * 3
        assert reduced_close(sg.grad, sr.grad, abs_s, slack=sl_s)
        assert reduced_close(zg.grad, zr.grad, abs_g, slack=sl_z)


def test_qnoise_base_class_raises_in_backward(M):
    from mhaq_amd import ops_generic as G
    v = torch.randn(8, device=DEV, requires_grad=True)
    s = torch.ones(1, device=DEV, requires_grad=True)
    n = G.QNoise.apply(v, s)
    assert torch.equal(n, torch.round(v.detach()) - v.detach())
    with pytest.raises(AttributeError):
        n.sum().backward()
    Q = M.Quantizer(torch.nn.Identity(), s, 0.0, -math.inf, math.inf, qnmethod="bogus")
    with pytest.raises(AttributeError):
        Q.quantize(v)


# ------------------------------------------------------------------ model level
def test_resnet20_qat_step_matches_oracle_model(M):
    """Whole wrapped ResNet-20 (18 quantized convs, per-tensor, LSQ = no random term): loss and every
    gradient of one step, HIP layers against the oracle's eager layers.  The oracle model runs on
    the same device so both use the same MIOpen convolutions: quantization is discontinuous, and a
    CPU-vs-GPU convolution difference in the last bits flips rounding decisions layer after layer
    (measured: eager-on-CPU vs eager-on-GPU gradients agree only to cos 0.96 at a 2^-7 grid, while
    HIP vs eager-on-GPU agree to cos 1 - 1e-13 with a bit-identical loss)."""
    from mhaq_amd import nets, wrap
    torch.manual_seed(5)
    base = nets.resnet20_cifar(10)
    ref = copy.deepcopy(base)
    gpu = copy.deepcopy(base).to(DEV)
    excl = ("features.init_block.conv", "output")
    wrap.quantize_model(ref, 0, "LSQ", excl, layers=ORACLE_LAYERS)
    ref.to(DEV)
    wrap.quantize_model(gpu, 0, "LSQ", excl)
    assert sum(1 for m in gpu.modules() if isinstance(m, M.NoisyAct)) == 18
    assert all(m.signed for m in gpu.modules() if isinstance(m, M.NoisyAct))      # Appendix A
    assert sorted(n for n, _ in gpu.named_parameters()) == sorted(n for n, _ in ref.named_parameters())
    with torch.no_grad():
        for net in (ref, gpu):
            for m in net.modules():
                if hasattr(m, "log_act_s"):
                    m.log_act_s.fill_(-4.3); m.log_act_q.fill_(3.1); m.act_b.fill_(-3.7)
                    # the wrapping rule builds NoisyAct with the default STE estimator, whose scale
                    # gradient is a random draw; LSQ makes the comparison deterministic
                    if hasattr(m, "Q"):
                        m.Q.qnmethod = M.QNMethod.LSQ
                    else:
                        m.qnmethod = "LSQ"
                if hasattr(m, "log_wght_s"):
                    m.log_wght_s.fill_(-7.4)
    x = torch.randn(8, 3, 32, 32, device=DEV)
    yl = torch.randint(0, 10, (8,), device=DEV)
    ref.train(); gpu.train()
    for a, b in zip(wrap.get_model_values(gpu, 0), wrap.get_model_values(ref, 0)):
        assert torch.equal(a, b)
    loss_r = torch.nn.functional.cross_entropy(ref(x), yl)
    rec = Recorder(gpu)
    loss_g = torch.nn.functional.cross_entropy(gpu(x), yl)
    loss_r.backward(); loss_g.backward()
    rec.close()
    # every quantizer of the model against its closed form on the tensors it actually saw: elementwise parts
    # exact, reduced gradients within 1e-6 * sum|terms| (the op-level bar, inside the model)
    assert rec.check(rel=1e-6) == 36
    assert abs(float(loss_r.detach()) - float(loss_g.detach())) <= 1e-6 * abs(float(loss_r.detach()))
    ref_params = dict(ref.named_parameters())
    worst = (2.0, None)
    for n, pg in gpu.named_parameters():
        pr = ref_params[n]
        if pr.grad is None:
            assert pg.grad is None, n
            continue
        a, b = pg.grad.flatten().double(), pr.grad.flatten().double()
        err = float((a - b).abs().max())
        # scalar quantizer parameters are sums of cancelling terms (e.g. act_b = sum g - sum g1 + ...):
        # the reference's own fp32 summation noise is ~1e-7 of sum|g|, hence the absolute floor
        if a.numel() == 1:
            continue        # scalar quantizer parameters: pinned at 1e-6 * sum|terms| by rec.check() above
        assert err <= 1e-4 * float(b.abs().max()) + 1e-6 * float(b.abs().sum()) + 1e-6, (n, err)
        if float(b.norm()) > 1e-9 and a.numel() >= 16:
            worst = min(worst, (float(torch.dot(a, b) / (a.norm() * b.norm())), n))
    assert worst[0] > 1 - 1e-8, worst


# ------------------------------------------------------------------ bit-width statistics (8f rank 3)
def test_bit_width_statistics_match_reference_definition(M):
    from mhaq_amd import stats, wrap
    from mhaq_amd.gdnsq import check_model_integrity
    torch.manual_seed(8)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(),
                              torch.nn.Conv2d(8, 6, 3, padding=1)).to(DEV)
    for qs in (0, 1):
        import copy
        qnet = copy.deepcopy(net)
        wrap.quantize_model(qnet, qs, "STE", ())
        with torch.no_grad():
            for m in qnet.modules():
                if hasattr(m, "log_wght_s"):
                    m.log_wght_s.fill_(-4.0)
                if hasattr(m, "log_act_s"):
                    m.log_act_s.fill_(-3.0); m.log_act_q.fill_(2.0); m.act_b.fill_(-2.0 if m.signed else 0.0)
        qnet.eval()
        with torch.no_grad():
            qnet(torch.randn(4, 3, 10, 10, device=DEV))
        check_model_integrity(qnet)                      # no flag set -> no AssertionError
        # literal restatement of model_stats.get_true_layer_bit_width with the oracle quantizer
        expect = []
        for m in qnet.modules():
            if isinstance(m, M.NoisyConv2d):
                w = m.weight.detach().cpu()
                s = torch.exp2(m.log_wght_s.detach().cpu())
                zp = O.weight_zero_point(w, bool(qs))
                q = O.quantize(w, s, zp, -math.inf, math.inf, "STE", r=torch.zeros_like(w))
                if qs:
                    expect.append(max(float(np.log2(float(c.max() - c.min() + 1))) for c in q.reshape(q.shape[0], -1)))
                else:
                    expect.append(float(np.log2(float(q.max() - q.min() + 1))))
                assert abs(stats.get_true_layer_bit_width(m) - expect[-1]) < 1e-6
        assert abs(stats.get_true_weights_width(qnet) - max(expect)) < 1e-6
        assert abs(stats.get_true_weights_width(qnet, max=False) - float(np.mean(expect))) < 1e-6
        acts = [m for m in qnet.modules() if isinstance(m, M.NoisyAct)]
        assert abs(stats.get_true_activations_width(qnet) - max(float(a.bw) for a in acts)) < 1e-6
        assert abs(float(stats.get_activations_bit_width_mean(qnet)) - 5.0) < 1e-6      # log_q - log_s
        assert torch.isfinite(stats.get_weights_bit_width_mean(qnet))
        crit = types.SimpleNamespace(wt=32, at=32)
        assert stats.is_converged(qnet, crit) and not stats.is_converged(qnet, types.SimpleNamespace(wt=1, at=1))


@pytest.mark.parametrize("qscheme", [0, 1])
def test_bit_widths_from_group_extremes_equal_the_literal_definition(M, qscheme):
    """stats._layer_bit_widths takes max(q) - min(q) + 1 from the group's min and max (q is monotone in w); it must
    equal the reference's literal definition -- quantize every weight, reduce q -- exactly, for odd scales, tied
    extremes, constant channels and channels_last storage."""
    from mhaq_amd import stats
    gen = torch.Generator().manual_seed(17 + qscheme)
    for shape, kind in (((16, 8, 3, 3), "conv"), ((64, 64, 3, 3), "conv"), ((50, 50, 3, 3), "conv"), ((7, 5, 1, 1), "conv"),
                        ((10, 64), "lin")):
        if kind == "conv":
            m = M.NoisyConv2d(shape[1], shape[0], shape[2], qscheme=M.QScheme(qscheme)).to(DEV)
        else:
            m = M.NoisyLinear(shape[1], shape[0], qscheme=M.QScheme(qscheme)).to(DEV)
        with torch.no_grad():
            m.weight.copy_(torch.randn(*shape, generator=gen) * 0.3)
            m.weight[0].fill_(0.125)                                      # a constant channel: one level
            m.weight.view(shape[0], -1)[1, :2] = m.weight[1].max() + 0.01    # tied maxima
            m.log_wght_s.copy_((torch.rand(m.log_wght_s.shape, generator=gen) * 6 - 8).to(DEV))   # 2^-8 .. 2^-2, not powers of two
        for fmt in ((torch.contiguous_format, torch.channels_last) if kind == "conv" else (torch.contiguous_format,)):
            m.weight.data = m.weight.data.contiguous(memory_format=fmt)
            q = stats._weight_indices(m)                                  # the literal q of Quantizer.quantize
            if qscheme:
                flat = q.reshape(q.shape[0], -1)
                want = torch.log2(flat.amax(1) - flat.amin(1) + 1)
            else:
                want = torch.log2(q.max() - q.min() + 1).reshape(1)
            got = stats._layer_bit_widths(m)
            assert torch.equal(got, want), (shape, qscheme, fmt)
            assert float(flat.amin(1).abs().max() if qscheme else q.min().abs()) == 0.0   # min(q) is q(zp) = 0


# ------------------------------------------------------------------ RFDN (config 5) and ResNet-18 wrap rule
def test_rfdn_lsq_step_and_wrap_rule(M):
    """config/gdnsq_config_rfdn_lsq_w2a2.yaml: per-channel LSQ, convs with bias, L1 loss, no distillation.
    33 wrapped convs, all signed; one training step HIP vs oracle layers on the same device."""
    from mhaq_amd import nets, wrap
    torch.manual_seed(21)
    base = nets.rfdn()
    ref, gpu = copy.deepcopy(base), copy.deepcopy(base).to(DEV)
    excl = ("fea_conv", "upsampler.0")
    wrap.quantize_model(ref, 1, "LSQ", excl, layers=ORACLE_LAYERS)
    ref.to(DEV)
    wrap.quantize_model(gpu, 1, "LSQ", excl)
    acts = [m for m in gpu.modules() if isinstance(m, M.NoisyAct)]
    assert len(acts) == 33 and all(a.signed for a in acts)
    with torch.no_grad():
        for net in (ref, gpu):
            for m in net.modules():
                if hasattr(m, "log_act_s"):
                    m.log_act_s.fill_(-6.1); m.log_act_q.fill_(4.2); m.act_b.fill_(-9.0)
                    if hasattr(m, "Q"):
                        m.Q.qnmethod = M.QNMethod.LSQ
                    else:
                        m.qnmethod = "LSQ"
                if hasattr(m, "log_wght_s"):
                    m.log_wght_s.fill_(-9.4)
    x = torch.rand(2, 3, 24, 24, device=DEV)
    hr = torch.rand(2, 3, 96, 96, device=DEV)
    losses = []
    rec = Recorder(gpu)
    for net in (ref, gpu):
        net.train()
        out = net(x)
        vals = wrap.get_model_values(net, 1)
        for v in vals[:3]:
            v.retain_grad()                   # the regulariser's direct reading of the log-parameters
        loss = torch.nn.functional.l1_loss(out, hr) + 0.01 * (vals[3] - vals[2]).relu().mean()
        loss.backward()
        losses.append(float(loss.detach()))
    rec.close()
    # 33 activation + 33 weight quantizers, incl. the regulariser's share through lwq
    assert rec.check(rel=1e-6, direct=Recorder.direct_grads(gpu, vals)) == 66
    assert abs(losses[0] - losses[1]) <= 1e-6 * abs(losses[0])
    rp = dict(ref.named_parameters())
    for n, pg in gpu.named_parameters():
        pr = rp[n]
        if pr.grad is None:
            assert pg.grad is None, n
            continue
        a, b = pg.grad.flatten().double(), pr.grad.flatten().double()
        err = float((a - b).abs().max())
        if a.numel() == 1:
            continue        # scalar quantizer parameters: pinned at 1e-6 * sum|terms| by rec.check() above
        tol = 1e-4 * float(b.abs().max()) + 1e-6 * float(b.abs().sum()) + 1e-6
        assert err <= tol, (n, err, tol)


def test_resnet18_wrap_rule_matches_appendix_a(M):
    from mhaq_amd import nets, wrap
    net = nets.resnet18(10).to(DEV)
    wrap.quantize_model(net, 1, "AEWGS", ("conv1", "fc"))
    acts = [(n, m.signed) for n, m in net.named_modules() if isinstance(m, M.NoisyAct)]
    assert len(acts) == 16
    for n, signed in acts:
        assert signed == (".conv1." in n), n        # conv2 follows the block's ReLU -> unsigned
    assert isinstance(net.layer2[0].downsample[0], torch.nn.Conv2d) and not isinstance(net.layer2[0].downsample[0], M.NoisyConv2d)
    with pytest.raises(AttributeError):
        wrap.quantize_model(nets.resnet18(10), 1, "STE", ("nope",))
    with pytest.raises(AttributeError):             # a non-excluded nn.Linear has no kernel_size (reference behaviour)
        wrap.quantize_model(nets.resnet18(10), 1, "STE", ("conv1",))
    keys = set(net.state_dict())
    assert "layer1.0.conv1.activations_quantizer.log_act_s" in keys and "layer1.0.conv1.0.log_wght_s" in keys


@pytest.mark.parametrize("method", ["STE", "LSQ", "AEWGS"])
def test_quantizer_facade_per_element_scale(M, method):
    """Quantizer with one scale / zero point per element: how the reference quantizes the bias
    (Q_b.scale = s.ravel(), Q_b.zero_point = min.ravel(), gdnsq_conv2d.py:87-94) through the facade."""
    from mhaq_amd import ops_generic as G
    torch.manual_seed(31)
    n = 24
    x = torch.randn(n) * 0.3
    g = torch.randn(n)
    r = torch.randint(0, 2, (n,)).float() - 0.5
    s0 = torch.rand(n) * 0.05 + 0.02
    zp0 = -torch.rand(n)
    xr, sr, zr = (t.clone().requires_grad_(True) for t in (x, s0, zp0))
    yr = O.dequantize(O.quantize(xr, sr, zr, -math.inf, math.inf, method, r), sr, zr)
    yr.backward(g)
    xg, sg, zg = (t.clone().to(DEV).requires_grad_(True) for t in (x, s0, zp0))
    Q = M.Quantizer(torch.nn.Identity().train(), sg, zg, -math.inf, math.inf, qnmethod=M.QNMethod[method])
    cls = G._BY_METHOD[M.QNMethod[method]]
    cls.r_sign = (r * 2).to(torch.int8).to(DEV)
    try:
        y = Q.dequantize(Q.quantize(xg))
        y.backward(g.to(DEV))
    finally:
        cls.r_sign = None
    assert bit_equal(y.detach().cpu().numpy(), yr.detach().numpy())
    if method == "AEWGS":
        # no unit dimension in the scale's shape: reduce_to_shape averages over everything -- one group of 24
        assert within(xg.grad, xr.grad, aewgs_gx_bound((x - zp0) / s0, g, (0,)))
    else:
        assert value_equal(xg.grad.cpu().numpy(), xr.grad.numpy())
    # per-element parameters: every "reduction" has exactly one term g*q - gv*(v/s) + noise, resp. g - gv/s
    v = (x - zp0) / s0
    gq = g * s0
    yard_s = (g * torch.round(v)).abs() + (gq * (v / s0)).abs() * 2 + gq.abs()
    sl_s = sl_z = 0.0
    if method == "AEWGS":            # one term per "sum": the element's own propagated slack, times |v| for d/ds
        sl_z = aewgs_gx_bound(v, g, (0,))
        sl_s = sl_z * v.abs().double()
    assert reduced_close(sg.grad, sr.grad, yard_s, slack=sl_s)
    assert reduced_close(zg.grad, zr.grad, g.abs() * 3, slack=sl_z)


def test_noisy_act_aewgs_estimator_path(M):
    """NoisyAct(qnmethod=AEWGS) -- legal in the reference's API though no config builds it -- takes the
    unfused parameter chain + per-tensor op with per-position statistics; y and dL/dx are deterministic."""
    torch.manual_seed(41)
    ref = RL.NoisyAct(init_s=-3, init_q=2, signed=True, qnmethod="AEWGS").to(DEV)
    act = M.NoisyAct(init_s=-3, init_q=2, signed=True, qnmethod=M.QNMethod.AEWGS).to(DEV)
    x = torch.randn(6, 4, 5, 5, device=DEV) * 2
    g = torch.randn_like(x)
    xr, xg = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    yr = ref(xr); yr.backward(g)
    yg = act(xg); yg.backward(g)
    assert torch.equal(yg, yr)
    s_ = torch.exp2(act.log_act_s.detach()).cpu()
    qr_ = torch.exp2(act.log_act_q.detach()).cpu()
    b_ = act.act_b.detach().cpu()
    xc, gc = x.cpu(), g.cpu()
    vq = (torch.clamp(xc, b_, b_ + qr_ - s_) - b_) / s_
    bound = aewgs_gx_bound(vq, gc, (0,))                     # per-position statistics: means over the batch dim
    assert within(xg.grad, xr.grad, bound)
    # d/dlog_act_q = (sum of gx over x > hi) * qr * ln2 -- no random term in it; its terms are elements of gx, so their
    # propagated slack sums up next to the reduction's own 1e-6 of sum|terms|
    over = (xc > b_ + qr_ - s_)
    yard = float((gc.abs() * over).double().sum()) * float(qr_) * math.log(2.0)
    slack = float((bound * over).sum()) * float(qr_) * math.log(2.0)
    assert abs(float(act.log_act_q.grad) - float(ref.log_act_q.grad)) <= 1e-6 * yard + slack + 1e-30
    assert act.log_act_s.grad is not None and torch.isfinite(act.log_act_s.grad).all()


def test_quantizer_with_non_positive_scale_skips_rounding(M):
    # gdnsq.py:186,201-202,226-227: a Quantizer constructed with scale <= 0 only clamps and shifts
    x = torch.randn(32, device=DEV)
    Q = M.Quantizer(torch.nn.Identity(), torch.tensor([0.0], device=DEV), torch.tensor([0.5], device=DEV),
                    torch.tensor([-1.0], device=DEV), torch.tensor([1.0], device=DEV))
    assert Q.positive_scale is False
    want = torch.clamp(x, -1.0, 1.0) - 0.5
    assert torch.equal(Q.quantize(x), want)
    assert torch.equal(Q.dequantize(want), want + 0.5)
    assert torch.equal(Q.fake_quant(x), (want) + 0.5)


def test_noisy_linear_inside_a_model_trains(M):
    net = torch.nn.Sequential(torch.nn.Flatten(), M.NoisyLinear(48, 16, qscheme=M.QScheme.PER_CHANNEL,
                                                                 log_s_init=-6, qnmethod=M.QNMethod.STE),
                              torch.nn.ReLU(), M.NoisyLinear(16, 4, log_s_init=-6)).to(DEV)
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    x = torch.randn(64, 3, 4, 4, device=DEV)
    y = torch.randint(0, 4, (64,), device=DEV)
    first = last = None
    for _ in range(30):
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(net(x), y)
        loss.backward()
        opt.step()
        first = first if first is not None else float(loss.detach())
        last = float(loss.detach())
    assert last < first


def test_facade_backward_serves_more_than_65535_scale_groups():
    """mhaq_fq_noise_bwd (the QN* facade of gdnsq.py:35-147) with one scale per output channel of a [70001, 3] tensor -- a
    wide Linear through Quantizer.quantize / dequantize -- and with one scale per ELEMENT of a 70001-element vector: the
    groups ride grid.x (they rode grid.y, which ends at 65535 workgroups, until round 5)."""
    import mhaq_amd as M
    from mhaq_amd import ops_generic as G
    torch.manual_seed(3)
    co, row = 70001, 3
    for shape, sshape in (((co, row), (co, 1)), ((co,), (co,))):
        v = (torch.randn(shape, device=DEV) * 3).requires_grad_(True)
        s = (torch.rand(sshape, device=DEV) + 0.5).requires_grad_(True)
        g = torch.randn(shape, device=DEV)
        n = G.QNLSQ.apply(v, s)
        n.backward(g)
        e = torch.round(v.detach()) - v.detach()
        assert torch.equal(n.detach(), e)
        assert torch.equal(v.grad, torch.zeros_like(v))                      # grad_output * 0 (gdnsq.py:75-77)
        ref = (g * e).double().reshape(co, -1).sum(1).float().reshape(sshape)
        yard = (g * e).double().abs().reshape(co, -1).sum(1).float().reshape(sshape)
        assert torch.all((s.grad - ref).abs() <= 1e-6 * yard + 1e-30)
        assert M.QNMethod.LSQ.value == 3


# ------------------------------------------------------------------ round 6: the facade on the fused kernels
def test_facade_pair_is_one_fused_node_and_equals_the_fused_op(M):
    """Q.dequantize(Q.quantize(x)) on per-tensor parameters is the fused op's launch pair: q and y from ONE forward launch,
    the fused backward kernel behind them -- the same bits as Quantizer.fake_quant, value and every gradient."""
    from mhaq_amd import ops, ops_generic as G
    torch.manual_seed(5)
    x = (torch.randn(4, 8, 6, 6) * 0.7).to(DEV)
    g = torch.randn_like(x)
    r = torch.randint(0, 2, x.shape, device=DEV).to(torch.int8) * 2 - 1
    outs = []
    for pair in (True, False):
        xg = x.clone().requires_grad_(True)
        s = torch.tensor([0.11], device=DEV, requires_grad=True)
        zp = torch.tensor([-0.8], device=DEV, requires_grad=True)
        lo = torch.tensor([-0.8], device=DEV, requires_grad=True)
        hi = torch.tensor([0.9], device=DEV, requires_grad=True)
        Q = M.Quantizer(torch.nn.Identity().train(), s, zp, lo, hi, qnmethod=M.QNMethod.STE)
        if pair:
            G.QNSTE.r_sign = r
            try:
                q = Q.quantize(xg)
                assert type(q.grad_fn).__name__.startswith("QuantizePairPT")
                y = Q.dequantize(q)
                assert y.grad_fn is q.grad_fn                      # the same node: nothing was recomputed for y
                y.backward(g)
            finally:
                G.QNSTE.r_sign = None
            assert torch.equal(q.detach(), torch.round(q.detach()))
        else:
            y = Q.fake_quant(xg, r_sign=r)
            y.backward(g)
        outs.append((y.detach(), xg.grad, s.grad, zp.grad, lo.grad, hi.grad))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_facade_q_consumed_directly_still_gets_the_reference_gradients(M):
    """A q that something other than dequantize differentiates takes the chain (the reference's own graph): LSQ, no random
    term, against the eager oracle."""
    torch.manual_seed(6)
    x = torch.randn(3, 5, 4) * 0.6
    g = torch.randn_like(x)
    s0, zp0, lo0, hi0 = torch.tensor([0.09]), torch.tensor([-0.7]), torch.tensor([-0.7]), torch.tensor([0.6])
    xr, sr, zr = x.clone().requires_grad_(True), s0.clone().requires_grad_(True), zp0.clone().requires_grad_(True)
    qr = O.quantize(xr, sr, zr, lo0, hi0, "LSQ", None)
    (qr * g).sum().backward()
    xg, sg, zg = (t.clone().to(DEV).requires_grad_(True) for t in (x, s0, zp0))
    Q = M.Quantizer(torch.nn.Identity().train(), sg, zg, lo0.to(DEV), hi0.to(DEV), qnmethod=M.QNMethod.LSQ)
    q = Q.quantize(xg)
    (q * g.to(DEV)).sum().backward()
    assert bit_equal(q.detach().cpu().numpy(), qr.detach().numpy())
    assert value_equal(xg.grad.cpu().numpy(), xr.grad.numpy())
    yard = float((g.abs() * (x.abs() + 1) / s0 ** 2).double().sum())
    assert abs(float(sg.grad) - float(sr.grad)) <= 1e-6 * yard
    assert abs(float(zg.grad) - float(zr.grad)) <= 1e-6 * float((g.abs() / s0).double().sum())


def test_eval_mode_facade_quantize_never_synchronises(M):
    """gdnsq.py:211-217 raises from three host syncs per call; here an eval-mode Q.quantize() -- per-tensor, per-channel
    and per-element parameters -- leaves a device flag word in Q.last_flags and issues NO synchronisation."""
    torch.manual_seed(8)
    mod = torch.nn.Identity().eval()
    x = (torch.randn(6, 4, 3, 3) * 0.5).to(DEV)
    Qs = [M.Quantizer(mod, torch.tensor([0.07], device=DEV), torch.tensor([-0.9], device=DEV),
                      torch.tensor([-0.9], device=DEV), torch.tensor([0.8], device=DEV)),
          M.Quantizer(mod, (torch.rand(6, 1, 1, 1) * 0.05 + 0.02).to(DEV), x.amin((1, 2, 3), keepdim=True),
                      -math.inf, math.inf),
          M.Quantizer(mod, (torch.rand(6, 4, 3, 3) * 0.05 + 0.02).to(DEV), torch.full_like(x, -2.0), -math.inf, math.inf)]
    with torch.no_grad():
        for Q in Qs:
            Q.quantize(x)                     # warm-up: library load, allocator
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            got = []
            for Q in Qs:
                q = Q.quantize(x)
                got.append((q, Q.dequantize(q), Q.last_flags))
        finally:
            torch.cuda.set_sync_debug_mode("default")
    for (q, y, flags), Q in zip(got, Qs):
        assert flags is not None and int(flags.item()) == 0
        Q.check_integrity()
        v = (torch.clamp(x, Q.min_val, Q.max_val) - Q.zero_point) / Q.scale
        want = v + (torch.round(v) - v)
        assert torch.equal(q, want) and torch.equal(y, want * Q.scale + Q.zero_point)
    # ... and the flag word carries the reference's asserts: a NaN input is "not an integer" on every route
    bad = x.clone()
    bad[1, 2, 0, 1] = float("nan")
    with torch.no_grad():
        for Q in Qs:
            Q.quantize(bad)
            assert int(Q.last_flags.item()) & 4
            with pytest.raises(AssertionError, match="integer values"):
                Q.check_integrity()


def test_facade_per_channel_quantize_uses_the_zero_point_the_quantizer_holds(M):
    """model_stats.py:118,123 quantizes the detached weights with the zero point the layer's LAST forward left in Q -- not
    with the row minimum of the weights as they are now (which the fused per-channel forward would take)."""
    torch.manual_seed(9)
    w = (torch.randn(8, 4, 3, 3) * 0.3).to(DEV)
    s = (torch.rand(8, 1, 1, 1) * 0.05 + 0.02).to(DEV)
    stale_zp = w.amin((1, 2, 3), keepdim=True) - 3 * s              # the optimizer has moved the weights since
    Q = M.Quantizer(torch.nn.Identity().train(), s, stale_zp, -math.inf, math.inf)
    with torch.no_grad():
        q = Q.quantize(w)
    v = (w - stale_zp) / s
    assert torch.equal(q, v + (torch.round(v) - v))
    assert torch.equal(q.amin((1, 2, 3)), torch.full((8,), 3.0, device=DEV))     # the fresh row minimum would give 0


# ------------------------------------------------------------------ the reference's layers, only `Quantizer` swapped
class _ReferenceStyleAct(torch.nn.Module):
    """The forward of gdnsq_act.py:39-55 as the reference writes it -- parameters, the scalar chain, the attribute
    assignments, quantize, the eval-mode bit width, dequantize -- around an injected Quantizer class (test code: what a
    maintainer's layer looks like after `from mhaq_amd import Quantizer`)."""

    def __init__(self, Q_cls, init_s, init_q, signed, qnmethod):
        super().__init__()
        zp = 0.0 if not signed else -2.0 ** (init_q - 1)
        self.log_act_s = torch.nn.Parameter(torch.tensor([float(init_s)]))
        self.log_act_q = torch.nn.Parameter(torch.tensor([float(init_q)]))
        self.act_b = torch.nn.Parameter(torch.tensor([zp]), requires_grad=bool(signed))
        self.Q = Q_cls(self, torch.exp2(self.log_act_s.detach()), 0, -math.inf, math.inf, qnmethod=qnmethod)
        self.bw = torch.tensor(0.0)

    def forward(self, x):
        s, q = torch.exp2(self.log_act_s), torch.exp2(self.log_act_q)
        self.Q.zero_point = self.act_b
        self.Q.min_val = self.act_b
        self.Q.max_val = self.act_b + q - s
        self.Q.scale = s
        qv = self.Q.quantize(x)
        if not self.training:
            mn, mx = qv.aminmax()
            self.bw = torch.log2(mx - mn + 1)
        return self.Q.dequantize(qv)


@pytest.mark.parametrize("signed", [True, False])
def test_reference_style_layer_with_only_the_quantizer_swapped(M, signed):
    """a layer that keeps the reference's own forward and imports `Quantizer` from mhaq_amd runs
    the fused launch pair (one node for quantize + dequantize) and gives the product layer's outputs and gradients -- LSQ,
    no random term: y and dL/dx bit for bit, the three parameter gradients through torch's scalar chain within
    1e-6 * sum|terms| of the fused layer's in-kernel chain and of the eager oracle; eval: same y, same bit width, flag word."""
    torch.manual_seed(17)
    x = (torch.randn(5, 6, 7, 7) * 2).to(DEV)
    g = torch.randn_like(x)
    ref_style = _ReferenceStyleAct(M.Quantizer, -3, 2, signed, M.QNMethod.LSQ).to(DEV)
    product = M.NoisyAct(init_s=-3, init_q=2, signed=signed, qnmethod=M.QNMethod.LSQ).to(DEV)
    oracle = RL.NoisyAct(init_s=-3, init_q=2, signed=signed, qnmethod="LSQ").to(DEV)
    xs = [x.clone().requires_grad_(True) for _ in range(3)]
    ys = [m(xi) for m, xi in zip((ref_style, product, oracle), xs)]
    assert type(ys[0].grad_fn).__name__.startswith("QuantizePairPT")            # the pair is one fused node
    for y, xi in zip(ys, xs):
        y.backward(g)
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])
    assert torch.equal(xs[0].grad, xs[1].grad) and torch.equal(xs[0].grad, xs[2].grad)
    s, qr, b = 2.0 ** -3, 2.0 ** 2, float(product.act_b)
    cf = CF.per_tensor(x.cpu(), g.cpu(), None, torch.tensor([s]), torch.tensor([b]), torch.tensor([b]),
                       torch.tensor([b + qr - s]), "LSQ")
    yard_s = (float(cf["abs_s"]) + float(cf["abs_g"])) * s * math.log(2.0)
    yard_q = float(cf["abs_g"]) * qr * math.log(2.0)
    for other in (product, oracle):
        assert abs(float(ref_style.log_act_s.grad) - float(other.log_act_s.grad)) <= 1e-6 * yard_s
        assert abs(float(ref_style.log_act_q.grad) - float(other.log_act_q.grad)) <= 1e-6 * yard_q + 1e-30
        if signed:
            assert abs(float(ref_style.act_b.grad) - float(other.act_b.grad)) <= 1e-6 * float(cf["abs_g"])
    ref_style.eval(); product.eval(); oracle.eval()
    with torch.no_grad():
        ye = [m(x) for m in (ref_style, product, oracle)]
    assert torch.equal(ye[0], ye[1]) and torch.equal(ye[0], ye[2])
    assert float(ref_style.bw) == float(product.bw) == float(oracle.bw)
    assert int(ref_style.Q.last_flags.item()) == 0
    ref_style.Q.check_integrity()
