"""CPU: the oracle (oracle/fq_eager.py) against the golden vectors recorded from the real
reference (tests/golden/, oracle/gen_golden.py).  Elementwise results must be bit-identical;
reduced parameter gradients must agree to 1e-6 relative (they are the same op sequence, so in
practice they are identical too)."""
import math

import numpy as np
import pytest
import torch

from oracle import fq_eager as O
from tests.golden_util import T, bit_equal, load_cases, r_from_sign, value_equal

# + the EWGS cases: the reference's own QNEWGS.backward lines run with the misspelled attribute of gdnsq.py:102 supplied
# (oracle/gen_golden.py `ewgs_enabled`; the shipped reference raises there)
ACT = {**load_cases("act_cases.npz"), **load_cases("ewgs_act_cases.npz")}
WGT = {**load_cases("weight_cases.npz"), **load_cases("ewgs_weight_cases.npz")}
MODEL = load_cases("model_cases.npz")


def close(a, b, rtol=1e-6, atol=1e-7):
    return np.allclose(np.asarray(a, np.float32), np.asarray(b, np.float32), rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", sorted(ACT))
def test_act_oracle_matches_reference(name):
    c = ACT[name]
    x = T(c["x"]).requires_grad_(True)
    ls = T(c["log_act_s"]).reshape(1).requires_grad_(True)
    lq = T(c["log_act_q"]).reshape(1).requires_grad_(True)
    b = T(c["act_b"]).reshape(1).requires_grad_(bool(c["signed"]))
    method = O.METHODS[int(c["method"])]         # NoisyAct(qnmethod=...): STE, LSQ, AEWGS and EWGS cases (gdnsq_act.py:17)
    y, q = O.act_fake_quant(x, ls, lq, b, r=r_from_sign(c["r"]), method=method)
    y.backward(T(c["g"]))
    assert bit_equal(y.detach().numpy(), c["y"])
    assert value_equal(x.grad.numpy(), c["gx"])
    assert close(ls.grad, c["g_log_act_s"])
    assert close(lq.grad, c["g_log_act_q"])
    if c["signed"]:
        assert close(b.grad, c["g_act_b"])
    # eval-mode behaviour
    s, qr = torch.exp2(ls.detach()), torch.exp2(lq.detach())
    bd = b.detach()
    if c["eval_raises"]:
        with pytest.raises(AssertionError):
            O.check_integrity(q.detach(), s, bd, bd, bd + qr - s)
    else:
        O.check_integrity(q.detach(), s, bd, bd, bd + qr - s)
        assert bit_equal(O.act_bit_width(q.detach()).numpy(), c["bw"])
        assert bit_equal(y.detach().numpy(), c["y_eval"])
    # integer-valued rounding indices (gdnsq.py:216)
    qn = q.detach().numpy()
    assert np.array_equal(qn, np.rint(qn))


@pytest.mark.parametrize("name", sorted(WGT))
def test_weight_oracle_matches_reference(name):
    c = WGT[name]
    w = T(c["w"]).requires_grad_(True)
    ls = T(c["log_wght_s"]).requires_grad_(True)
    pc = bool(c["per_channel"])
    method = O.METHODS[int(c["method"])]
    wq, q, zp = O.weight_fake_quant(w, ls, pc, method, r=r_from_sign(c["r"]))
    outs, grads = [wq], [T(c["G"])]
    if "bias" in c:
        bias = T(c["bias"]).requires_grad_(True)
        bq = O.bias_fake_quant(bias, w, ls, method, r=r_from_sign(c["rb"]))
        outs.append(bq)
        grads.append(T(c["Gb"]))
    torch.autograd.backward(outs, grads)
    assert bit_equal(wq.detach().numpy(), c["wq"])
    assert bit_equal(zp.detach().numpy(), c["zp"])
    assert close(w.grad, c["gw"])
    assert close(ls.grad, c["g_log_wght_s"], rtol=2e-6, atol=1e-6)
    if "bias" in c:
        assert bit_equal(bq.detach().numpy(), c["bq"])
        assert close(bias.grad, c["gbias"])


def test_ewgs_reference_raises_oracle_implements_intended():
    # gdnsq.py:102 `ctx.need_input_grad` typo: the reference raises; the oracle restates the intent -- pinned by the
    # ewgs_* fixtures above, which the reference's own lines produced once the misspelled attribute existed.
    assert sum(n.startswith(("ewgs_", "tied_ewgs", "wide_ewgs", "qbias_ewgs", "linear_ewgs")) for n in list(ACT) + list(WGT)) == 13
    w = torch.randn(4, 2, 3, 3, requires_grad=True)
    ls = torch.full((4, 1, 1, 1), -3.0, requires_grad=True)
    wq, _, _ = O.weight_fake_quant(w, ls, True, "EWGS", r=torch.full_like(w, 0.5))
    wq.backward(torch.ones_like(w))
    assert torch.isfinite(w.grad).all() and torch.isfinite(ls.grad).all()


def test_unknown_method_raises_attribute_error():
    with pytest.raises(AttributeError):
        O.quantize(torch.ones(3), torch.ones(1), 0.0, -math.inf, math.inf, "NOPE")


@pytest.mark.parametrize("name", sorted(MODEL))
def test_regulariser_and_potential_loss(name):
    c = MODEL[name]
    pc = bool(c["per_channel"])
    ws = [T(c[f"w{i}"]).requires_grad_(True) for i in range(2)]
    lss = [T(c[f"log_wght_s{i}"]).requires_grad_(True) for i in range(2)]
    las_l = [T(c[f"log_act_s{i}"]).requires_grad_(True) for i in range(2)]
    laq_l = [T(c[f"log_act_q{i}"]).requires_grad_(True) for i in range(2)]
    lws, lwq = O.regulariser_inputs(ws, lss, pc)
    las = torch.cat(las_l) if pc else torch.stack(las_l).ravel()
    laq = torch.cat(laq_l) if pc else torch.stack(laq_l).ravel()
    assert bit_equal(lws.detach().numpy(), c["lws"])
    assert bit_equal(lwq.detach().numpy(), c["lwq"])
    if name.endswith("nopred"):
        base = torch.tensor(float(c["base"])) * 1.0
    else:
        prd = torch.linspace(-1, 1, 12).view(3, 4)
        tgt = torch.linspace(1, -1, 12).view(3, 4) * 0.5
        base = torch.nn.functional.mse_loss(prd, tgt)
    ploss, _ = O.potential_loss(base, las, laq, lws, lwq, int(c["a_bits"]), int(c["w_bits"]),
                                float(c["t"]), torch.tensor(float(c["loss_sum"])), int(c["cnt"]))
    assert close(ploss.detach(), c["ploss"])
    ploss.backward()
    for i in range(2):
        assert close(ws[i].grad, c[f"gw{i}"])
        assert close(lss[i].grad, c[f"g_log_wght_s{i}"])
        assert close(las_l[i].grad, c[f"g_log_act_s{i}"])
        assert close(laq_l[i].grad, c[f"g_log_act_q{i}"])


# ------------------------------------------------------------------ closed forms vs eager oracle
from oracle import fq_closed_form as CF  # noqa: E402


@pytest.mark.parametrize("method", ["STE", "LSQ", "EWGS"])
@pytest.mark.parametrize("case", ["clip", "inverted", "wide"])
def test_closed_form_per_tensor_matches_eager(method, case):
    g0 = torch.Generator().manual_seed(hash((method, case)) % 1000)
    x = torch.randn(3, 5, 7, 9, generator=g0) * 3
    g = torch.randn(3, 5, 7, 9, generator=g0)
    r = torch.randint(0, 2, x.shape, generator=g0).float() - 0.5
    s0, zp0, lo0, hi0 = {"clip": (0.37, -1.3, -1.3, 2.1), "inverted": (2.0, -0.5, -0.5, -1.5),
                         "wide": (2.0 ** -8, -20.0, -20.0, 20.0)}[case]
    xs = x.clone().requires_grad_(True)
    P = [torch.tensor([v], requires_grad=True) for v in (s0, zp0, lo0, hi0)]
    y = O.dequantize(O.quantize(xs, P[0], P[1], P[2], P[3], method, r), P[0], P[1])
    y.backward(g)
    cf = CF.per_tensor(x, g, r, s0, zp0, lo0, hi0, method)
    assert bit_equal(cf["y"].numpy(), y.detach().numpy())
    assert value_equal(cf["gx"].numpy(), xs.grad.numpy())
    for key, p, yard in (("g_s", P[0], "abs_s"), ("g_zp", P[1], "abs_g"), ("g_lo", P[2], "abs_g"),
                         ("g_hi", P[3], "abs_g")):
        assert abs(float(cf[key]) - float(p.grad)) <= 1e-6 * float(cf[yard]) + 1e-30, key


@pytest.mark.parametrize("method", ["STE", "LSQ", "EWGS", "AEWGS"])
def test_closed_form_per_channel_matches_eager(method):
    g0 = torch.Generator().manual_seed(7)
    w = torch.randn(6, 4, 3, 3, generator=g0) * 0.2
    w[2].flatten()[[3, 9]] = w[2].min() - 0.01  # tied minimum
    G = torch.randn(6, 4, 3, 3, generator=g0)
    r = torch.randint(0, 2, w.shape, generator=g0).float() - 0.5
    ls = (torch.randn(6, 1, 1, 1, generator=g0) * 0.3 - 4.0)
    ws = w.clone().requires_grad_(True)
    s = torch.exp2(ls).requires_grad_(True)
    zp = O.weight_zero_point(ws, True)
    wq = O.dequantize(O.quantize(ws, s, zp, -math.inf, math.inf, method, r), s, zp)
    wq.backward(G)
    cf = CF.per_channel(w, G, r, s.detach().reshape(6), method)
    assert bit_equal(cf["wq"].numpy(), wq.detach().numpy())
    assert np.allclose(cf["gw"].numpy(), ws.grad.numpy(), rtol=1e-6, atol=1e-6 * float(cf["abs_g"].max()))
    err = (cf["g_s"] - s.grad.reshape(6).double()).abs()
    assert bool((err <= 1e-6 * cf["abs_s"] + 1e-30).all())
