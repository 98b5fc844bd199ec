"""CPU: the plain-C oracle (oracle/fq_ref.c, independent of torch) against the vectors recorded from the real
reference (tests/golden/, oracle/gen_golden.py) and against the eager oracle: elementwise results bit for bit (STE /
LSQ / EWGS; AEWGS to the summation slack of its three group means, which the reference takes in fp32), reduced
parameter gradients to 1e-6 of their magnitude."""
import numpy as np
import pytest
import torch

from oracle import fq_c
from tests.golden_util import bit_equal, exact_off_extremes, load_cases, value_equal

ACT = {**load_cases("act_cases.npz"), **load_cases("ewgs_act_cases.npz")}          # (EWGS: oracle/gen_golden.py ewgs_enabled)
WGT = {**load_cases("weight_cases.npz"), **load_cases("ewgs_weight_cases.npz")}
NAMES = {0: "STE", 1: "EWGS", 2: "AEWGS", 3: "LSQ"}


def _exp2(v):
    return float(torch.exp2(torch.tensor(np.float32(v))))     # the exp2 the reference's layers call (gdnsq_act.py:42-43)


def close(a, b, rtol=2e-6, atol=1e-6):
    return np.allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", sorted(ACT))
def test_c_oracle_activation_cases(name):
    c = ACT[name]
    method = NAMES[int(c["method"])]
    s, qr, b = _exp2(c["log_act_s"]), _exp2(c["log_act_q"]), float(c["act_b"])
    out = fq_c.act(c["x"], c["g"], c["r"], s, qr, b, method)
    assert bit_equal(out["y"], c["y"])
    assert np.array_equal(out["q"], np.rint(out["q"]))          # integer-valued rounding indices (gdnsq.py:216)
    if method == "AEWGS":
        # delta = num / max(e2 - me^2, 1e-3) carries the fp32-vs-fp64 difference of its three group means (over dim 0:
        # four samples here), amplified where e2 - me^2 cancels; the bound propagates a 4e-6 relative slack of the means
        lo, hi = np.float32(b), np.float32(np.float32(b + np.float32(qr)) - np.float32(s))
        v = (np.minimum(np.maximum(c["x"], lo), hi) - np.float32(b)) / np.float32(s)
        e = np.rint(v) - v
        gq = c["g"] * np.float32(s)
        num, e2, me, ae = ((np.sign(gq) * e).mean(0, keepdims=True), (e * e).mean(0, keepdims=True),
                           e.mean(0, keepdims=True), np.abs(e).mean(0, keepdims=True))
        den = np.maximum(e2 - me * me, 1e-3)
        ddelta = 4e-6 * (ae / den + np.abs(num) * (e2 + 2 * np.abs(me) * ae) / den ** 2)
        tol = np.abs(c["g"]) * (np.abs(e) * ddelta + 1e-6) + 1e-9
        assert bool((np.abs(out["gx"] - c["gx"]) <= tol).all()), float((np.abs(out["gx"] - c["gx"]) / tol).max())
    else:
        assert value_equal(out["gx"], c["gx"])
    scale = float(np.abs(c["g"]).sum()) * max(1.0, qr / s)
    assert abs(out["g_log_act_s"] - float(c["g_log_act_s"][0])) <= 1e-6 * scale * s + 1e-7
    assert abs(out["g_log_act_q"] - float(c["g_log_act_q"][0])) <= 1e-6 * scale * s + 1e-7
    if c["signed"]:
        assert abs(out["g_act_b"] - float(c["g_act_b"][0])) <= 1e-6 * float(np.abs(c["g"]).sum()) + 1e-7


@pytest.mark.parametrize("name", sorted(n for n in WGT if "bias" not in WGT[n]))
def test_c_oracle_weight_cases(name):
    c = WGT[name]
    method = NAMES[int(c["method"])]
    pc = bool(c["per_channel"])
    # exp2 over the WHOLE parameter tensor, as the layer calls it (gdnsq_conv2d.py:72): torch's vectorised exp2 and its
    # scalar path may differ in the last bit, and the fixtures pin the bits of wq
    s = torch.exp2(torch.from_numpy(np.asarray(c["log_wght_s"], dtype=np.float32))).numpy().reshape(-1)
    out = fq_c.weight(c["w"], c["G"], c["r"], s, pc, method)
    assert bit_equal(out["wq"], c["wq"])
    assert bit_equal(out["zp"].reshape(np.asarray(c["zp"]).shape), c["zp"])
    w2 = c["w"].reshape(c["w"].shape[0], -1)
    Gabs = np.abs(c["G"]).reshape(w2.shape)
    if method == "AEWGS":
        tol = (1e-5 * Gabs + 1e-6 * (Gabs.sum(axis=1, keepdims=True) if pc else Gabs.sum()) + 1e-7).reshape(c["w"].shape)
        assert bool((np.abs(out["gw"] - c["gw"]) <= tol).all())
    else:
        assert exact_off_extremes(out["gw"], c["gw"], c["w"], pc)
        assert close(out["gw"], c["gw"], rtol=1e-5, atol=1e-6 * float(Gabs.sum()))      # the tie shares: reduced sums
    want = np.asarray(c["g_log_wght_s"], dtype=np.float64).reshape(-1)
    qmax = float(((w2.max() - w2.min()) / s.min()))
    yard = (Gabs.sum(axis=1) if pc else np.array([Gabs.sum()])) * s * (qmax + 1.0)
    assert bool((np.abs(out["g_log_wght_s"] - want) <= 1e-6 * yard + 1e-7).all())


def test_c_oracle_agrees_with_the_eager_oracle_on_random_tensors():
    """Beyond the fixtures: random shapes, all four estimators, both schemes, against oracle/fq_eager.py."""
    from oracle import fq_eager as O
    rng = np.random.default_rng(3)
    for trial in range(24):
        method = NAMES[trial % 4]
        pc = bool((trial // 4) % 2)
        co, ci, k = int(rng.integers(2, 9)), int(rng.integers(1, 6)), int(rng.choice([1, 3]))
        w = (rng.standard_normal((co, ci, k, k)) * 0.1).astype(np.float32)
        G = rng.standard_normal(w.shape).astype(np.float32)
        r = rng.choice(np.array([-1, 1], dtype=np.int8), size=w.shape)
        ls = (rng.uniform(-7, -3, size=(co, 1, 1, 1) if pc else (1,))).astype(np.float32)
        wt = torch.from_numpy(w).requires_grad_(True)
        lt = torch.from_numpy(ls).requires_grad_(True)
        wq, q, zp = O.weight_fake_quant(wt, lt, pc, method, r=torch.from_numpy(r.astype(np.float32) * 0.5))
        wq.backward(torch.from_numpy(G))
        s = torch.exp2(lt.detach()).numpy().reshape(-1)
        out = fq_c.weight(w, G, r, s, pc, method)
        assert bit_equal(out["wq"], wq.detach().numpy())
        if method == "AEWGS":
            assert np.allclose(out["gw"], wt.grad.numpy(), rtol=1e-4, atol=1e-5 * float(np.abs(G).max()))
        else:
            assert exact_off_extremes(out["gw"], wt.grad.numpy(), w, pc)
        assert np.allclose(out["g_log_wght_s"], lt.grad.numpy().reshape(-1).astype(np.float64), rtol=1e-4,
                           atol=1e-5 * float(np.abs(G).sum()))


MODEL = load_cases("model_cases.npz")


@pytest.mark.parametrize("name", sorted(MODEL))
def test_c_oracle_regulariser_inputs_and_potential_loss(name):
    """ModelHelper.get_model_values' weight half and the PotentialLoss value (gdnsq_loss.py:47-71, 129-153) against
    what the reference's own functions returned for the two-layer toy net."""
    c = MODEL[name]
    pc = bool(c["per_channel"])
    lwq = np.concatenate([fq_c.regulariser_input(c[f"w{i}"], c[f"log_wght_s{i}"], pc) for i in (0, 1)])
    assert np.allclose(lwq, np.asarray(c["lwq"]).reshape(-1), rtol=2e-7, atol=1e-6)      # libm log2f / exp2f: last bit
    if name.endswith("nopred"):
        base = float(c["base"])
    else:       # PotentialLoss with a prediction: the criterion of the recorded case (oracle/gen_golden.py) on its fixed pair
        prd = np.linspace(-1, 1, 12, dtype=np.float32)
        tgt = np.linspace(1, -1, 12, dtype=np.float32) * np.float32(0.5)
        base = float(np.mean((prd - tgt).astype(np.float64) ** 2))
    ploss, rloss = fq_c.potential_loss(base, c["las"], c["laq"], c["lws"], c["lwq"], float(c["a_bits"]),
                                       float(c["w_bits"]), float(c["t"]), float(c["loss_sum"]), float(c["cnt"]))
    assert abs(ploss - float(c["ploss"])) <= 2e-6 * abs(float(c["ploss"])) + 1e-7


def test_c_oracle_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """The CPU build under ASan + UBSan (GPU sanitizers are not available on the pool; the C restatement shares the
    indexing conventions -- [co][row] rows, dim-0 groups, ragged tails -- the kernels implement)."""
    import os
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fq_ref_selftest")
    subprocess.run(["gcc", "-O1", "-g", "-std=c99", "-ffp-contract=off", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=all", os.path.join(here, "oracle", "fq_ref.c"),
                    os.path.join(here, "oracle", "fq_ref_selftest.c"), "-o", exe, "-lm"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 problems" in out.stdout
