"""PotentialLoss on the GPU (mhaq_fq_potential_loss_fwd/bwd through ops.potential_loss and the
FusedPotentialLoss modules) against the eager oracle (oracle/fq_eager.py:potential_loss, which follows
gdnsq_loss.py:47-71) and the reference-generated golden model cases.

Tolerance: the kernel sums the hinge terms in fp64, torch in fp32 -> |diff| <= 1e-6 relative on the loss
and on every gradient (north_star: <= 1e-6 fp32)."""
import numpy as np
import pytest
import torch

from tests.golden_util import load_cases

pytestmark = pytest.mark.gpu
MODEL = load_cases("model_cases.npz")
RTOL, ATOL = 1e-6, 1e-7


@pytest.fixture(scope="module")
def ops():
    from mhaq_amd import ops as _ops
    return _ops


def _inputs(na, nw, seed, ties=False):
    g = torch.Generator().manual_seed(seed)
    las = -5 + torch.randn(na, generator=g)
    laq = las + 4 + 1.5 * torch.randn(na, generator=g)      # hinge threshold 4 - 1e-3: roughly half active
    lws = -6 + torch.randn(nw, generator=g)
    lwq = lws + 4 + 1.5 * torch.randn(nw, generator=g)
    if ties:     # exact ties h == 0: torch.max splits the gradient there
        lws[::3] = -6.0
        lwq[::3] = (torch.tensor(-6.0) + (4 - torch.tensor(1e-3)))
    return las, laq, lws, lwq


@pytest.mark.parametrize("na,nw", [(2, 2), (21, 4800), (1500, 37)])
@pytest.mark.parametrize("p", [1, 2])
@pytest.mark.parametrize("lossless", [False, True])
def test_potential_loss_matches_oracle(ops, na, nw, p, lossless):
    from oracle import fq_eager as O
    dev = torch.device("cuda:0")
    las, laq, lws, lwq = _inputs(na, nw, seed=na + nw + p, ties=(nw > 100))
    t, loss_sum, cnt = 0.35, 3.3, 3
    ref_leaves = [v.clone().to(dev).requires_grad_(True) for v in (las, laq, lws, lwq)]
    ref_base = torch.tensor(1.7, device=dev, requires_grad=True)
    ref, _ = O.potential_loss(ref_base * 1.0, *ref_leaves, 4, 4, t, torch.tensor(loss_sum), cnt,
                              lossless=lossless, p=p)
    (ref * 1.25).backward()

    leaves = [v.clone().to(dev).requires_grad_(True) for v in (las, laq, lws, lwq)]
    base = torch.tensor(1.7, device=dev, requires_grad=True)
    state = torch.tensor([loss_sum, cnt, t], device=dev)
    out, stats = ops.potential_loss(base * 1.0, *leaves, state, 4, 4, p=p, lossless=lossless, update_state=True)
    (out * 1.25).backward()
    torch.testing.assert_close(out, ref.detach(), rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(base.grad, ref_base.grad, rtol=RTOL, atol=ATOL)
    for a, b in zip(leaves, ref_leaves):
        torch.testing.assert_close(a.grad, b.grad, rtol=RTOL, atol=ATOL)
    # state update and the logged statistics (gdnsq_loss.py:69-84)
    torch.testing.assert_close(state[0], torch.tensor(loss_sum, device=dev) + 1.7 ** p, rtol=RTOL, atol=ATOL)
    assert float(state[1]) == cnt + 1 and abs(float(state[2]) - t) < 1e-7
    d = lwq - lws
    expect = {7: -lws.mean(), 8: lwq.mean(), 9: -las.mean(), 10: laq.mean(), 11: d.max()}
    for i, v in expect.items():
        torch.testing.assert_close(stats[i].cpu(), v, rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("name", sorted(MODEL))
def test_fused_modules_match_reference_golden(name):
    """The reference's own PotentialLoss / PotentialLossNoPred outputs (tests/golden/model_cases.npz)."""
    from mhaq_amd.loss import FusedPotentialLoss, FusedPotentialLossNoPred
    c = MODEL[name]
    dev = torch.device("cuda:0")
    vecs = [torch.from_numpy(np.asarray(c[k])).to(dev).requires_grad_(True) for k in ("las", "laq", "lws", "lwq")]
    if name.endswith("nopred"):
        L = FusedPotentialLossNoPred(None, p=1, a=int(c["a_bits"]), w=int(c["w_bits"])).to(dev)
        L.t, L.loss_sum, L.cnt = float(c["t"]), float(c["loss_sum"]), int(c["cnt"])
        ploss = L((torch.tensor(float(c["base"]), device=dev) * 1.0, *vecs))
    else:
        L = FusedPotentialLoss(torch.nn.MSELoss(), p=1, a=int(c["a_bits"]), w=int(c["w_bits"])).to(dev)
        L.t, L.loss_sum, L.cnt = float(c["t"]), float(c["loss_sum"]), int(c["cnt"])
        prd = torch.linspace(-1, 1, 12).view(3, 4).to(dev)
        tgt = (torch.linspace(1, -1, 12).view(3, 4) * 0.5).to(dev)
        ploss = L((prd, *vecs), tgt)
    assert np.allclose(ploss.item(), c["ploss"], rtol=RTOL, atol=ATOL)
    assert L.cnt == int(c["cnt"]) + 1            # training mode advanced the state
    ploss.backward()
    if not bool(c["per_channel"]):               # per-tensor: las/laq stack the layers' scalars one-to-one
        for i in range(2):
            assert np.allclose(vecs[0].grad[i].item(), c[f"g_log_act_s{i}"], rtol=RTOL, atol=ATOL)
            assert np.allclose(vecs[1].grad[i].item(), c[f"g_log_act_q{i}"], rtol=RTOL, atol=ATOL)


def test_eval_mode_leaves_state_alone_and_rejects_cpu():
    from mhaq_amd.loss import FusedPotentialLossNoPred
    dev = torch.device("cuda:0")
    las, laq, lws, lwq = (v.to(dev) for v in _inputs(4, 9, seed=1))
    L = FusedPotentialLossNoPred(None, p=1, a=4, w=4).to(dev).eval()
    L.loss_sum, L.cnt, L.t = 2.0, 5, 0.1
    L((torch.tensor(0.5, device=dev), las, laq, lws, lwq))
    assert L.cnt == 5 and float(L.loss_sum) == 2.0
    with pytest.raises((ValueError, RuntimeError)):
        L((torch.tensor(0.5), las.cpu(), laq.cpu(), lws.cpu(), lwq.cpu()))


def test_gradients_follow_the_input_shapes(ops):
    dev = torch.device("cuda:0")
    las, laq, lws, lwq = (v.to(dev).reshape(-1, 1).requires_grad_(True) for v in _inputs(6, 10, seed=3))
    base = torch.tensor([0.8], device=dev, requires_grad=True)
    out, _ = ops.potential_loss(base, las, laq, lws, lwq, torch.tensor([1.0, 2.0, 0.5], device=dev), 4, 4)
    out.backward()
    assert base.grad.shape == (1,) and las.grad.shape == (6, 1) and lwq.grad.shape == (10, 1)
