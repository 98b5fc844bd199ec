"""GPU (-m gpu): row extremes that land in ONE thread's share of a register-resident row.

The register-resident backward body (fq_pc.hip, pc_bwd_reg_body) stores every float4 of gW before the row reduction,
parks a thread's FIRST float4 that holds a row minimum / maximum in an 8-register slot and patches any further one of the
same thread through a read-back of its own store.  Thread t of a T-thread workgroup holds the float4 t, t + T, t + 2T, ...:
random weights never put two extremes there, so these rows are built by hand -- tied minima and maxima at float4 strides of
64 .. 1024 (every T the launch plans choose), a float4 that holds the minimum AND the maximum, a constant row -- and checked
(1) against the eager oracle on the same device, as tests/test_gpu_fused_layers.py does for random rows, and (2) grouped ==
per-layer bit for bit through the model-wide launches (mhaq_fq_wlayer_fwd_multi / mhaq_fq_wlayer_bwd_group)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fq_closed_form as CF  # noqa: E402
from oracle import fq_eager as O  # noqa: E402
from tests.aewgs_bound import aewgs_weight_slacks  # noqa: E402
from tests.golden_util import bit_equal  # noqa: E402

DEV = "cuda:0"
STRIDES = (64, 128, 256, 512, 1024)          # float4 strides: thread t also holds float4 t + T for the T in use


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from mhaq_amd import _lib, ops
    _lib.lib()
    return ops


def _plant(w2):
    """w2: [co, row] (row % 4 == 0, co >= 4), in place.  Returns the rows that were rebuilt."""
    co, row = w2.shape
    items = row // 4

    def at(f4, lane):                         # element index of lane `lane` of float4 `f4`
        return 4 * f4 + lane
    lo, hi = float(w2.min()) - 0.5, float(w2.max()) + 0.5
    # row 0: tied minima down one thread's share (and one more in a neighbouring thread)
    for k, sgap in enumerate((0,) + STRIDES):
        if 5 + sgap < items:
            w2[0, at(5 + sgap, k % 4)] = lo
    w2[0, at(6, 2)] = lo
    # row 1: tied maxima down one thread's share; the first of those float4 also holds the row minimum
    for k, sgap in enumerate((0,) + STRIDES):
        if 9 + sgap < items:
            w2[1, at(9 + sgap, (k + 1) % 4)] = hi
    w2[1, at(9, 0)] = lo
    # row 2: the minimum in float4 3, the maximum in float4 3 + T for every candidate T (all but one are plain elements
    # of other threads: then that thread's slot is taken by a maximum tie)
    w2[2, at(3, 1)] = lo
    for sgap in STRIDES:
        if 3 + sgap < items:
            w2[2, at(3 + sgap, 2)] = hi
    # row 3: constant -- max == min, every element tied, every float4 of every thread an extreme
    w2[3, :] = 0.125
    return [0, 1, 2, 3]


@pytest.mark.parametrize("method", ["STE", "LSQ", "EWGS", "AEWGS"])
# (1024, 8192): 33.5 MB -- past the 32 MB from which a layer's launch takes non-temporal stores: the read-back patch then reads a
# float4 the same thread has just written with a streaming store
@pytest.mark.parametrize("shape", [(8, 2048), (5, 128, 3, 3), (4, 512, 3, 3), (6, 8192), (4, 32768), (1024, 8192)])
def test_extremes_in_one_threads_share_match_the_eager_oracle(ops, method, shape):
    gen = torch.Generator().manual_seed(shape[0] * 7 + len(shape))
    fan = int(np.prod(shape[1:]))
    w = torch.randn(*shape, generator=gen) * math.sqrt(2.0 / fan)
    _plant(w.view(shape[0], -1))
    G = torch.randn(*shape, generator=gen)
    h = torch.randn(shape[0], generator=gen)            # upstream gradient of the regulariser inputs
    r = torch.randint(0, 2, shape, generator=gen).float() - 0.5
    dims = tuple(range(1, len(shape)))
    span = (w.amax(dims) - w.amin(dims)).clamp_min(0.05)
    ls0 = (torch.log2(span / 15.0) + 0.2 * torch.randn(shape[0], generator=gen)).reshape([shape[0]] + [1] * len(dims))
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
    # the bounds of tests/test_gpu_fused_layers.py::test_weight_layer_with_fused_regulariser: gW within 1e-6 of the sum of
    # the magnitudes that enter it (the tied extremes carry shares of REDUCED gradients), d/dlog_s within 1e-6 (AEWGS: + the
    # propagated slack of its group means)
    cf = CF.per_channel(w.cpu(), G.cpu(), r.cpu(), s.detach().cpu().reshape(-1), method)
    bc = [-1] + [1] * len(dims)
    abs_g = cf["abs_g"].reshape(bc).numpy() + np.abs(h.cpu().numpy()).reshape(bc) * 4
    got, ref = wg.grad.cpu().numpy(), wr.grad.cpu().numpy()
    err = np.abs(got - ref)
    assert np.all(err <= 1e-6 * (abs_g + np.abs(ref))), err.max()
    yard = (cf["abs_s"].numpy() + np.abs(h.cpu().numpy()) * 4) * math.log(2.0) * s.detach().cpu().numpy().reshape(-1) * 2
    ref_ls = lsr.grad.cpu().numpy().reshape(-1)
    errs = np.abs(lsg.grad.cpu().numpy().reshape(-1) - ref_ls)
    # (+ two ulps of the result: on the constant row every quantization error is 0, the yardstick above shrinks to the
    # regulariser's share times s, and d/dlog_s = h itself -- rounded once by each side)
    sl_ls = aewgs_weight_slacks(w, G, s.detach().reshape(-1), True)[1] if method == "AEWGS" else 0.0     # tests/aewgs_bound.py
    assert np.all(errs <= 1e-6 * yard + sl_ls + 2.4e-7 * np.abs(ref_ls) + 1e-9), (errs / yard).max()


@pytest.mark.parametrize("method", ["STE", "LSQ", "AEWGS"])
def test_grouped_launch_equals_per_layer_on_planted_extremes(ops, method):
    """The model-wide forward and ONE grouped backward over three layers whose rows carry the planted extremes:
    bit for bit what each layer's own fused op returns with the group's sign stream."""
    import mhaq_amd as M
    from mhaq_amd.multi import MultiTensorWeightQuant
    shapes = [(16, 128, 3, 3), (8, 256, 3, 3), (4, 512, 3, 3), (4, 64, 1, 1)]       # rows of 1152, 2304, 4608, 64 floats
    torch.manual_seed(11)
    net = torch.nn.ModuleList([M.NoisyConv2d(s[1], s[0], s[2], bias=False, qscheme=M.QScheme.PER_CHANNEL,
                                             log_s_init=-6, qnmethod=M.QNMethod[method]) for s in shapes]).to(DEV)
    with torch.no_grad():
        for m in net:
            w2 = m.weight.detach().cpu().reshape(m.weight.shape[0], -1).clone()
            _plant(w2)
            m.weight.copy_(w2.reshape(m.weight.shape).to(DEV))
            m.log_wght_s.add_(torch.randn_like(m.log_wght_s) * 0.3)
    Gs = [torch.randn(s, device=DEV) for s in shapes]
    hs = [torch.randn(s[0], device=DEV) for s in shapes]
    plan = MultiTensorWeightQuant(net, joint_backward=False, backward_group_elems=1 << 30)
    assert [(g.first, g.first + g.n) for g in plan.groups] == [(0, 4)]
    seed = 77
    ops.manual_seed(seed)
    plan.run()
    outs = []
    for m in net:
        wq, _, _ = m._quantized_weight()
        outs.append((wq, m.regulariser_input()))
    loss = sum((wq * G).sum() for (wq, _), G in zip(outs, Gs)) + sum((l * h).sum() for (_, l), h in zip(outs, hs))
    loss.backward()
    got = [(m.weight.grad.clone(), m.log_wght_s.grad.clone()) for m in net]
    g = plan.groups[0]

    def all_layers_match(offset):
        for i, m in enumerate(net):
            m.weight.grad = m.log_wght_s.grad = None
            n = m.weight.numel()
            e0 = plan.elem_off[i] - g.elem0
            r = ops.fill_r(g.elems, seed, offset, DEV)[e0:e0 + n]
            wq, zp, s, lwq = ops.fake_quant_weight_layer(m.weight, m.log_wght_s, method,
                                                         r_sign=None if method == "LSQ" else r)
            assert torch.equal(wq, outs[i][0]) and torch.equal(lwq, outs[i][1])
            ((wq * Gs[i]).sum() + (lwq * hs[i]).sum()).backward()
            if not (torch.equal(m.weight.grad, got[i][0]) and torch.equal(m.log_wght_s.grad, got[i][1])):
                return False
        return True
    assert any(all_layers_match(o) for o in (1, 2)), "grouped backward differs from the per-layer ops"
