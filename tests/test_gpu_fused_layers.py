"""GPU (-m gpu): the layer-level fused entry points (mhaq_fq_act_fwd/bwd, mhaq_fq_wlayer_fwd/bwd) that
take the LEARNABLE parameters and fold the scalar chain (exp2, clamp bounds, regulariser input) into
the kernels.  Checker: the eager oracle executed on the same device (so exp2/log2 are the same device
functions torch's own eager path uses) with explicit random signs."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fq_closed_form as CF  # noqa: E402
from oracle import fq_eager as O  # noqa: E402
from tests.aewgs_bound import aewgs_weight_slacks  # noqa: E402
from tests.golden_util import bit_equal, exact_off_extremes, value_equal  # noqa: E402
from tests.teacher_forced import Recorder  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from mhaq_amd import _lib, ops
    _lib.lib()
    return ops


def P(v, grad=True):
    return torch.tensor([v], device=DEV, requires_grad=grad)


@pytest.mark.parametrize("method", ["STE", "LSQ", "EWGS"])
@pytest.mark.parametrize("logs,b,signed", [((-3.0, 1.0), -1.0, True), ((-4.37, 2.21), -2.3, True),
                                           ((-9.913, 0.087), -0.47, True), ((-4.0, 2.5), 0.0, False)])
def test_act_layer_matches_eager_oracle_on_device(ops, method, logs, b, signed):
    gen = torch.Generator().manual_seed(int(abs(logs[0]) * 1000))
    shape = (6, 16, 20, 20)
    x = (torch.randn(*shape, generator=gen) * 1.5)
    if not signed:
        x = x.relu()
    g = torch.randn(*shape, generator=gen)
    r = torch.randint(0, 2, shape, generator=gen).float() - 0.5
    x, g, r = x.to(DEV), g.to(DEV), r.to(DEV)
    # oracle on the device
    xr = x.clone().requires_grad_(True)
    ls_r, lq_r, b_r = P(logs[0]), P(logs[1]), P(b, signed)
    y_r, q_r = O.act_fake_quant(xr, ls_r, lq_r, b_r, r=r, method=method)
    y_r.backward(g)
    # fused layer op
    xg = x.clone().requires_grad_(True)
    ls, lq, bb = P(logs[0]), P(logs[1]), P(b, signed)
    y, params = ops.fake_quant_act_layer(xg, ls, lq, bb, method, r_sign=(r * 2).to(torch.int8))
    y.backward(g)
    s, qr = torch.exp2(ls.detach()), torch.exp2(lq.detach())
    assert torch.equal(params[0:1], s) and torch.equal(params[4:5], qr)          # device exp2 == torch's
    assert torch.equal(params[3:4], (bb.detach() + qr) - s) and torch.equal(params[1:3], bb.detach().repeat(2))
    assert bit_equal(y.detach().cpu().numpy(), y_r.detach().cpu().numpy())
    assert value_equal(xg.grad.cpu().numpy(), xr.grad.cpu().numpy())
    hi = (bb.detach() + qr) - s
    cf = CF.per_tensor(x.cpu(), g.cpu(), r.cpu(), s.cpu(), bb.detach().cpu(), bb.detach().cpu(), hi.cpu(), method)
    ln2 = math.log(2.0)
    yard_s = (float(cf["abs_s"]) + float(cf["abs_g"])) * float(s) * ln2
    yard_q = float(cf["abs_g"]) * float(qr) * ln2
    assert abs(float(ls.grad) - float(ls_r.grad)) <= 1e-6 * yard_s
    assert abs(float(lq.grad) - float(lq_r.grad)) <= 1e-6 * yard_q + 1e-30
    exact_ls = (float(cf["g_s"]) - float(cf["g_hi"])) * float(s) * ln2
    assert abs(float(ls.grad) - exact_ls) <= 2e-7 * yard_s
    if signed:
        assert abs(float(bb.grad) - float(b_r.grad)) <= 1e-6 * float(cf["abs_g"])
    else:
        assert bb.grad is None


def test_act_layer_eval_stats_and_flags(ops):
    x = torch.randn(3, 8, 9, 9, device=DEV) * 2
    ls, lq, b = P(-3.0, False), P(2.0, False), P(-2.0, False)
    y, params, qstats, flags = ops.fake_quant_act_layer_eval(x, ls, lq, b)
    y_r, q_r = O.act_fake_quant(x, ls, lq, b, method="LSQ")
    assert torch.equal(y, y_r)
    assert float(qstats[0]) == float(q_r.min()) and float(qstats[1]) == float(q_r.max())
    assert int(flags.item()) == 0
    x[0, 0, 0, 0] = float("inf")
    _, _, _, flags = ops.fake_quant_act_layer_eval(x, ls, lq, b)
    assert int(flags.item()) == 0            # +inf clamps to hi: still a valid index
    x[0, 0, 0, 1] = float("nan")
    _, _, _, flags = ops.fake_quant_act_layer_eval(x, ls, lq, b)
    assert int(flags.item()) & 4             # NaN is not an integer (gdnsq.py:216)


@pytest.mark.parametrize("method", ["STE", "LSQ", "EWGS", "AEWGS"])
@pytest.mark.parametrize("shape", [(16, 8, 3, 3), (64, 64, 3, 3), (12, 12, 3, 3), (10, 37)])
def test_weight_layer_with_fused_regulariser(ops, method, shape):
    gen = torch.Generator().manual_seed(shape[0] + len(shape))
    fan = int(np.prod(shape[1:]))
    w = torch.randn(*shape, generator=gen) * math.sqrt(2.0 / fan)
    w[1].flatten()[[0, 3]] = w[1].min() - 0.01          # tied minima
    w[2].flatten()[[1, 2, 5]] = w[2].max() + 0.02       # tied maxima
    G = torch.randn(*shape, generator=gen)
    h = torch.randn(shape[0], generator=gen)            # upstream gradient of the regulariser inputs
    r = torch.randint(0, 2, shape, generator=gen).float() - 0.5
    dims = tuple(range(1, len(shape)))
    span = w.amax(dims) - w.amin(dims)
    ls0 = (torch.log2(span / 15.0) + 0.2 * torch.randn(shape[0], generator=gen)).reshape([shape[0]] + [1] * len(dims))
    w, G, h, r, ls0 = (t.to(DEV) for t in (w, G, h, r, ls0))
    # eager oracle on the device: layer forward + ModelHelper's second amin/amax sweep
    wr, lsr = w.clone().requires_grad_(True), ls0.clone().requires_grad_(True)
    wq_r, _, zp_r = O.weight_fake_quant(wr, lsr, True, method, r=r)
    lwq_r = torch.log2(wr.amax(dims) - wr.amin(dims) + torch.exp2(lsr.ravel()))
    ((wq_r * G).sum() + (lwq_r * h).sum()).backward()
    # fused
    wg, lsg = w.clone().requires_grad_(True), ls0.clone().requires_grad_(True)
    wq, zp, s, lwq = ops.fake_quant_weight_layer(wg, lsg, method, r_sign=(r * 2).to(torch.int8))
    ((wq * G).sum() + (lwq * h).sum()).backward()
    assert torch.equal(s.ravel(), torch.exp2(ls0).ravel())
    assert torch.equal(zp.ravel(), zp_r.detach().ravel())
    assert bit_equal(wq.detach().cpu().numpy(), wq_r.detach().cpu().numpy())
    assert bit_equal(lwq.detach().cpu().numpy(), lwq_r.detach().cpu().numpy())
    cf = CF.per_channel(w.cpu(), G.cpu(), r.cpu(), s.detach().cpu().reshape(-1), method)
    abs_g = cf["abs_g"].reshape([-1] + [1] * len(dims)).numpy() + np.abs(h.cpu().numpy()).reshape([-1] + [1] * len(dims)) * 4
    err = np.abs(wg.grad.cpu().numpy() - wr.grad.cpu().numpy())
    assert np.all(err <= 1e-6 * (abs_g + np.abs(wr.grad.cpu().numpy()))), err.max()
    yard = (cf["abs_s"].numpy() + np.abs(h.cpu().numpy()) * 4) * math.log(2.0) * s.detach().cpu().numpy().reshape(-1) * 2
    errs = np.abs(lsg.grad.cpu().numpy().reshape(-1) - lsr.grad.cpu().numpy().reshape(-1))
    # AEWGS: + the propagated slack of its group means (tests/aewgs_bound.py), not a blanket factor on the yardstick
    sl_ls = aewgs_weight_slacks(w, G, s.detach().reshape(-1), True)[1] if method == "AEWGS" else 0.0
    assert np.all(errs <= 1e-6 * yard + sl_ls + 1e-9), (errs / yard).max()


def test_get_model_values_uses_fresh_fused_value_only(ops):
    import mhaq_amd as M
    from mhaq_amd import wrap
    torch.manual_seed(0)
    net = torch.nn.Sequential(M.NoisyAct(signed=True),
                              M.NoisyConv2d(3, 6, 3, qscheme=M.QScheme.PER_CHANNEL, log_s_init=-5,
                                            qnmethod=M.QNMethod.LSQ)).to(DEV).train()
    conv = net[1]
    assert conv.regulariser_input() is None
    net(torch.randn(2, 3, 8, 8, device=DEV))
    fused = conv.regulariser_input()
    assert fused is not None
    las, laq, lws, lwq = wrap.get_model_values(net, M.QScheme.PER_CHANNEL)
    assert lwq.data_ptr() == fused.data_ptr() or torch.equal(lwq, fused)
    ref = torch.log2(conv.weight.amax((1, 2, 3)) - conv.weight.amin((1, 2, 3)) + torch.exp2(conv.log_wght_s.ravel()))
    assert torch.equal(lwq.detach(), ref.detach())
    with torch.no_grad():
        conv.weight.mul_(1.5)                 # optimizer step: the cached value is stale now
    assert conv.regulariser_input() is None
    _, _, _, lwq2 = wrap.get_model_values(net, M.QScheme.PER_CHANNEL)
    ref2 = torch.log2(conv.weight.amax((1, 2, 3)) - conv.weight.amin((1, 2, 3)) + torch.exp2(conv.log_wght_s.ravel()))
    assert torch.equal(lwq2.detach(), ref2.detach())


def test_per_channel_model_with_potential_loss_matches_oracle(ops):
    """A 3-conv per-channel model + PotentialLossNoPred (bit-width hinge active): every gradient,
    including the regulariser's amin/amax scatter into the weights, HIP vs eager oracle on device."""
    import copy
    import mhaq_amd as M
    from mhaq_amd import wrap
    from oracle.loss import PotentialLossNoPred
    from oracle.ref_layers import ORACLE_LAYERS
    torch.manual_seed(3)
    base = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(),
                               torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.ReLU(),
                               torch.nn.Conv2d(8, 4, 3, padding=1))
    ref, gpu = copy.deepcopy(base), copy.deepcopy(base).to(DEV)
    wrap.quantize_model(ref, 1, "LSQ", (), layers=ORACLE_LAYERS)
    ref.to(DEV)
    wrap.quantize_model(gpu, 1, "LSQ", ())
    with torch.no_grad():
        for net in (ref, gpu):
            for m in net.modules():
                if hasattr(m, "log_act_s"):
                    m.log_act_s.fill_(-5.3); m.log_act_q.fill_(3.4); m.act_b.fill_(-3.1 if m.signed else 0.0)
                    if hasattr(m, "Q"):
                        m.Q.qnmethod = M.QNMethod.LSQ
                    else:
                        m.qnmethod = "LSQ"
                if hasattr(m, "log_wght_s"):
                    m.log_wght_s.fill_(-9.2)
    signed = [m.signed for m in gpu.modules() if isinstance(m, M.NoisyAct)]
    assert signed == [True, False, False]           # conv after ReLU -> unsigned (gdnsq_quant.py:128-139)
    x = torch.randn(4, 3, 12, 12, device=DEV)
    losses = []
    rec = Recorder(gpu)
    for net in (ref, gpu):
        net.train()
        crit = PotentialLossNoPred(None, p=1, a=4, w=4)
        crit.t, crit.loss_sum, crit.cnt = 0.7, torch.tensor(2.0, device=DEV), 2
        out = net(x)
        vals = wrap.get_model_values(net, 1)
        for v in vals[:3]:
            v.retain_grad()                   # PotentialLoss also reads the log-parameters directly
        loss = crit((out.square().mean(), *vals))
        loss.backward()
        losses.append(float(loss.detach()))
    rec.close()
    # every quantizer against its closed form on the tensors it saw (incl. the PotentialLoss regulariser's share
    # of d/dlog_wght_s and of gW at the row extremes): the op-level bar inside the model
    assert rec.check(rel=1e-6, direct=Recorder.direct_grads(gpu, vals)) == 6
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
        else:
            assert err <= 1e-4 * float(b.abs().max()) + 1e-6 * float(b.abs().sum()) + 1e-6, (n, err)


@pytest.mark.parametrize("method", ["STE", "LSQ", "EWGS"])
@pytest.mark.parametrize("shape", [(16, 16, 3, 3), (64, 64, 3, 3), (50, 50, 3, 3), (10, 64), (3, 1, 1, 1)])
def test_small_per_tensor_weight_layer(ops, method, shape):
    """One-workgroup PER_TENSOR layer (mhaq_fq_wlayer_pt_*) vs the eager oracle on the device, with the
    regulariser input log2(max - min + s) and global tied minima / maxima."""
    gen = torch.Generator().manual_seed(shape[0] * 3 + len(shape))
    fan = int(np.prod(shape[1:]))
    w = torch.randn(*shape, generator=gen) * math.sqrt(2.0 / fan)
    if w.numel() > 8:
        w.flatten()[[1, 5]] = w.min() - 0.01
        w.flatten()[[2, 3, 7]] = w.max() + 0.02
    G = torch.randn(*shape, generator=gen)
    h = torch.randn(1, generator=gen)
    r = torch.randint(0, 2, shape, generator=gen).float() - 0.5
    ls0 = torch.log2((w.max() - w.min()) / 15.0).reshape(1) + 0.137
    w, G, h, r, ls0 = (t.to(DEV) for t in (w, G, h, r, ls0))
    assert ops.small_pt_layer_supported(w, method)
    wr, lsr = w.clone().requires_grad_(True), ls0.clone().requires_grad_(True)
    wq_r, _, zp_r = O.weight_fake_quant(wr, lsr, False, method, r=r)
    lwq_r = torch.log2(wr.amax() - wr.amin() + torch.exp2(lsr.ravel()))
    ((wq_r * G).sum() + (lwq_r * h).sum()).backward()
    wg, lsg = w.clone().requires_grad_(True), ls0.clone().requires_grad_(True)
    wq, zp, s, lwq = ops.fake_quant_weight_layer_pt(wg, lsg, method, r_sign=(r * 2).to(torch.int8))
    ((wq * G).sum() + (lwq * h).sum()).backward()
    assert torch.equal(s, torch.exp2(ls0)) and torch.equal(zp, zp_r.detach())
    assert bit_equal(wq.detach().cpu().numpy(), wq_r.detach().cpu().numpy())
    assert bit_equal(lwq.detach().cpu().numpy(), lwq_r.detach().cpu().numpy())
    cf = CF.per_channel(w.reshape(1, -1).cpu(), G.reshape(1, -1).cpu(), r.reshape(1, -1).cpu(), s.cpu(), method)
    abs_g = float(cf["abs_g"]) + abs(float(h)) * 4
    err = (wg.grad - wr.grad).abs().cpu().numpy()
    assert np.all(err <= 1e-6 * (abs_g + wr.grad.abs().cpu().numpy())), err.max()
    yard = (float(cf["abs_s"]) + abs(float(h)) * 4) * math.log(2.0) * float(s) * 2
    assert abs(float(lsg.grad) - float(lsr.grad)) <= 1e-6 * yard + 1e-9
    assert not ops.small_pt_layer_supported(w, "AEWGS")
    assert not ops.small_pt_layer_supported(torch.empty(70000, device=DEV), method)


@pytest.mark.parametrize("method", ["LSQ", "STE", "AEWGS"])
def test_multi_tensor_weight_quant_equals_per_layer_ops(ops, method):
    """mhaq_fq_wlayer_fwd_multi / _bwd_multi (one launch for every per-channel layer) against the per-layer
    fused ops, on a ResNet-20-like set of layers with different row lengths."""
    import mhaq_amd as M
    from mhaq_amd.multi import MultiTensorWeightQuant
    torch.manual_seed(4)
    shapes = [(16, 16, 3, 3), (32, 16, 3, 3), (32, 32, 3, 3), (64, 32, 3, 3), (12, 12, 3, 3), (512, 512, 3, 3)]
    net = torch.nn.ModuleList([M.NoisyConv2d(s[1], s[0], 3, bias=False, qscheme=M.QScheme.PER_CHANNEL,
                                             log_s_init=-6, qnmethod=M.QNMethod[method]) for s in shapes]).to(DEV)
    with torch.no_grad():
        for m in net:
            m.log_wght_s.add_(torch.randn_like(m.log_wght_s) * 0.3)
    Gs = [torch.randn(s, device=DEV) for s in shapes]
    hs = [torch.randn(s[0], device=DEV) for s in shapes]
    multi = MultiTensorWeightQuant(net)
    assert multi.total_co == sum(s[0] for s in shapes) and multi.max_row == 512 * 9
    seed = 77
    ops.manual_seed(seed)
    wqs = multi.run()
    lwqs = [m._precomputed[3] for m in net]
    loss = sum((wq * G).sum() for wq, G in zip(wqs, Gs)) + sum((l * h).sum() for l, h in zip(lwqs, hs))
    loss.backward()                                     # ONE backward launch, Philox offset 1
    got = [(m.weight.grad.clone(), m.log_wght_s.grad.clone()) for m in net]
    r_all = ops.fill_r(multi.total_elems, seed, 1, DEV)
    for i, m in enumerate(net):
        m.weight.grad = None
        m.log_wght_s.grad = None
        n = m.weight.numel()
        r = r_all[multi.elem_off[i]:multi.elem_off[i] + n].view(shapes[i])
        wq, zp, s, lwq = ops.fake_quant_weight_layer(m.weight, m.log_wght_s, method, r_sign=r)
        assert torch.equal(wq, wqs[i]) and torch.equal(lwq, lwqs[i])
        ((wq * Gs[i]).sum() + (lwq * hs[i]).sum()).backward()
        assert torch.equal(m.weight.grad, got[i][0]), i
        assert torch.equal(m.log_wght_s.grad, got[i][1]), i
    # the layers pick the precomputed slices up in their forward, exactly once
    multi.run()
    x = torch.randn(1, 16, 8, 8, device=DEV)
    y = net[0](x)
    assert net[0]._precomputed is None and net[0].regulariser_input() is not None
    assert torch.equal(y, torch.nn.functional.conv2d(x, ops.fake_quant_weight_layer(net[0].weight, net[0].log_wght_s, method)[0]))


def test_act_layer_on_channels_last_tensor(ops):
    """A per-tensor quantizer is elementwise: a channels_last activation is processed in place of its memory
    order (no .contiguous() copy), output and gradient keep the layout, values equal the NCHW run."""
    gen = torch.Generator().manual_seed(3)
    x = (torch.randn(4, 16, 9, 9, generator=gen) * 2).to(DEV)
    g = torch.randn(4, 16, 9, 9, generator=gen).to(DEV)
    ls, lq, b = P(-3.3), P(2.2), P(-2.4)
    xa = x.clone().requires_grad_(True)
    ya, _ = ops.fake_quant_act_layer(xa, ls, lq, b, "LSQ")
    ya.backward(g)
    ga = (ls.grad.clone(), lq.grad.clone(), b.grad.clone())
    ls.grad = lq.grad = b.grad = None
    xc = x.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yc, _ = ops.fake_quant_act_layer(xc, ls, lq, b, "LSQ")
    assert yc.is_contiguous(memory_format=torch.channels_last)
    yc.backward(g)                      # NCHW-strided upstream gradient against a channels_last input
    assert torch.equal(yc, ya) and torch.equal(xc.grad, xa.grad)
    assert xc.grad.is_contiguous(memory_format=torch.channels_last)
    # same terms, summed in a different order (the memory order differs): equal to ~1e-6 of sum|terms|
    yard = float((g.abs() * 20).sum())
    for a, c in zip(ga, (ls.grad, lq.grad, b.grad)):
        assert abs(float(a) - float(c)) <= 1e-6 * yard


def test_weight_layer_on_channels_last_weight(ops):
    gen = torch.Generator().manual_seed(5)
    w = (torch.randn(16, 8, 3, 3, generator=gen) * 0.2).to(DEV)
    G = torch.randn(16, 8, 3, 3, generator=gen).to(DEV)
    ls = torch.full((16, 1, 1, 1), -5.2, device=DEV)
    wa, la = w.clone().requires_grad_(True), ls.clone().requires_grad_(True)
    wqa, zpa, sa, lwqa = ops.fake_quant_weight_layer(wa, la, "LSQ")
    (wqa * G).sum().backward()
    wc = w.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    lc = ls.clone().requires_grad_(True)
    wqc, zpc, sc, lwqc = ops.fake_quant_weight_layer(wc, lc, "LSQ")
    assert wqc.is_contiguous(memory_format=torch.channels_last)
    (wqc * G).sum().backward()
    assert torch.equal(wqc, wqa) and torch.equal(zpc, zpa) and torch.equal(lwqc, lwqa)
    assert torch.equal(wc.grad, wa.grad) and wc.grad.is_contiguous(memory_format=torch.channels_last)
    assert torch.allclose(lc.grad, la.grad, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("n,off", [(4099, 0), (4099, 1), (3, 0), (1, 0), (8192, 3), (100003, 2)])
def test_act_layer_ragged_and_misaligned(ops, n, off):
    """Scalar tail (n % 4 != 0) and 4-byte-aligned-only views through the LOGP/ACT kernels."""
    gen = torch.Generator().manual_seed(n + off)
    base = (torch.randn(n + off, generator=gen) * 2).to(DEV)
    gbase = torch.randn(n + off, generator=gen).to(DEV)
    r = (torch.randint(0, 2, (n,), generator=gen).float() - 0.5).to(DEV)
    x, g = base[off:], gbase[off:]
    ls_r, lq_r, b_r = P(-3.1), P(1.9), P(-1.7)
    xr = x.clone().requires_grad_(True)
    y_r, _ = O.act_fake_quant(xr, ls_r, lq_r, b_r, r=r, method="STE")
    y_r.backward(g)
    ls, lq, b = P(-3.1), P(1.9), P(-1.7)
    xg = x.detach().requires_grad_(True)              # keeps the misaligned data pointer
    assert xg.data_ptr() % 16 == (4 * off) % 16
    y, _ = ops.fake_quant_act_layer(xg, ls, lq, b, "STE", r_sign=(r * 2).to(torch.int8))
    y.backward(g)
    assert torch.equal(y, y_r) and torch.equal(xg.grad, xr.grad)
    yard = float((g.abs() * 40).sum()) + 1e-6
    for a, c in ((ls, ls_r), (lq, lq_r), (b, b_r)):
        assert abs(float(a.grad) - float(c.grad)) <= 1e-6 * yard


@pytest.mark.parametrize("shape", [(3, 13001), (5, 7000), (2, 40001), (2, 40000), (3, 9216), (2, 16384), (3, 20000),
                                   (2, 32768), (2, 32772), (7, 8192), (5, 8196)])
@pytest.mark.parametrize("method", ["LSQ", "AEWGS", "STE"])
def test_per_channel_long_rows_every_code_path(ops, shape, method):
    """Long rows through every form of the per-channel kernels: register-resident rows (a whole number of float4, up to
    256 x 8, 512 x 8 and 1024 x 8 float4 per workgroup: 8192 / 16384 / 32768 floats), LDS staging for odd lengths
    (up to 64 KiB as is, up to 144 KiB through the large-LDS opt-in; backward stages a row PAIR), beyond that the
    re-reading code path.  Same results everywhere."""
    gen = torch.Generator().manual_seed(shape[1])
    w = (torch.randn(*shape, generator=gen) * 0.1).to(DEV)
    G = torch.randn(*shape, generator=gen).to(DEV)
    h = torch.randn(shape[0], generator=gen).to(DEV)
    r = (torch.randint(0, 2, shape, generator=gen).float() - 0.5).to(DEV)
    ls0 = torch.log2((w.amax(1) - w.amin(1)) / 255.0).reshape(-1, 1)
    wr, lsr = w.clone().requires_grad_(True), ls0.clone().requires_grad_(True)
    wq_r, _, zp_r = O.weight_fake_quant(wr, lsr, True, method, r=r)
    lwq_r = torch.log2(wr.amax(1) - wr.amin(1) + torch.exp2(lsr.ravel()))
    ((wq_r * G).sum() + (lwq_r * h).sum()).backward()
    wg, lsg = w.clone().requires_grad_(True), ls0.clone().requires_grad_(True)
    wq, zp, s, lwq = ops.fake_quant_weight_layer(wg, lsg, method, r_sign=(r * 2).to(torch.int8))
    ((wq * G).sum() + (lwq * h).sum()).backward()
    assert torch.equal(wq, wq_r) and torch.equal(lwq, lwq_r)
    # reduced gradients within 1e-6 * sum|terms| (AEWGS: + the propagated slack of its group means -- fp64 here, fp32 in
    # torch --, tests/aewgs_bound.py), the yardsticks of oracle/fq_closed_form.py plus the regulariser's share t = h / (u ln2)
    cf = CF.per_channel(w.cpu(), G.cpu(), r.cpu(), s.detach().cpu().reshape(-1), method)
    u = (w.amax(1) - w.amin(1) + s.detach().reshape(-1)).cpu().numpy()
    t = np.abs(h.cpu().numpy()) / (u * math.log(2.0))
    sl_gw, sl_ls = aewgs_weight_slacks(w, G, s.detach().reshape(-1), True) if method == "AEWGS" else (0.0, 0.0)
    gw, gw_r = wg.grad.cpu().numpy(), wr.grad.cpu().numpy()
    if method != "AEWGS":
        assert exact_off_extremes(gw, gw_r, w.cpu().numpy(), True, also_max=True)
    abs_g = (cf["abs_g"].numpy() + 4 * t).reshape(-1, 1)
    assert np.all(np.abs(gw - gw_r) <= 1e-6 * (abs_g + np.abs(gw_r)) + sl_gw)
    yard = (cf["abs_s"].numpy() + 4 * t) * math.log(2.0) * s.detach().cpu().numpy().reshape(-1) * 2
    errs = np.abs(lsg.grad.cpu().numpy().reshape(-1) - lsr.grad.cpu().numpy().reshape(-1))
    assert np.all(errs <= 1e-6 * yard + sl_ls + 1e-9), float((errs / (yard + 1e-30)).max())


@pytest.mark.parametrize("method", ["LSQ", "STE", "AEWGS"])
def test_large_tensor_streaming_policy_changes_no_bits(ops, method):
    """Per-channel tensors of 32 MB and more run the register-resident kernels with non-temporal accesses (fq_pc.hip,
    kPcNtBytes); the policy must not show in the results: a [2048, 4096] layer (32 MB) equals its two [1024, 4096]
    halves (16 MB each: default policy) row for row, forward and backward, the sign stream replayed at the rows'
    element offsets."""
    torch.manual_seed(11)
    co, row = 2048, 4096
    w = (torch.randn(co, row, device=DEV) * 0.05)
    G = torch.randn(co, row, device=DEV)
    ls = (torch.full((co, 1), -6.0, device=DEV) + torch.randn(co, 1, device=DEV) * 0.2)
    h = torch.randn(co, device=DEV)
    r = ops.fill_r(co * row, 3, 1, DEV).view(co, row)

    def run(sl):
        ww = w[sl].clone().requires_grad_(True)
        ll = ls[sl].clone().requires_grad_(True)
        wq, zp, s, lwq = ops.fake_quant_weight_layer(ww, ll, method, r_sign=None if method == "LSQ" else r[sl].contiguous())
        ((wq * G[sl]).sum() + (lwq * h[sl]).sum()).backward()
        return wq.detach(), lwq.detach(), ww.grad, ll.grad
    full = run(slice(0, co))
    for half in (slice(0, co // 2), slice(co // 2, co)):
        part = run(half)
        for a, b in zip(full, part):
            assert torch.equal(a[half], b)


@pytest.mark.parametrize("method", ["EWGS", "AEWGS"])
def test_ewgs_aewgs_weight_gradient_bits_over_the_exponent_range(ops, method):
    """gW = gv / s for the EWGS / AEWGS estimators is the IEEE quotient: the kernels divide by the wave-uniform scale
    with Markstein corrections (fq_common.hpp: quot) and fall back to the division outside their range.  Checked bit
    for bit against the same elementwise chain in torch on the GPU, gradients from 1e-36 to 1e30, zeros, power-of-two
    scales and a scale whose significand is all ones; AEWGS with given statistics so that delta is the same number."""
    from mhaq_amd import _lib
    L = _lib.lib()
    torch.manual_seed(21)
    co, row = 96, 1152
    w = torch.randn(co, row, device=DEV) * 0.05
    mag = 10 ** torch.empty(co, 1, device=DEV).uniform_(-36, 30)
    G = torch.randn(co, row, device=DEV) * mag
    G[:, ::9] = 0
    ls = torch.empty(co, device=DEV).uniform_(-9, -2)
    ls[:8] = ls[:8].round()
    ls[8:12] = torch.log2(torch.tensor(0.124999992549419403076171875))      # 0x3dffffff: all-ones significand
    aux = torch.empty(4, co, device=DEV)
    wq = torch.empty_like(w)
    st = torch.cuda.current_stream().cuda_stream
    assert L.mhaq_fq_wlayer_fwd(w.data_ptr(), wq.data_ptr(), ls.data_ptr(), co, row, aux[0].data_ptr(),
                                aux[1].data_ptr(), aux[2].data_ptr(), aux[3].data_ptr(), st) == 0
    s, zp = aux[0].reshape(co, 1), aux[1].reshape(co, 1)
    stats = torch.empty(3, co, device=DEV)
    assert L.mhaq_fq_pc_aewgs_stats(w.data_ptr(), G.data_ptr(), aux[0].data_ptr(), aux[1].data_ptr(), co, row,
                                    stats.data_ptr(), st) == 0
    gw = torch.empty_like(w)
    gls = torch.empty(co, device=DEV)
    m = {"EWGS": 1, "AEWGS": 2}[method]
    assert L.mhaq_fq_wlayer_bwd(w.data_ptr(), G.data_ptr(), gw.data_ptr(), gls.data_ptr(), aux[0].data_ptr(),
                                aux[1].data_ptr(), aux[2].data_ptr(), None, co, row, m,
                                stats.data_ptr() if m == 2 else None, None, None, 5, 1, None, st) == 0
    v = (w - zp) / s
    e = torch.round(v) - v
    gq = G * s
    if method == "EWGS":
        gv = gq + (-gq.abs() * e * 0.01)
    else:
        num, e2, me = (stats[i].reshape(co, 1) for i in range(3))
        delta = num / torch.clamp(e2 - me * me, min=1e-3)
        gsc = torch.clamp(1.0 * delta * (gq.sign() * e), max=0.99)
        gv = gq + (-gq * gsc)
    want = gv / s
    not_min = w != zp
    assert torch.equal(gw[not_min].view(torch.int32), want[not_min].view(torch.int32))
