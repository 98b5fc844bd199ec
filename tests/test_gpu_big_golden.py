"""GPU (-m gpu): FULL-SIZE vectors from the real reference (tests/golden/big_cases.npz, oracle/gen_golden_big.py).

The other golden fixtures are tensors of <= 9216 elements and reach the small instantiations of the kernels; the BIG form of
the streaming backward (>= 20 Mi elements), the non-temporal forward, the headline tensor [250,64,56,56] itself and the
streaming policy of the per-channel kernels (>= 32 MB) were held to the eager oracle only.  Here the
reference's own modules have run on full-size seeded inputs; the inputs are regenerated bit for bit on this box (numpy
PCG64 for the data, torch's CPU generator for the reference's randint_like draw), the product's layer entry points run on
them, and every elementwise output is compared through three 64-bit checksums of its bit pattern (+ a 256-element window),
the reduced gradients as values within 1e-6 * sum|terms|."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.golden_util import big_inputs, bits_checksum, load_cases  # noqa: E402

BIG = load_cases("big_cases.npz")


def _f(a):
    """the one value of a 0-dim or one-element fixture array"""
    return float(np.asarray(a).reshape(-1)[0])

DEV = "cuda:0"
from oracle import fq_eager as O  # noqa: E402  (the estimator names by the reference's enum value)

METHODS = O.METHODS


def _reference_signs(seed, shape):
    """The +-1 int8 form of the reference's draw: torch.manual_seed(seed) right before backward, randint_like(v, 2) - 0.5
    inside QN*.backward (gdnsq.py:54) -- a CPU generator stream, the same on every machine."""
    torch.manual_seed(int(seed))
    return (torch.randint_like(torch.empty(*shape), 2) * 2 - 1).to(torch.int8)


@pytest.mark.parametrize("name", sorted(k for k in BIG if k.startswith("act_")))
def test_full_size_activation_vectors_from_the_reference(name):
    from mhaq_amd import ops
    c = BIG[name]
    n, shape, method = int(c["n"]), tuple(int(v) for v in c["shape"]), METHODS[int(c["method"])]
    x, g = big_inputs(c["seed"], n, float(c["scale"]))
    xg = torch.from_numpy(x).reshape(shape).to(DEV).requires_grad_(True)
    gg = torch.from_numpy(g).reshape(shape).to(DEV)
    sign = None if method == "LSQ" else _reference_signs(c["seed"], shape).to(DEV)
    P = lambda v, rg=True: torch.tensor([float(v)], device=DEV, requires_grad=rg)  # noqa: E731
    ls, lq, b = P(c["log_act_s"]), P(c["log_act_q"]), P(c["act_b"])
    assert float(ls) == round(float(ls)) and float(lq) == round(float(lq))      # the device's exp2 gives the host's bits
    y, _ = ops.fake_quant_act_layer(xg, ls, lq, b, method, r_sign=sign)
    y.backward(gg)
    yh, gxh = y.detach().cpu().numpy(), xg.grad.cpu().numpy()
    mid = n // 2
    assert np.array_equal(yh.reshape(-1)[mid:mid + 256], c["y_win"]) and np.array_equal(gxh.reshape(-1)[mid:mid + 256], c["gx_win"])
    assert np.array_equal(bits_checksum(yh), c["y_sum"]), "y: some element differs from the reference's bits"
    assert np.array_equal(bits_checksum(gxh + np.float32(0.0)), c["gx_sum"]), "gx: some element differs from the reference"
    s, qr = 2.0 ** float(c["log_act_s"]), 2.0 ** float(c["log_act_q"])
    yard_s = (float(c["abs_s"]) + float(c["abs_g"])) * s * math.log(2.0)        # sum|g q| + sum|g v| + the noise and bound terms
    yard_g = float(c["abs_g"])
    assert abs(float(ls.grad) - _f(c["g_log_act_s"])) <= 1e-6 * yard_s
    assert abs(float(lq.grad) - _f(c["g_log_act_q"])) <= 1e-6 * yard_g * qr * math.log(2.0)
    assert abs(float(b.grad) - _f(c["g_act_b"])) <= 1e-6 * yard_g


@pytest.mark.parametrize("name", sorted(k for k in BIG if k.startswith("w_") and not k.startswith("wpt_")))
def test_full_size_per_channel_weight_vectors_from_the_reference(name):
    from mhaq_amd import ops
    c = BIG[name]
    co, ci, method = int(c["co"]), int(c["ci"]), METHODS[int(c["method"])]
    n = co * ci * 9
    assert n * 4 >= 32 << 20                                    # the per-channel kernels' streaming policy (fq_pc.hip kPcNtBytes)
    w, G = big_inputs(c["seed"], n, float(c["scale"]))
    shape = (co, ci, 3, 3)
    wg = torch.from_numpy(w).reshape(shape).to(DEV).requires_grad_(True)
    Gg = torch.from_numpy(G).reshape(shape).to(DEV)
    sign = None if method == "LSQ" else _reference_signs(c["seed"], shape).to(DEV)
    ls = torch.from_numpy(np.asarray(c["log_wght_s"])).reshape(co, 1, 1, 1).to(DEV).requires_grad_(True)
    wq, zp, s, _ = ops.fake_quant_weight_layer(wg, ls, method, r_sign=sign)
    wq.backward(Gg)
    assert np.array_equal(zp.detach().cpu().numpy().reshape(-1), c["zp"])
    assert np.array_equal(bits_checksum(wq.detach().cpu().numpy()), c["wq_sum"]), "wq: some element differs from the reference's bits"
    gw = wg.grad.cpu().numpy().reshape(-1)
    w2 = w.reshape(co, -1)
    off = (w2 != w2.min(axis=1, keepdims=True)).reshape(-1)    # off the row minima gW is elementwise: the reference's values
    win = slice(n // 2, n // 2 + 256)
    assert np.array_equal(gw[win][off[win]], c["gw_win"][off[win]])
    assert np.array_equal(bits_checksum(np.where(off, gw, np.float32(0.0)) + np.float32(0.0)), c["gw_off_sum"])
    err = np.abs(ls.grad.cpu().numpy().reshape(-1).astype(np.float64) - c["g_log_wght_s"].astype(np.float64))
    assert np.all(err <= 1e-6 * c["abs_s"]), float((err / c["abs_s"]).max())


@pytest.mark.parametrize("name", sorted(k for k in BIG if k.startswith("wpt_")))
def test_full_size_per_tensor_weight_vectors_from_the_reference(name):
    """A PER_TENSOR layer of 2.36 M weights: beyond one workgroup, i.e. the streaming per-tensor layer path
    (minmax -> ptl_aux -> pt_fwd / pt_bwd<COUNT> -> sum_finalize -> ptl_scalar -> tie2_scatter)."""
    from mhaq_amd import ops
    c = BIG[name]
    co, ci, method = int(c["co"]), int(c["ci"]), METHODS[int(c["method"])]
    n = co * ci * 9
    w, G = big_inputs(c["seed"], n, float(c["scale"]))
    shape = (co, ci, 3, 3)
    wg = torch.from_numpy(w).reshape(shape).to(DEV).requires_grad_(True)
    Gg = torch.from_numpy(G).reshape(shape).to(DEV)
    sign = None if method == "LSQ" else _reference_signs(c["seed"], shape).to(DEV)
    ls = torch.tensor([float(c["log_wght_s"])], device=DEV, requires_grad=True)
    assert not ops.small_pt_layer_supported(wg, method)
    wq, zp, s, _ = ops.fake_quant_weight_layer_ptl(wg, ls, method, r_sign=sign)
    wq.backward(Gg)
    assert float(zp) == float(c["zp"][0])
    assert np.array_equal(bits_checksum(wq.detach().cpu().numpy()), c["wq_sum"]), "wq: some element differs from the reference's bits"
    gw = wg.grad.cpu().numpy().reshape(-1)
    off = w != w.min()
    win = slice(n // 2, n // 2 + 256)
    assert np.array_equal(gw[win][off[win]], c["gw_win"][off[win]])
    assert np.array_equal(bits_checksum(np.where(off, gw, np.float32(0.0)) + np.float32(0.0)), c["gw_off_sum"])
    assert abs(float(ls.grad) - float(c["g_log_wght_s"][0])) <= 1e-6 * float(c["abs_s"])


@pytest.mark.parametrize("entry", ["wlayer_bwd", "wlayer_bwd_group"])
def test_full_size_aewgs_weight_gradient_given_the_reference_statistics(entry):
    """Per-channel AEWGS on 2048 rows of 4608 floats: given the reference's own three group means (recorded; the path a
    data-parallel trainer takes after its all-reduce) the estimator is elementwise, and gW off the row extremes must be the
    reference's bits -- through the per-layer launch (512 threads x 4 float4: the scalar element) and through the grouped
    launch (256 threads x 5 float4: the packed-fp32 element of fq_pc.hip), 37.7 MB each: the streaming policy."""
    from mhaq_amd import _lib
    from mhaq_amd.multi import _Desc
    L = _lib.lib()
    c = BIG["waewgs_2048x4608"]
    co, ci = int(c["co"]), int(c["ci"])
    row, n = ci * 9, co * ci * 9
    w, G = big_inputs(c["seed"], n, float(c["scale"]))
    wd = torch.from_numpy(w).reshape(co, row).to(DEV)
    Gd = torch.from_numpy(G).reshape(co, row).to(DEV)
    s = torch.exp2(torch.from_numpy(np.asarray(c["log_wght_s"]))).reshape(co).to(DEV)       # integer log-scales: exact
    zp, mx = wd.amin(1).contiguous(), wd.amax(1).contiguous()
    assert np.array_equal(zp.cpu().numpy(), c["zp"])
    stats = torch.from_numpy(np.asarray(c["stats"])).reshape(3, co).contiguous().to(DEV)
    gw = torch.full((co, row), float("nan"), device=DEV)
    gls = torch.empty(co, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    if entry == "wlayer_bwd":
        rc = L.mhaq_fq_wlayer_bwd(wd.data_ptr(), Gd.data_ptr(), gw.data_ptr(), gls.data_ptr(), s.data_ptr(), zp.data_ptr(),
                                  mx.data_ptr(), None, co, row, 2, stats.data_ptr(), None, None, 7, 1, None, st)
    else:
        desc = (_Desc * 1)(_Desc(wd.data_ptr(), None, Gd.data_ptr(), None, co, row, 0, 0))
        table = torch.frombuffer(bytearray(bytes(desc)), dtype=torch.uint8).to(DEV)
        aux = torch.stack([s, zp, mx, torch.log2((mx - zp) + s)]).contiguous()
        rc = L.mhaq_fq_wlayer_bwd_group(table.data_ptr(), 1, co, row, aux.data_ptr(), co, gw.data_ptr(), gls.data_ptr(), 2,
                                        stats.data_ptr(), 7, 1, None, st)
    assert rc == 0
    torch.cuda.synchronize()
    g = gw.cpu().numpy().reshape(-1)
    w2 = w.reshape(co, -1)
    off = ((w2 != w2.min(axis=1, keepdims=True)) & (w2 != w2.max(axis=1, keepdims=True))).reshape(-1)
    win = slice(n // 2, n // 2 + 256)
    assert np.array_equal(g[win][off[win]], c["gw_win"][off[win]])
    assert np.array_equal(bits_checksum(np.where(off, g, np.float32(0.0)) + np.float32(0.0)), c["gw_off_sum"])
    assert torch.isfinite(gls).all()
