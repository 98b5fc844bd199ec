"""GPU (-m gpu): the BIG form of the streaming backward (fq_pt.hip `pt_bwd_kernel<..., BIG = true>`, chosen at launch for
tensors of kBwdBigElems = 20 Mi elements and more: at most 6 waves per SIMD, ONE partial row per block through
block_sum_f32, nparts = blocks) and the form just below the threshold (one partial row per WAVE, nparts = 4 x blocks) --
the dominant production kernel of the ResNet-18 step and its workspace / nparts contract on both sides of the switch.

For the NoisyAct form (`act`), the generic per-tensor form (`pt`) and the tie-counting form of the PER_TENSOR weight
layer (`count`), at n = 20 Mi + 1031 (BIG; ragged last block, n % 4 == 3 tail) and n = 20 Mi - 5 (not BIG; same):
  * gx and the finalized gradients are the SAME BITS with in-kernel signs and with the materialised stream (r_sign);
  * gx equals the reference's elementwise closed form (g1 * [lo <= x <= hi]) value for value, and the
    finalized gradients an fp64 evaluation of the same fp32 terms within 1e-6 * sum|terms| (the bar of north_star);
  * BIG only: the same tensor run as two halves, each BELOW the threshold (the per-wave form), gives the same gx bit for
    bit and the same gradients within that bound -- tie counts exactly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
STE = 0
SEED, OFF = 0xB16B16B16, 5
BIG = 20 << 20                         # kBwdBigElems (mhaq_amd/csrc/fq_pt.hip)
LN2 = 0.69314718055994531
INV_SQRT3 = np.float32(0.57735026918962584)


@pytest.fixture(scope="module")
def L():
    from mhaq_amd import _lib
    return _lib.lib()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _bits_equal(a, b):
    return torch.equal(a.view(torch.int32), b.view(torch.int32))


def _run(L, mode, x, g, params, r):
    """One backward launch + its finalize over the flat tensors x, g (views allowed); r: int8 signs or None."""
    n = x.numel()
    nb = L.mhaq_fq_pt_bwd_workspace_bytes(n)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    gx = torch.empty(n, device=DEV)
    rp = r.data_ptr() if r is not None else None
    if mode == "act":
        grads = torch.empty(3, device=DEV)
        rc = L.mhaq_fq_act_bwd(x.data_ptr(), g.data_ptr(), gx.data_ptr(), n, params.data_ptr(), STE, rp, SEED, OFF, None,
                               grads.data_ptr(), ws.data_ptr(), nb, _stream())
    else:
        grads = torch.empty(5, device=DEV)
        p = params
        rc = L.mhaq_fq_pt_bwd(x.data_ptr(), g.data_ptr(), gx.data_ptr(), n, p[0:].data_ptr(), p[1:].data_ptr(),
                              p[2:].data_ptr(), p[3:].data_ptr(), STE, None, 0, rp, SEED, OFF, None,
                              1 if mode == "count" else 0, grads.data_ptr(), ws.data_ptr(), nb, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    return gx, grads


def _reference(mode, x, g, r, params):
    """The closed forms of SURVEY.md 8a (K1) from fp32 terms summed in fp64 on the GPU -> (gx, grads[], yardsticks[])."""
    s, zp, lo, hi, qr = (params[i] for i in range(5))
    v0 = torch.minimum(torch.maximum(x, lo), hi)
    v = (v0 - zp) / s
    nz = torch.round(v) - v
    gq = g * s
    g1 = gq / s
    gx = torch.where((x >= lo) & (x <= hi), g1, torch.zeros_like(g1))
    rr = r.float() * 0.5
    t_s = g * nz + (float(INV_SQRT3) * gq) * rr                 # d/ds terms (the g * (q - v) form)
    t_zp = g - g1
    s_lo = torch.where(x < lo, g1, torch.zeros_like(g1))
    s_hi = torch.where(x > hi, g1, torch.zeros_like(g1))
    d = lambda t: float(t.double().sum())                        # noqa: E731
    a = lambda t: float(t.double().abs().sum())                  # noqa: E731
    if mode == "act":
        sf, qf = float(s), float(qr)
        grads = [(d(t_s) - d(s_hi)) * sf * LN2, d(s_hi) * qf * LN2, d(t_zp) + d(s_lo) + d(s_hi)]
        yards = [(a(t_s) + a(s_hi)) * sf * LN2, a(s_hi) * qf * LN2 + 1e-30, a(t_zp) + a(s_lo) + a(s_hi)]
    elif mode == "pt":
        grads = [d(t_s), d(t_zp), d(s_lo), d(s_hi), 0.0]
        yards = [a(t_s), a(g), a(g), a(g), 1.0]
    else:                                                        # count: hi never clips, slot 3 / 4 are tie counts
        grads = [d(t_s), d(t_zp), d(s_lo), float((x == hi).sum()), float((x == zp).sum())]
        yards = [a(t_s), a(g), a(g), 0.0, 0.0]
    return gx, grads, yards


@pytest.mark.parametrize("n", [BIG + 1031, BIG - 5])
@pytest.mark.parametrize("mode", ["act", "pt", "count"])
def test_streaming_backward_on_both_sides_of_the_big_threshold(L, mode, n):
    gen = torch.Generator(device=DEV).manual_seed(n % 1000 + len(mode))
    x = torch.randn(n, device=DEV, generator=gen) * 2
    g = torch.randn(n, device=DEV, generator=gen)
    s = 0.2371
    if mode == "count":
        # the weight layer's launch: lo = -inf, hi = the tensor's maximum, zp = its minimum (mhaq_fq_wlayer_ptl_bwd);
        # ties planted in the first block, in the last (ragged) one and in the n % 4 tail
        zp, hi = float(x.min()), float(x.max())
        for i in (3, n // 2 + 1, n - 1):
            x[i] = zp
        for i in (7, n - 2, n - 1030):
            x[i] = hi
        params = torch.tensor([s, zp, -float("inf"), hi, 0.0], device=DEV)
    else:
        zp = -1.9
        params = torch.tensor([s, zp, zp, zp + 16 * s - s, 16 * s], device=DEV)      # s, zp, lo, hi, qr
        x[5], x[n - 1] = params[2], params[3]                                         # on the (inclusive) bounds
    r = torch.empty(n, dtype=torch.int8, device=DEV)
    assert L.mhaq_fq_fill_r(r.data_ptr(), n, SEED, OFF, _stream()) == 0

    gx_k, gr_k = _run(L, mode, x, g, params, None)               # in-kernel signs (LDS sign tile)
    gx_r, gr_r = _run(L, mode, x, g, params, r)                  # the materialised stream
    assert _bits_equal(gx_k, gx_r) and _bits_equal(gr_k, gr_r)

    gx_ref, grads_ref, yards = _reference(mode, x, g, r, params)
    assert torch.equal(gx_k, gx_ref)
    got = gr_k.double().cpu().tolist()
    for i, (ref, yard) in enumerate(zip(grads_ref, yards)):
        assert abs(got[i] - ref) <= 1e-6 * yard, (mode, n, i, got[i], ref, yard)

    if n >= BIG:
        # the same elements as two launches below the threshold (one partial row per wave): element i keeps sign i
        h = 10 << 20
        assert h % 2048 == 0 and n - h < BIG
        parts = [_run(L, mode, x[a:b], g[a:b], params, r[a:b]) for a, b in ((0, h), (h, n))]
        assert _bits_equal(torch.cat([p[0] for p in parts]), gx_k)
        both = (parts[0][1].double() + parts[1][1].double()).cpu().tolist()
        for i, yard in enumerate(yards):
            assert abs(both[i] - got[i]) <= 1e-6 * yard, (mode, i, both[i], got[i], yard)


def test_partial_row_count_and_workspace_contract_at_the_threshold(L):
    """mhaq_fq_pt_bwd_partials reports the rows it wrote (nparts): blocks from 20 Mi elements up, 4 x blocks below; the
    workspace query covers both, and a workspace one byte short is refused before anything is launched."""
    import ctypes
    for n, per_block in ((BIG, 1), (BIG - 4, 4)):
        x = torch.zeros(n, device=DEV)
        params = torch.tensor([0.25, -2.0, -2.0, 1.75, 4.0], device=DEV)
        nb = L.mhaq_fq_act_bwd_workspace_bytes(n)
        blocks = ((n >> 2) + 511) // 512
        assert nb >= (blocks * per_block * 5 + 2) * 4
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        gx = torch.empty(n, device=DEV)
        nparts = ctypes.c_int32(-1)
        assert L.mhaq_fq_act_bwd_partials(x.data_ptr(), x.data_ptr(), gx.data_ptr(), n, params.data_ptr(), STE, None, SEED,
                                          OFF, None, ws.data_ptr(), nb, ctypes.byref(nparts), _stream()) == 0
        assert nparts.value == blocks * per_block
        assert L.mhaq_fq_act_bwd_partials(x.data_ptr(), x.data_ptr(), gx.data_ptr(), n, params.data_ptr(), STE, None, SEED,
                                          OFF, None, ws.data_ptr(), nb - 1, ctypes.byref(nparts), _stream()) != 0
    torch.cuda.synchronize()
