"""GPU (-m gpu): sign stream v3 (include/mhaq_fq.h: one Philox call = 128 consecutive elements) on every code path that
draws signs in-kernel.  The executable definition of the stream is mhaq_fq_fill_r (held to the numpy restatement
tests/philox_ref.py by test_gpu_parity.py::test_fill_r_is_the_documented_philox_stream); here each kernel family runs
once with its in-kernel stream and once with the materialised one (`r_sign`) and must give the SAME BITS -- on the paths
that read a workgroup's LDS sign tile (full and ragged blocks, rows that start in the middle of a call, a row at the
tile's capacity) and on the ones that draw call by call (n % 4 tails, n < 4, unaligned views, rows longer than the tile,
odd-length rows, the per-element kernels)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
STE, EWGS, AEWGS, LSQ = 0, 1, 2, 3
SEED, OFF = 0x5EED1234ABCD, 77


@pytest.fixture(scope="module")
def L():
    from mhaq_amd import _lib
    return _lib.lib()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _fill_r(L, n, offset=OFF):
    r = torch.empty(n, dtype=torch.int8, device=DEV)
    assert L.mhaq_fq_fill_r(r.data_ptr(), n, SEED, offset, _stream()) == 0
    return r


def _same(a, b):
    return np.array_equal(a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("n,shift", [(3 * 2048 + 1031, 0), (2048, 0), (4099, 0), (3, 0), (1, 0), (5000, 1), (8191, 3)])
@pytest.mark.parametrize("mode", ["pt", "count", "act"])
def test_streaming_backward_in_kernel_signs_equal_the_materialised_stream(L, n, shift, mode):
    """mhaq_fq_pt_bwd / mhaq_fq_act_bwd: full blocks, a ragged last block, the n % 4 tail, n < 4 and views that are not
    16-byte aligned (`shift` elements into an allocation: the dword kernel, which draws call by call)."""
    g = torch.Generator().manual_seed(n + shift)
    base_x = (torch.randn(n + shift, generator=g) * 2).to(DEV)
    base_g = torch.randn(n + shift, generator=g).to(DEV)
    x, gr = base_x[shift:], base_g[shift:]
    params = torch.tensor([0.2371, -1.9, -1.9, 1.6565, 3.7936], device=DEV)      # s, zp, lo, hi, qr
    nb = L.mhaq_fq_pt_bwd_workspace_bytes(n)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    r = _fill_r(L, n)          # element i of the (possibly shifted) view takes sign i of the stream
    outs = []
    for rs in (None, r):
        gx = torch.empty(n + shift, device=DEV)[shift:]
        if mode == "act":
            grads = torch.empty(3, device=DEV)
            rc = L.mhaq_fq_act_bwd(x.data_ptr(), gr.data_ptr(), gx.data_ptr(), n, params.data_ptr(), STE,
                                   rs.data_ptr() if rs is not None else None, SEED, OFF, None, grads.data_ptr(),
                                   ws.data_ptr(), nb, _stream())
        else:
            grads = torch.empty(5, device=DEV)
            p = params
            rc = L.mhaq_fq_pt_bwd(x.data_ptr(), gr.data_ptr(), gx.data_ptr(), n, p[0:].data_ptr(), p[1:].data_ptr(),
                                  p[2:].data_ptr(), p[3:].data_ptr(), STE, None, 0,
                                  rs.data_ptr() if rs is not None else None, SEED, OFF, None,
                                  1 if mode == "count" else 0, grads.data_ptr(), ws.data_ptr(), nb, _stream())
        assert rc == 0
        outs.append((gx.clone(), grads.clone()))
    assert _same(outs[0][0], outs[1][0]) and _same(outs[0][1], outs[1][1]), (n, shift, mode)


def test_device_resident_offset_word_shifts_the_stream(L):
    """offset_dev (a captured step's replay counter): in-kernel signs at (offset, *offset_dev = k) are the stream offset + k."""
    n = 6000
    g = torch.Generator().manual_seed(5)
    x, gr = (torch.randn(n, generator=g) * 2).to(DEV), torch.randn(n, generator=g).to(DEV)
    params = torch.tensor([0.2371, -1.9, -1.9, 1.6565, 3.7936], device=DEV)
    nb = L.mhaq_fq_pt_bwd_workspace_bytes(n)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    word = torch.tensor([5], dtype=torch.int64, device=DEV)
    res = []
    for off, dev_word, rs in ((OFF, word, None), (OFF + 5, None, None), (0, None, _fill_r(L, n, OFF + 5))):
        gx, grads = torch.empty(n, device=DEV), torch.empty(3, device=DEV)
        assert L.mhaq_fq_act_bwd(x.data_ptr(), gr.data_ptr(), gx.data_ptr(), n, params.data_ptr(), STE,
                                 rs.data_ptr() if rs is not None else None, SEED, off,
                                 dev_word.data_ptr() if dev_word is not None else None, grads.data_ptr(), ws.data_ptr(), nb,
                                 _stream()) == 0
        res.append(grads.clone())
    assert _same(res[0], res[1]) and _same(res[0], res[2])


# rows: register-resident float4 rows; a 64-thread row; odd lengths (dword path, LDS-staged); a row at the sign tile's capacity;
# rows LONGER than the tile (call by call: float4 nibbles and, for an odd length, single elements)
# ... and odd rows whose LDS stage (two rows) sits just below / above / well above the point where the launch must opt in to more
# than 64 KB of LDS per workgroup now that the sign tile rides on top of the dynamic request (7167, 7169, 7401 floats)
@pytest.mark.parametrize("co,row", [(5, 4100), (7, 768), (3, 450), (4, 37), (3, 36864), (2, 40004), (2, 40001),
                                    (3, 7167), (3, 7169), (3, 7401), (2, 18001)])
@pytest.mark.parametrize("method", [STE, AEWGS])
def test_per_channel_backward_in_kernel_signs_equal_the_materialised_stream(L, co, row, method):
    g = torch.Generator().manual_seed(co * row)
    w = (torch.randn(co, row, generator=g) * 0.1).to(DEV)
    G = torch.randn(co, row, generator=g).to(DEV)
    s = torch.full((co,), 2.0 ** -6, device=DEV)
    zp = w.amin(1).contiguous()
    r = _fill_r(L, co * row)
    outs = []
    for rs in (None, r):
        gw, gs = torch.empty_like(w), torch.empty(co, device=DEV)
        assert L.mhaq_fq_pc_bwd(w.data_ptr(), G.data_ptr(), gw.data_ptr(), gs.data_ptr(), s.data_ptr(), zp.data_ptr(), co, row,
                                method, None, None, rs.data_ptr() if rs is not None else None, SEED, OFF, None,
                                _stream()) == 0
        outs.append((gw.clone(), gs.clone()))
    assert _same(outs[0][0], outs[1][0]) and _same(outs[0][1], outs[1][1]), (co, row, method)


@pytest.mark.parametrize("n", [1, 777, 4096, 65536])
def test_small_per_tensor_layer_in_kernel_signs_equal_the_materialised_stream(L, n):
    g = torch.Generator().manual_seed(n)
    w = (torch.randn(n, generator=g) * 0.1).to(DEV)
    G = torch.randn(n, generator=g).to(DEV)
    ls = torch.tensor([-6.0], device=DEV)
    wq, aux = torch.empty_like(w), torch.empty(4, device=DEV)
    assert L.mhaq_fq_wlayer_pt_fwd(w.data_ptr(), wq.data_ptr(), ls.data_ptr(), n, aux.data_ptr(), _stream()) == 0
    r = _fill_r(L, n)
    outs = []
    for rs in (None, r):
        gw, gls = torch.empty_like(w), torch.empty(1, device=DEV)
        assert L.mhaq_fq_wlayer_pt_bwd(w.data_ptr(), G.data_ptr(), gw.data_ptr(), gls.data_ptr(), aux.data_ptr(), None, n, STE,
                                       rs.data_ptr() if rs is not None else None, SEED, OFF, None, _stream()) == 0
        outs.append((gw.clone(), gls.clone()))
    assert _same(outs[0][0], outs[1][0]) and _same(outs[0][1], outs[1][1])


def test_per_element_and_facade_kernels_draw_the_same_stream(L):
    """mhaq_fq_vec_bwd (the quantized bias) and mhaq_fq_noise_bwd (the QN* facade): call-by-call draws."""
    n = 300
    g = torch.Generator().manual_seed(9)
    x, gr = (torch.randn(n, generator=g) * 0.3).to(DEV), torch.randn(n, generator=g).to(DEV)
    s = (torch.rand(n, generator=g) * 0.05 + 0.02).to(DEV)
    zp = (-torch.rand(n, generator=g)).to(DEV)
    r = _fill_r(L, n)
    outs = []
    for rs in (None, r):
        gx, gs, gz = torch.empty(n, device=DEV), torch.empty(n, device=DEV), torch.empty(n, device=DEV)
        assert L.mhaq_fq_vec_bwd(x.data_ptr(), gr.data_ptr(), gx.data_ptr(), gs.data_ptr(), gz.data_ptr(), s.data_ptr(),
                                 zp.data_ptr(), n, STE, None, rs.data_ptr() if rs is not None else None, SEED, OFF, None,
                                 _stream()) == 0
        outs.append(gs.clone())
    assert _same(outs[0], outs[1])
    groups, length = 6, 50
    nb = L.mhaq_fq_noise_bwd_workspace_bytes(groups, length)
    ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=DEV)
    outs = []
    for rs in (None, r):
        gv, gs = torch.empty(n, device=DEV), torch.empty(groups, device=DEV)
        assert L.mhaq_fq_noise_bwd(x.data_ptr(), gr.data_ptr(), gv.data_ptr(), gs.data_ptr(), groups, length, STE, None, 0,
                                   rs.data_ptr() if rs is not None else None, SEED, OFF, None, ws.data_ptr(), nb,
                                   _stream()) == 0
        outs.append(gs.clone())
    assert _same(outs[0], outs[1])


def test_the_stream_is_fair_and_its_calls_are_independent(L):
    """128 consecutive elements share one Philox call: the four words of a call and neighbouring calls must still look like
    independent fair coins -- mean 0, no correlation at lags 1, 32 (next word), 128 (next call)."""
    n = 1 << 22
    r = _fill_r(L, n).float()
    tol = 4.0 / np.sqrt(n)
    assert abs(float(r.mean())) < tol
    for lag in (1, 4, 32, 128):
        assert abs(float((r[lag:] * r[:-lag]).mean())) < tol, lag
    other = _fill_r(L, n, OFF + 1).float()
    assert abs(float((r * other).mean())) < tol
