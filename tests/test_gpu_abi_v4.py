"""GPU (-m gpu): what ABI v4 / round 6 added at the C boundary.

* mhaq_fq_pc_quantize: Quantizer.quantize (+ dequantize) of a [co][row] tensor with GIVEN per-row scale and zero point
  (gdnsq.py:197-208, 221-229 with gdnsq_conv2d.py:76-77's infinite bounds), against torch's eager chain on the same device:
  bit for bit on every row length and alignment, the NaN flag of gdnsq.py:216-217.
* launch status: an entry point returns ITS launch's status -- a sticky HIP error an earlier call of the caller's thread left
  behind is not reported as ours (rounds 1-5 returned hipGetLastError(), which would have)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("co,row,off", [(1, 1, 0), (3, 7, 0), (8, 36, 0), (5, 1024, 0), (2, 4099, 0), (6, 450, 1),
                                        (64, 576, 0), (4, 4608, 3), (2, 70001, 0)])
def test_pc_quantize_equals_the_eager_chain(co, row, off):
    from mhaq_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device=DEV).manual_seed(co * 131 + row)
    base = torch.randn(co * row + off, device=DEV, generator=g) * 0.4
    x = base[off:].view(co, row)                              # off != 0: a 4-byte-aligned-only view
    s = torch.rand(co, device=DEV, generator=g) * 0.07 + 0.011          # not powers of two: the division must be IEEE
    zp = x.amin(1) - torch.rand(co, device=DEV, generator=g) * 0.3      # given, NOT the row minimum
    x[0, 0] = zp[0] + 2.5 * s[0]                               # a .5 tie: round half to even
    q = torch.full((co, row), float("nan"), device=DEV)
    y = torch.full((co, row), float("nan"), device=DEV)
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert L.mhaq_fq_pc_quantize(x.data_ptr(), q.data_ptr(), y.data_ptr(), s.data_ptr(), zp.data_ptr(), co, row,
                                 flags.data_ptr(), st) == 0
    v = (x - zp[:, None]) / s[:, None]
    want = v + (torch.round(v) - v)
    assert torch.equal(q, want) and torch.equal(q, torch.round(q))
    assert torch.equal(y, want * s[:, None] + zp[:, None])
    assert int(flags.item()) == 0
    # y_out and flags are optional; a NaN anywhere sets MHAQ_FQ_FLAG_NOT_INTEGER (and only that)
    x[co - 1, row - 1] = float("nan")
    q2 = torch.empty(co, row, device=DEV)
    assert L.mhaq_fq_pc_quantize(x.data_ptr(), q2.data_ptr(), None, s.data_ptr(), zp.data_ptr(), co, row, None, st) == 0
    assert L.mhaq_fq_pc_quantize(x.data_ptr(), q2.data_ptr(), None, s.data_ptr(), zp.data_ptr(), co, row,
                                 flags.data_ptr(), st) == 0
    assert int(flags.item()) == 4 and torch.isnan(q2[co - 1, row - 1])
    assert torch.equal(q2.reshape(-1)[:-1], want.reshape(-1)[:-1])


def test_an_earlier_sticky_hip_error_is_not_reported_as_ours():
    from mhaq_amd import _lib
    L = _lib.lib()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMalloc.restype = ctypes.c_int
    hip.hipGetLastError.restype = ctypes.c_int
    # everything torch allocates or launches comes BEFORE the sticky error: torch checks hipGetLastError() after its own
    # launches and would raise the caller's stale error as its own (the behaviour this library had until round 6)
    r = torch.empty(4096, dtype=torch.int8, device=DEV)
    x = torch.randn(4096, device=DEV)
    out = torch.empty(2, device=DEV)
    nb = L.mhaq_fq_minmax_workspace_bytes(4096)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    hip.hipGetLastError()                                      # start clean
    p = ctypes.c_void_p()
    rc = hip.hipMalloc(ctypes.byref(p), 1 << 60)               # the caller's own failed call: out of memory
    try:
        assert rc != 0
        assert L.mhaq_fq_fill_r(r.data_ptr(), 4096, 7, 1, st) == 0             # OUR launch succeeded and says so
        assert L.mhaq_fq_minmax(x.data_ptr(), 4096, out.data_ptr(), ws.data_ptr(), nb, st) == 0     # a two-launch entry point
    finally:
        hip.hipGetLastError()                                  # the caller's error is the caller's to collect: clear it for torch
    torch.cuda.synchronize()
    assert float(out[0]) == float(x.min()) and float(out[1]) == float(x.max())
    assert set(r.unique().tolist()) <= {-1, 1}
