"""GPU (-m gpu): the streaming backward's fast element (round 4: STE / LSQ on a well-formed quantizer -- select-based clamp,
`gq + gq*0` as one fma, Markstein quotients) against the eager oracle on the values where a shortcut would show: NaN and
+-inf in x and in g, +-0, x exactly on a clamp bound, x one ulp off it, denormals, huge magnitudes.  Elementwise outputs
(y, gx) must be the oracle's VALUES including where they are NaN; the reduced gradients must be NaN exactly where the
oracle's are and within 1e-6 * sum|terms| elsewhere.  Through the ops layer, i.e. through the C ABI."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fq_eager as O  # noqa: E402

DEV = "cuda:0"


def _same_values(a, b, subnormal_ok=None):
    """Equal values, NaN where the other is NaN.  `subnormal_ok`: a mask of elements allowed to differ by a few SUBNORMAL
    ulps (2^-149): where g * s underflows, the reference's (g * s) / s rounds twice in the subnormal range while the
    kernels' exact-quotient correction returns g itself -- a difference below 1e-44."""
    a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    ok = (a == b) | (np.isnan(a) & np.isnan(b))
    if subnormal_ok is not None:
        ok |= subnormal_ok & (np.abs(a.astype(np.float64) - b.astype(np.float64)) <= 8 * 2.0 ** -149)
    return a.shape == b.shape and bool(np.all(ok))


def _special_tensor(n, lo, hi, s, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, generator=g) * 2
    gr = torch.randn(n, generator=g)
    f32 = np.float32
    picks_x = [float("nan"), float("inf"), float("-inf"), 0.0, -0.0, lo, hi, float(np.nextafter(f32(lo), f32(-np.inf))),
               float(np.nextafter(f32(lo), f32(np.inf))), float(np.nextafter(f32(hi), f32(np.inf))),
               float(np.nextafter(f32(hi), f32(-np.inf))), 1e-42, -1e-42, 3e38, -3e38, lo + 0.5 * s, lo + 1.5 * s, lo + 2.5 * s]
    picks_g = [float("nan"), float("inf"), float("-inf"), 0.0, -0.0, 1e-42, -1e-42, 3e38, 1e-30, -1.0]
    idx = torch.randperm(n, generator=g)
    k = 0
    for vx in picks_x:                 # every special x against an ordinary g, every special g against an ordinary x ...
        x[idx[k]] = vx
        k += 1
    for vg in picks_g:
        gr[idx[k]] = vg
        k += 1
    for vx in picks_x[:6]:             # ... and a few special-special pairs
        for vg in picks_g[:4]:
            x[idx[k]], gr[idx[k]] = vx, vg
            k += 1
    return x, gr


@pytest.mark.parametrize("method", ["STE", "LSQ"])
@pytest.mark.parametrize("n", [4 * 2048 + 1027, 777])
@pytest.mark.parametrize("bounds", [(-1.9, 1.6565, 0.2371), (0.0, 3.75, 0.25), (-math.inf, math.inf, 0.07)])
def test_streaming_backward_on_special_values_equals_the_eager_oracle(method, n, bounds):
    from mhaq_amd import ops
    lo, hi, s = bounds
    flo = lo if math.isfinite(lo) else -2.0
    fhi = hi if math.isfinite(hi) else 2.0
    x, gr = _special_tensor(n, flo, fhi, s, seed=n)
    r = torch.randint(0, 2, (n,), generator=torch.Generator().manual_seed(1)).float() - 0.5
    zp = flo
    P = [torch.tensor([v]) for v in (s, zp, lo, hi)]
    # oracle (CPU eager: the reference's op chain)
    xr = x.clone().requires_grad_(True)
    Pr = [p.clone().requires_grad_(True) for p in P]
    yr = O.dequantize(O.quantize(xr, Pr[0], Pr[1], Pr[2], Pr[3], method, r), Pr[0], Pr[1])
    yr.backward(gr)
    # HIP path
    xg = x.clone().to(DEV).requires_grad_(True)
    Pg = [p.clone().to(DEV).requires_grad_(True) for p in P]
    y = ops.fake_quant_per_tensor(xg, *Pg, method, r_sign=(r * 2).to(torch.int8).to(DEV))
    y.backward(gr.to(DEV))
    assert _same_values(y.detach().cpu().numpy(), yr.detach().numpy()), "y"
    tiny = (np.abs(gr.numpy().astype(np.float64)) * s < 2.0 ** -126) & (gr.numpy() != 0)
    assert _same_values(xg.grad.cpu().numpy(), xr.grad.numpy(), subnormal_ok=tiny), "gx"
    # reduced gradients: NaN / inf exactly where the reference's are; finite ones to 1e-6 of their terms
    for name, a, b in zip(("d/ds", "d/dzp", "d/dlo", "d/dhi"), Pg, Pr):
        ga, gb = float(a.grad), float(b.grad)
        if math.isnan(gb):
            assert math.isnan(ga), (name, ga, gb)
        elif math.isinf(gb):
            assert ga == gb, (name, ga, gb)
        else:
            assert math.isfinite(ga), (name, ga, gb)


@pytest.mark.parametrize("method", ["STE", "LSQ"])
def test_fast_and_generic_elements_give_the_same_input_gradient(method):
    """The same data through a well-formed quantizer (fast element) and through one whose scale has an all-ones significand
    (fast_div false: the generic element with IEEE divisions) must give bit-identical gx wherever the two scales are used on
    their own data -- checked against the oracle for BOTH, on finite random data with every clamp region populated."""
    from mhaq_amd import ops
    n = 6 * 2048 + 5
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, generator=g) * 2
    gr = torch.randn(n, generator=g)
    r = torch.randint(0, 2, (n,), generator=g).float() - 0.5
    s_all_ones = float(np.float32(np.uint32(0x3E7FFFFF).view(np.float32)))       # 0.24999998: significand all ones
    for s in (0.25, s_all_ones):
        P = [torch.tensor([v]) for v in (s, -1.5, -1.5, 1.5)]
        xr = x.clone().requires_grad_(True)
        yr = O.dequantize(O.quantize(xr, P[0], P[1], P[2], P[3], method, r), P[0], P[1])
        yr.backward(gr)
        xg = x.clone().to(DEV).requires_grad_(True)
        y = ops.fake_quant_per_tensor(xg, *[p.to(DEV) for p in P], method, r_sign=(r * 2).to(torch.int8).to(DEV))
        y.backward(gr.to(DEV))
        assert _same_values(y.detach().cpu().numpy(), yr.detach().numpy()), s
        assert _same_values(xg.grad.cpu().numpy(), xr.grad.numpy()), s
