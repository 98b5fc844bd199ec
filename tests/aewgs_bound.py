"""The propagated bound of an AEWGS input gradient against the reference's fp32 evaluation (test infrastructure).

gdnsq.py:113-141:  e = round(v) - v;  num, e2, me = group means of sign(gq) * e, e^2, e;  delta = num / clamp_min(e2 - me^2, 1e-3);
g_scale = clamp_max(delta * sign(gq) * e, 0.99);  gv = gq - gq * g_scale;  gx = (gv / s) * mask.

The kernels sum the three means in fp64 and round once, torch sums them in fp32 (in an order that differs between its
back ends): each mean carries a relative summation slack `rel` of the sum of its terms' magnitudes,
    |d num| <= rel * mean|e|,   |d den| <= rel * (e2 + 2 |me| mean|e|),
hence
    |d delta| <= |d num| / den + |num| |d den| / den^2,
and, since everything after delta is elementwise (tests/test_gpu_aewgs_apply_exact.py pins it bit for bit given the
statistics),
    |d gx| <= |g| * (|e| * |d delta| + rel).
The same derivation test_gpu_parity.py::test_act_matches_reference_golden states inline; here for any grouping."""
import torch


def aewgs_gx_bound(v, g, dims, rel=1e-6):
    """v: the quantizer's v = (clamp(x) - zp) / s (fp32, CPU); g = dL/dy; dims: the dims the group means run over
    (reduce_to_shape, gdnsq.py:150-152).  Returns the per-element bound on |gx - gx_reference| (fp64 tensor)."""
    v, g = v.detach().double().cpu(), g.detach().double().cpu()
    e = torch.round(v) - v
    mean = lambda t: t.mean(dim=dims, keepdim=True)  # noqa: E731
    num, e2, me, mabs = mean(g.sign() * e), mean(e * e), mean(e), mean(e.abs())
    den = (e2 - me * me).clamp_min(1e-3)
    ddelta = rel * (mabs / den + num.abs() * (e2 + 2 * me.abs() * mabs) / den ** 2)
    return g.abs() * (e.abs() * ddelta + rel)


def within(got, ref, bound):
    got, ref = (torch.as_tensor(t).detach().double().cpu() for t in (got, ref))
    return bool(((got - ref).abs() <= bound + 1e-30).all())
