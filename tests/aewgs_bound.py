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


def aewgs_slack(v, g, dims, weight=None, where=None, sum_dims=None, rel=1e-6):
    """What a REDUCED gradient inherits from the elementwise slack above: sum over `sum_dims` (default: everything) of
    |weight| * [where] * aewgs_gx_bound.  A reduced AEWGS gradient is held to 1e-6 * sum|terms| PLUS this -- the sum's own
    rounding and the amplified last bits of the three group means are separate budgets; a blanket factor on the first
    (rounds 2-5: 4e-6 / 5e-6) hides the second.
      d/ds         = sum [ g q - (gv / s) v + noise ]         -> weight = v          (gv / s is what the bound bounds)
      d/dzp        = sum g - sum gv / s (+ clipped shares)    -> weight = None
      d/dlo, d/dhi = sum of gv / s over the clipped elements  -> where = x < lo / x > hi
    (per-tensor activations, gdnsq_act.py:39-55; weights: the same with nothing clipped, and gW at a group's extremes carries
    g_zp / count, i.e. at most the group's whole d/dzp slack)."""
    b = aewgs_gx_bound(v, g, dims, rel)
    if weight is not None:
        b = b * torch.as_tensor(weight).detach().double().cpu().abs()
    if where is not None:
        b = b * torch.as_tensor(where).detach().cpu().double()
    return b.sum() if sum_dims is None else b.sum(dim=sum_dims)


def aewgs_weight_slacks(w, G, s, per_channel, rel=1e-6):
    """For a WEIGHT quantizer (zero point = the group minimum, bounds never clip; gdnsq_conv2d.py:71-98): the propagated
    AEWGS slack (a) on gW, per element -- its own gv / s plus, since gW at a group's extremes carries g_zp / count with
    g_zp = sum G - sum gv / s, the group's whole sum -- and (b) on d/dlog_s = d/ds * s ln2, per scale, d/ds summing
    (gv / s) * v.  Statistics groups (reduce_to_shape, gdnsq.py:150-152): dims 1.. for a [C,1,..] scale, dim 0 for the
    [1]-shaped one; gradient groups: the channel, resp. the whole tensor.  numpy arrays: (like w, [n_scales])."""
    import math
    w, G = w.detach().cpu().float(), G.detach().cpu().float()
    rows = tuple(range(1, w.dim()))
    if per_channel:
        sv = s.detach().cpu().float().reshape([w.shape[0]] + [1] * (w.dim() - 1))
        zp, stat_dims = w.amin(rows, keepdim=True), rows
        grp = lambda t: t.sum(rows, keepdim=True)  # noqa: E731
    else:
        sv = s.detach().cpu().float().reshape(())
        zp, stat_dims = w.min(), (0,)
        grp = lambda t: t.sum()  # noqa: E731
    v = (w - zp) / sv
    b = aewgs_gx_bound(v, G, stat_dims, rel)
    sl_gw = b + grp(b)
    sl_ls = grp(b * v.abs().double()).reshape(-1) * sv.double().reshape(-1) * math.log(2.0)
    return sl_gw.numpy(), sl_ls.numpy()
