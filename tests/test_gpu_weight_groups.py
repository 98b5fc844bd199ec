"""GPU (-m gpu): the weight backward in GROUPS of consecutive layers (mhaq_fq_wlayer_bwd_group,
mhaq_fq_wlayer_aewgs_stats_group, multi.py::_WeightGroup) -- what the data-parallel trainer runs: one launch, and
for AEWGS one packed statistics exchange, per group instead of per layer.  Checked against the per-layer fused ops
(which the golden / oracle suites pin), the sign stream replayed through mhaq_fq_fill_r."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(16, 16, 3, 3), (32, 16, 3, 3), (12, 12, 3, 3), (10, 5, 1, 1), (64, 32, 3, 3), (512, 512, 3, 3),
          (24, 50, 3, 3)]


def _net(M, method, shapes=SHAPES, channels_last=False):
    torch.manual_seed(4)
    net = torch.nn.ModuleList([M.NoisyConv2d(s[1], s[0], s[2], bias=False, qscheme=M.QScheme.PER_CHANNEL,
                                             log_s_init=-6, qnmethod=M.QNMethod[method]) for s in shapes]).to(DEV)
    if channels_last:
        net = net.to(memory_format=torch.channels_last)
    with torch.no_grad():
        for m in net:
            m.log_wght_s.add_(torch.randn_like(m.log_wght_s) * 0.3)
    return net


def _quantized(m):
    """(wq, lwq) of one layer the way NoisyConv2d.forward obtains them."""
    wq, _, _ = m._quantized_weight()
    return wq, m.regulariser_input()


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("method", ["LSQ", "STE", "AEWGS", "EWGS"])
def test_grouped_backward_equals_per_layer_ops(method, channels_last):
    import mhaq_amd as M
    from mhaq_amd import ops
    from mhaq_amd.multi import MultiTensorWeightQuant
    net = _net(M, method, channels_last=channels_last)
    Gs = [torch.randn(s, device=DEV) for s in SHAPES]
    if channels_last:
        Gs = [g.contiguous(memory_format=torch.channels_last) for g in Gs]
    hs = [torch.randn(s[0], device=DEV) for s in SHAPES]
    plan = MultiTensorWeightQuant(net, joint_backward=False, backward_group_elems=20000)
    # cut from the end: {24x450 = 10.8 K, 512x4608} | {64x288, 10x5, 12x108} = 19.8 K -> + 32x144 | {16x144}: alone
    assert [(g.first, g.first + g.n) for g in plan.groups] == [(5, 7), (1, 5)]
    assert plan.group_of[0] is None
    seed = 31
    ops.manual_seed(seed)
    plan.run()
    outs = [_quantized(m) for m in net]
    assert all(g.outs is not None for g in plan.groups)
    loss = sum((wq * G).sum() for (wq, _), G in zip(outs, Gs)) + sum((l * h).sum() for (_, l), h in zip(outs, hs))
    loss.backward()
    got = [(m.weight.grad.clone(), m.log_wght_s.grad.clone()) for m in net]

    def layer_matches(i, offset):
        """Layer i through its own fused op with the signs of stream (seed, offset) at the group's element offsets."""
        m, g = net[i], plan.group_of[i]
        m.weight.grad = m.log_wght_s.grad = None
        n = m.weight.numel()
        if g is None:
            r = ops.fill_r(n, seed, offset, DEV)
        else:
            e0 = plan.elem_off[i] - g.elem0
            r = ops.fill_r(g.elems, seed, offset, DEV)[e0:e0 + n]      # flat: the order the kernel walks the weight in
        wq, zp, s, lwq = ops.fake_quant_weight_layer(m.weight, m.log_wght_s, method,
                                                     r_sign=None if method == "LSQ" else r)
        assert torch.equal(wq, outs[i][0]) and torch.equal(lwq, outs[i][1])
        ((wq * Gs[i]).sum() + (lwq * hs[i]).sum()).backward()
        return torch.equal(m.weight.grad, got[i][0]) and torch.equal(m.log_wght_s.grad, got[i][1])

    # three backward launches drew the streams 1, 2, 3 (in the order autograd ran them): one per unit
    units = [[0], list(range(1, 5)), list(range(5, 7))]
    used = []
    for unit in units:
        hits = [o for o in (1, 2, 3) if all(layer_matches(i, o) for i in unit)]
        assert hits, unit
        used.append(hits[0] if method != "LSQ" else None)
    if method != "LSQ":
        assert sorted(used) == [1, 2, 3]


def test_group_statistics_launch_equals_per_layer_statistics():
    """mhaq_fq_wlayer_aewgs_stats_group over a window of the model-wide aux slab == mhaq_fq_pc_aewgs_stats per layer;
    mhaq_fq_wlayer_bwd_group with those statistics == the per-layer backward with them (the N > 1 path)."""
    import mhaq_amd as M
    from mhaq_amd import _lib, ops
    from mhaq_amd.multi import MultiTensorWeightQuant, _Desc
    L = _lib.lib()
    net = _net(M, "AEWGS")
    Gs = [torch.randn(s, device=DEV) for s in SHAPES]
    plan = MultiTensorWeightQuant(net, joint_backward=False, backward_group_elems=20000)
    plan.run()
    aux_all = plan.cur_aux
    st = torch.cuda.current_stream().cuda_stream
    for g in plan.groups:
        arr = (_Desc * g.n)()
        for k, i in enumerate(g.idx):
            arr[k] = _Desc(net[i].weight.data_ptr(), None, Gs[i].data_ptr(), None, plan.co[i], plan.row[i],
                           plan.elem_off[i] - g.elem0, plan.chan_off[i] - g.chan0)
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
        stats = torch.full((3, g.co), float("nan"), device=DEV)
        aux = aux_all.data_ptr() + 4 * g.chan0
        assert L.mhaq_fq_wlayer_aewgs_stats_group(table.data_ptr(), g.n, g.co, aux, plan.total_co, stats.data_ptr(),
                                                  st) == 0
        gw = torch.empty(g.elems, device=DEV)
        gls = torch.empty(g.co, device=DEV)
        # what another rank would contribute: perturb the averaged statistics so that `stats` is really used
        stats2 = stats * torch.tensor([[0.5], [1.0], [1.0]], device=DEV)
        assert L.mhaq_fq_wlayer_bwd_group(table.data_ptr(), g.n, g.co, g.max_row, aux, plan.total_co, gw.data_ptr(),
                                          gls.data_ptr(), 2, stats2.data_ptr(), 5, 9, None, st) == 0
        r_all = ops.fill_r(g.elems, 5, 9, DEV)
        for i in g.idx:
            w = net[i].weight
            co, row = plan.co[i], plan.row[i]
            c0, e0 = plan.chan_off[i] - g.chan0, plan.elem_off[i] - g.elem0
            sl = slice(plan.chan_off[i], plan.chan_off[i] + co)
            ref = torch.empty(3, co, device=DEV)
            assert L.mhaq_fq_pc_aewgs_stats(w.data_ptr(), Gs[i].data_ptr(), aux_all[0, sl].data_ptr(),
                                            aux_all[1, sl].data_ptr(), co, row, ref.data_ptr(), st) == 0
            got = stats[:, c0:c0 + co]
            # fp64 row sums rounded once: the partition (256 threads here, 64-256 per layer) does not show in fp32
            assert torch.equal(got, ref), (i, (got - ref).abs().max())
            gw_ref = torch.empty_like(w)
            gls_ref = torch.empty(co, device=DEV)
            st2 = stats2[:, c0:c0 + co].contiguous()
            r = r_all[e0:e0 + co * row].contiguous()
            assert L.mhaq_fq_wlayer_bwd(w.data_ptr(), Gs[i].data_ptr(), gw_ref.data_ptr(), gls_ref.data_ptr(),
                                        aux_all[0, sl].data_ptr(), aux_all[1, sl].data_ptr(),
                                        aux_all[2, sl].data_ptr(), None, co, row, 2, st2.data_ptr(), None,
                                        r.data_ptr(), 0, 0, None, st) == 0
            assert torch.equal(gw[e0:e0 + co * row].view_as(w), gw_ref), i
            assert torch.equal(gls[c0:c0 + co], gls_ref), i


def test_group_entry_points_reject_bad_arguments():
    from mhaq_amd import _lib
    L = _lib.lib()
    t = torch.zeros(64, device=DEV)
    p = t.data_ptr()
    assert L.mhaq_fq_wlayer_bwd_group(None, 1, 4, 4, p, 4, p, p, 0, None, 0, 0, None, None) < 0
    assert L.mhaq_fq_wlayer_bwd_group(p, 0, 4, 4, p, 4, p, p, 0, None, 0, 0, None, None) < 0
    assert L.mhaq_fq_wlayer_bwd_group(p, 1, 4, 4, p, 3, p, p, 0, None, 0, 0, None, None) < 0     # stride < group_co
    assert L.mhaq_fq_wlayer_bwd_group(p, 1, 4, 4, p, 4, p, p, 7, None, 0, 0, None, None) < 0     # unknown estimator
    assert L.mhaq_fq_wlayer_aewgs_stats_group(p, 1, 4, p, 4, None, None) < 0
    assert L.mhaq_fq_wlayer_aewgs_stats_group(p, 1, 0, p, 4, p, None) < 0


def test_a_layer_touched_after_the_forward_launch_leaves_its_group_cleanly():
    """A weight modified between run() and its layer's forward must not use the stale slice: that layer runs its
    own op; its slot in the group receives no gradient and contributes exact zeros."""
    import mhaq_amd as M
    from mhaq_amd.multi import MultiTensorWeightQuant
    net = _net(M, "LSQ")
    Gs = [torch.randn(s, device=DEV) for s in SHAPES]
    plan = MultiTensorWeightQuant(net, joint_backward=False, backward_group_elems=20000)

    def run(touch):
        for p in net.parameters():
            p.grad = None
        plan.run()
        if touch:
            with torch.no_grad():
                net[2].weight.mul_(1.0)
        outs = [_quantized(m) for m in net]
        sum((wq * G).sum() + l.sum() for (wq, l), G in zip(outs, Gs)).backward()
        return [(m.weight.grad.clone(), m.log_wght_s.grad.clone()) for m in net]
    a, b = run(False), run(True)
    for (gw0, gs0), (gw1, gs1) in zip(a, b):
        assert torch.equal(gw0, gw1) and torch.equal(gs0, gs1)


def test_trainer_with_grouped_and_per_layer_weight_backward_agree():
    """LSQ has no random term: the grouped backward changes launches, not values."""
    import mhaq_amd as M
    from mhaq_amd import nets, ops
    from mhaq_amd.qat import QATConfig, QATTrainer
    det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        res = []
        for elems in (0, 30000):
            torch.manual_seed(5)
            ops.manual_seed(5)
            cfg = QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.LSQ, act_bit=4, weight_bit=4,
                            excluded_layers=("features.init_block.conv", "output"), warmup=2, distillation=True,
                            learning_rate=1e-3, weight_backward_group_elems=elems)
            g = torch.Generator().manual_seed(2)
            calib = torch.randn(8, 3, 32, 32, generator=g).to(DEV)
            tr = QATTrainer(nets.resnet20_cifar(10).to(memory_format=torch.channels_last), cfg, DEV,
                            calib_batches=[calib], distributed=False, capture_graph=False)
            assert len(tr.weight_forward.groups) == (0 if elems == 0 else 3)
            for m in tr.net.modules():          # LSQ activations too: nothing random in the step
                if isinstance(m, M.NoisyAct):
                    m.Q.qnmethod = M.QNMethod.LSQ
            x = torch.randn(8, 3, 32, 32, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
            y = torch.randint(0, 10, (8,), generator=g).to(DEV)
            losses = [float(tr.train_step(x, y)) for _ in range(4)]
            res.append((losses, [p.detach().clone() for p in tr.net.parameters()]))
        assert res[0][0] == res[1][0]
        for a, b in zip(res[0][1], res[1][1]):
            assert torch.equal(a, b)
    finally:
        torch.backends.cudnn.deterministic = det


def test_grouped_backward_inside_a_captured_step():
    """The group's descriptor table is uploaded from a pre-allocated pinned buffer, so the step can be captured;
    replays equal eager steps bit for bit (fresh sign streams through the device-resident offset word)."""
    import mhaq_amd as M
    from mhaq_amd import nets, ops
    from mhaq_amd.qat import QATConfig, QATTrainer
    det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        res = []
        for graph in (False, True):
            torch.manual_seed(6)
            ops.manual_seed(6)
            cfg = QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.AEWGS, act_bit=4, weight_bit=4,
                            excluded_layers=("features.init_block.conv", "output"), warmup=2, distillation=True,
                            learning_rate=1e-3, weight_backward_group_elems=30000)
            g = torch.Generator().manual_seed(2)
            calib = torch.randn(8, 3, 32, 32, generator=g).to(DEV)
            tr = QATTrainer(nets.resnet20_cifar(10), cfg, DEV, calib_batches=[calib], distributed=False,
                            capture_graph=graph)
            assert tr.weight_forward.groups
            x = torch.randn(8, 3, 32, 32, generator=g).to(DEV)
            y = torch.randint(0, 10, (8,), generator=g).to(DEV)
            losses = [float(tr.train_step(x, y)) for _ in range(6)]
            assert (tr._graph is not None) == graph
            res.append((losses, [p.detach().clone() for p in tr.net.parameters()]))
        assert res[0][0] == res[1][0]
        for a, b in zip(res[0][1], res[1][1]):
            assert torch.equal(a, b)
    finally:
        torch.backends.cudnn.deterministic = det


def test_a_table_that_does_not_cover_the_grid_leaves_the_rest_untouched():
    """Device-resident tables cannot be validated by the host entry point: workgroups beyond the last layer's
    channels return instead of indexing past its rows."""
    import mhaq_amd as M
    from mhaq_amd import _lib
    from mhaq_amd.multi import MultiTensorWeightQuant, _Desc
    L = _lib.lib()
    net = _net(M, "LSQ", shapes=[(8, 4, 3, 3)])
    w = net[0].weight
    G = torch.randn_like(w)
    plan = MultiTensorWeightQuant(net, joint_backward=False)
    plan.run()
    aux = plan.cur_aux                                   # [4][8]
    arr = (_Desc * 1)()
    arr[0] = _Desc(w.data_ptr(), None, G.data_ptr(), None, 8, 36, 0, 0)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    co_grid = 12                                         # 4 workgroups more than the table's 8 channels
    aux_wide = torch.zeros(4, co_grid, device=DEV)
    aux_wide[:, :8] = aux
    gw = torch.full((co_grid * 36,), float("nan"), device=DEV)
    gls = torch.full((co_grid,), float("nan"), device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert L.mhaq_fq_wlayer_bwd_group(table.data_ptr(), 1, co_grid, 36, aux_wide.data_ptr(), co_grid, gw.data_ptr(),
                                      gls.data_ptr(), 3, None, 0, 0, None, st) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(gw[:8 * 36]).all() and torch.isfinite(gls[:8]).all()
    assert torch.isnan(gw[8 * 36:]).all() and torch.isnan(gls[8:]).all()
    stats = torch.full((3, co_grid), float("nan"), device=DEV)
    assert L.mhaq_fq_wlayer_aewgs_stats_group(table.data_ptr(), 1, co_grid, aux_wide.data_ptr(), co_grid,
                                              stats.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(stats[:, :8]).all() and torch.isnan(stats[:, 8:]).all()


def test_descriptor_pool_recycles_its_entries_without_mixing_tables():
    """More distinct dL/dWq pointer sets than the pool has entries (all kept alive, so no address repeats): the
    least recently used entry is re-filled -- after its previous upload has run -- and every backward still uses ITS
    table: same gradients as the per-layer ops each time."""
    import mhaq_amd as M
    from mhaq_amd import ops
    from mhaq_amd.multi import MultiTensorWeightQuant
    net = _net(M, "LSQ", shapes=[(8, 4, 3, 3), (6, 8, 3, 3), (5, 6, 3, 3)])
    plan = MultiTensorWeightQuant(net, joint_backward=False, backward_group_elems=1 << 20)
    assert len(plan.groups) == 1
    keep = []
    for it in range(20):
        Gs = [torch.randn_like(m.weight) for m in net]
        keep.append(Gs)                                   # keep every gradient tensor alive: 20 distinct pointer sets
        for p in net.parameters():
            p.grad = None
        plan.run()
        outs = [_quantized(m) for m in net]
        sum((wq * G).sum() + l.sum() for (wq, l), G in zip(outs, Gs)).backward()
        got = [(m.weight.grad.clone(), m.log_wght_s.grad.clone()) for m in net]
        for i, m in enumerate(net):
            m.weight.grad = m.log_wght_s.grad = None
            wq, zp, s, lwq = ops.fake_quant_weight_layer(m.weight, m.log_wght_s, "LSQ")
            ((wq * Gs[i]).sum() + lwq.sum()).backward()
            assert torch.equal(m.weight.grad, got[i][0]) and torch.equal(m.log_wght_s.grad, got[i][1]), (it, i)
    pool = plan.groups[0].pool
    assert len([k for k in pool.keys if k is not None]) == len(pool.keys)        # full, and recycled 12 times
    assert not any(pool.held)


PT_SHAPES = [(16, 16, 3, 3), (32, 16, 3, 3), (12, 12, 3, 3), (10, 5, 1, 1), (64, 64, 3, 3), (24, 50, 3, 3)]


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("method", ["LSQ", "STE", "EWGS"])
def test_per_tensor_layers_ride_the_model_wide_launch_as_one_channel(method, channels_last):
    """A PER_TENSOR layer that fits one workgroup is served by the per-channel grids as co = 1, row = numel (same
    minimum, same quantizer, same sums): the model-wide forward launch and the grouped backward give the bits of the
    layer's own one-workgroup kernels (mhaq_fq_wlayer_pt_fwd / _bwd), the sign stream replayed at the group's offsets.
    Mixed with a per-channel layer in the same group."""
    import mhaq_amd as M
    from mhaq_amd import ops
    from mhaq_amd.multi import MultiTensorWeightQuant
    torch.manual_seed(8)
    net = torch.nn.ModuleList(
        [M.NoisyConv2d(s[1], s[0], s[2], bias=False, qscheme=M.QScheme.PER_TENSOR, log_s_init=-6,
                       qnmethod=M.QNMethod[method]) for s in PT_SHAPES] +
        [M.NoisyConv2d(8, 6, 3, bias=False, qscheme=M.QScheme.PER_CHANNEL, log_s_init=-6,
                       qnmethod=M.QNMethod[method])]).to(DEV)
    if channels_last:
        net = net.to(memory_format=torch.channels_last)
    shapes = PT_SHAPES + [(6, 8, 3, 3)]
    Gs = [torch.randn(s, device=DEV) for s in shapes]
    if channels_last:
        Gs = [g.contiguous(memory_format=torch.channels_last) for g in Gs]
    hs = [torch.randn(m.log_wght_s.numel(), device=DEV) for m in net]
    plan = MultiTensorWeightQuant(net, joint_backward=False, backward_group_elems=1 << 30)
    assert plan.nlayers == 7 and plan.per_tensor == [True] * 6 + [False]
    assert plan.co[:6] == [1] * 6 and plan.row[4] == 64 * 64 * 9 and len(plan.groups) == 1
    seed = 13
    ops.manual_seed(seed)
    plan.run()
    outs = [_quantized(m) for m in net]
    assert plan.groups[0].outs is not None                      # every layer took its slot in the group's node
    loss = sum((wq * G).sum() for (wq, _), G in zip(outs, Gs)) + sum((l * h).sum() for (_, l), h in zip(outs, hs))
    loss.backward()                                             # ONE backward launch: stream (seed, 1)
    got = [(m.weight.grad.clone(), m.log_wght_s.grad.clone()) for m in net]
    r_all = ops.fill_r(plan.total_elems, seed, 1, DEV)
    for i, m in enumerate(net):
        m.weight.grad = m.log_wght_s.grad = None
        n = m.weight.numel()
        r = None if method == "LSQ" else r_all[plan.elem_off[i]:plan.elem_off[i] + n]
        if plan.per_tensor[i]:
            # the layer's own kernels walk a contiguous copy: hand them the signs in that (logical) order
            if r is not None and channels_last:
                r = torch.as_strided(r, m.weight.shape, m.weight.stride()).contiguous().reshape(-1)
            wq, zp, s, lwq = ops.fake_quant_weight_layer_pt(m.weight, m.log_wght_s, method, r_sign=r)
        else:
            wq, zp, s, lwq = ops.fake_quant_weight_layer(m.weight, m.log_wght_s, method, r_sign=r)
        assert torch.equal(wq, outs[i][0]) and torch.equal(lwq.reshape(-1), outs[i][1].reshape(-1)), i
        ((wq * Gs[i]).sum() + (lwq.reshape(-1) * hs[i]).sum()).backward()
        assert torch.equal(m.weight.grad, got[i][0]), (i, float((m.weight.grad - got[i][0]).abs().max()))
        assert torch.equal(m.log_wght_s.grad, got[i][1]), i


def test_per_tensor_trainer_with_and_without_the_model_wide_launch_agree():
    """BASELINE configs[1] with the per-tensor override (qscheme 0), LSQ throughout: the model-wide forward + grouped
    backward change launches, not values."""
    import mhaq_amd as M
    from mhaq_amd import nets, ops
    from mhaq_amd.qat import QATConfig, QATTrainer
    det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        res = []
        for on in (False, True):
            torch.manual_seed(5)
            ops.manual_seed(5)
            cfg = QATConfig(qscheme=M.QScheme.PER_TENSOR, qnmethod=M.QNMethod.LSQ, act_bit=4, weight_bit=4,
                            excluded_layers=("features.init_block.conv", "output"), warmup=2, distillation=True,
                            learning_rate=1e-3, multi_weight_forward=on)
            g = torch.Generator().manual_seed(2)
            calib = torch.randn(8, 3, 32, 32, generator=g).to(DEV)
            tr = QATTrainer(nets.resnet20_cifar(100), cfg, DEV, calib_batches=[calib], distributed=False,
                            capture_graph=False)
            assert (tr.weight_forward is not None) == on
            if on:
                assert tr.weight_forward.nlayers == 18 and len(tr.weight_forward.groups) == 1
            for m in tr.net.modules():
                if isinstance(m, M.NoisyAct):
                    m.Q.qnmethod = M.QNMethod.LSQ
            x = torch.randn(8, 3, 32, 32, generator=g).to(DEV)
            y = torch.randint(0, 100, (8,), generator=g).to(DEV)
            losses = [float(tr.train_step(x, y)) for _ in range(4)]
            res.append((losses, [p.detach().clone() for p in tr.net.parameters()]))
        assert res[0][0] == res[1][0]
        for a, b in zip(res[0][1], res[1][1]):
            assert torch.equal(a, b)
    finally:
        torch.backends.cudnn.deterministic = det


def test_noisy_linear_layers_take_part_in_the_model_wide_launches():
    """NoisyLinear shares the weight path of NoisyConv2d (gdnsq_linear.py:61-78): per-channel (one row per output
    feature) and small per-tensor Linear layers ride the same forward launch and backward group."""
    import mhaq_amd as M
    from mhaq_amd import ops
    from mhaq_amd.multi import MultiTensorWeightQuant
    torch.manual_seed(12)
    net = torch.nn.ModuleList([
        M.NoisyLinear(64, 10, qscheme=M.QScheme.PER_CHANNEL, log_s_init=-6, qnmethod=M.QNMethod.LSQ),
        M.NoisyConv2d(8, 6, 3, bias=False, qscheme=M.QScheme.PER_CHANNEL, log_s_init=-6, qnmethod=M.QNMethod.LSQ),
        M.NoisyLinear(33, 7, qscheme=M.QScheme.PER_TENSOR, log_s_init=-6, qnmethod=M.QNMethod.LSQ)]).to(DEV)
    Gs = [torch.randn_like(m.weight) for m in net]
    plan = MultiTensorWeightQuant(net, joint_backward=False, backward_group_elems=1 << 20)
    assert plan.nlayers == 3 and plan.co == [10, 6, 1] and plan.row == [64, 72, 231] and len(plan.groups) == 1
    plan.run()
    outs = [_quantized(m) for m in net]
    sum((wq * G).sum() + l.sum() for (wq, l), G in zip(outs, Gs)).backward()
    got = [(m.weight.grad.clone(), m.log_wght_s.grad.clone()) for m in net]
    for i, m in enumerate(net):
        m.weight.grad = m.log_wght_s.grad = None
        if plan.per_tensor[i]:
            wq, zp, s, lwq = ops.fake_quant_weight_layer_pt(m.weight, m.log_wght_s, "LSQ")
        else:
            wq, zp, s, lwq = ops.fake_quant_weight_layer(m.weight, m.log_wght_s, "LSQ")
        assert torch.equal(wq, outs[i][0])
        ((wq * Gs[i]).sum() + lwq.sum()).backward()
        assert torch.equal(m.weight.grad, got[i][0]) and torch.equal(m.log_wght_s.grad, got[i][1]), i
    x = torch.randn(4, 64, device=DEV)
    plan.run()
    y = net[0](x)                                   # the Linear forward picks its slice up
    assert torch.equal(y, torch.nn.functional.linear(x, outs[0][0], net[0].bias))


@pytest.mark.parametrize("nlayers", [63, 64, 65, 97])
def test_models_on_both_sides_of_64_layers_find_their_rows(nlayers):
    """Up to 64 layers a workgroup of the model-wide launches finds its layer by a ballot over one parallel read of the
    descriptor table, beyond by binary search (fq_pc.hip find_layer): the forward slices and ONE grouped backward of
    63 / 64 / 65 / 97 small LSQ layers equal the per-layer fused ops bit for bit."""
    import mhaq_amd as M
    from mhaq_amd import ops
    from mhaq_amd.multi import MultiTensorWeightQuant
    torch.manual_seed(nlayers)
    shapes = [(2 + (i * 7) % 11, 4 + 4 * (i % 5), 3, 3) if i % 3 else (3 + i % 4, 8 + 4 * (i % 3), 1, 1) for i in range(nlayers)]
    net = torch.nn.ModuleList([M.NoisyConv2d(s[1], s[0], s[2], bias=False, qscheme=M.QScheme.PER_CHANNEL,
                                             log_s_init=-6, qnmethod=M.QNMethod.LSQ) for s in shapes]).to(DEV)
    with torch.no_grad():
        for m in net:
            m.log_wght_s.add_(torch.randn_like(m.log_wght_s) * 0.3)
    Gs = [torch.randn_like(m.weight) for m in net]
    plan = MultiTensorWeightQuant(net, joint_backward=False, backward_group_elems=1 << 30)
    assert plan.nlayers == nlayers and len(plan.groups) == 1
    plan.run()
    outs = [_quantized(m) for m in net]
    sum((wq * G).sum() + l.sum() for (wq, l), G in zip(outs, Gs)).backward()
    got = [(m.weight.grad.clone(), m.log_wght_s.grad.clone()) for m in net]
    for i, m in enumerate(net):
        m.weight.grad = m.log_wght_s.grad = None
        wq, zp, s, lwq = ops.fake_quant_weight_layer(m.weight, m.log_wght_s, "LSQ")
        assert torch.equal(wq, outs[i][0]) and torch.equal(lwq, outs[i][1]), i
        ((wq * Gs[i]).sum() + lwq.sum()).backward()
        assert torch.equal(m.weight.grad, got[i][0]) and torch.equal(m.log_wght_s.grad, got[i][1]), i


@pytest.mark.parametrize("method", ["LSQ", "STE", "AEWGS"])
def test_a_per_channel_model_with_one_long_row_layer_keeps_the_per_layer_bits(method):
    """A per-channel model whose longest row is >= 8192 floats because of ONE layer (a Linear of 16 K / 40 K inputs next to
    3x3 convolutions): the model-wide launches stay at 256 threads per row (fq_pc.hip multi_threads: 1024-thread workgroups
    are for launches whose rows are whole tensors), the rows that fit 8 float4 per thread stay register-resident and the long
    rows take the unstaged body of the same grid -- incl. a row beyond the sign tile's capacity (call-by-call draws) and a
    27-float row (dword path).  Forward slices and ONE grouped backward equal the per-layer fused ops bit for bit."""
    import mhaq_amd as M
    from mhaq_amd import ops
    from mhaq_amd.multi import MultiTensorWeightQuant
    torch.manual_seed(21)
    kw = dict(qscheme=M.QScheme.PER_CHANNEL, log_s_init=-6, qnmethod=M.QNMethod[method])
    net = torch.nn.ModuleList([
        M.NoisyConv2d(3, 8, 3, bias=False, **kw),            # 27-float rows: not whole float4s
        M.NoisyConv2d(64, 24, 3, bias=False, **kw),          # 576
        M.NoisyLinear(16384, 6, **kw),                       # 16 K: beyond 8 float4 per thread at 256 threads
        M.NoisyConv2d(512, 12, 3, bias=False, **kw),         # 4608
        M.NoisyLinear(40004, 3, **kw),                       # beyond the sign tile (37,888 elements)
        M.NoisyConv2d(128, 16, 3, bias=False, **kw)]).to(DEV)
    with torch.no_grad():
        for m in net:
            m.log_wght_s.add_(torch.randn_like(m.log_wght_s) * 0.3)
    Gs = [torch.randn_like(m.weight) for m in net]
    hs = [torch.randn(m.weight.shape[0], device=DEV) for m in net]
    # by default a trainer's plan leaves per-channel rows of more than 8 K floats to their own (1024-thread, register-
    # resident) launches; long_rows=True keeps them in, which is what this test is about
    assert max(MultiTensorWeightQuant(net, joint_backward=False, backward_group_elems=1 << 30).row) == 4608
    plan = MultiTensorWeightQuant(net, joint_backward=False, backward_group_elems=1 << 30, long_rows=True)
    assert len(plan.groups) == 1 and max(plan.row) == 40004 and plan.total_co > 2 * plan.nlayers
    seed = 77
    ops.manual_seed(seed)
    plan.run()
    outs = [_quantized(m) for m in net]
    (sum((wq * G).sum() for (wq, _), G in zip(outs, Gs)) + sum((l * h).sum() for (_, l), h in zip(outs, hs))).backward()
    got = [(m.weight.grad.clone(), m.log_wght_s.grad.clone()) for m in net]
    g = plan.groups[0]

    def layer_matches(i, offset):
        m = net[i]
        m.weight.grad = m.log_wght_s.grad = None
        n = m.weight.numel()
        e0 = plan.elem_off[i] - g.elem0
        r = None if method == "LSQ" else ops.fill_r(g.elems, seed, offset, DEV)[e0:e0 + n]
        wq, zp, s, lwq = ops.fake_quant_weight_layer(m.weight, m.log_wght_s, method, r_sign=r)
        assert torch.equal(wq, outs[i][0]) and torch.equal(lwq, outs[i][1]), i
        ((wq * Gs[i]).sum() + (lwq * hs[i]).sum()).backward()
        return torch.equal(m.weight.grad, got[i][0]) and torch.equal(m.log_wght_s.grad, got[i][1])

    hits = [o for o in (1, 2, 3) if all(layer_matches(i, o) for i in range(len(net)))]
    assert hits, method


@pytest.mark.parametrize("method", ["LSQ", "STE", "AEWGS"])
def test_default_plan_next_to_excluded_long_row_layers_keeps_the_per_layer_bits(method):
    """The DEFAULT plan (long_rows=False: what a trainer builds) on the same kind of model: per-channel rows of more than
    MAX_PLAN_ROW floats stay out of the model-wide launches and run their own per-layer launches NEXT to the plan -- the
    configuration a VGG-style model actually uses.  Every layer, planned or not, equals its per-layer fused op
    bit for bit; the plan's one grouped backward and the two excluded layers draw three different sign streams, and a
    planned layer's signs sit at its element offset inside the PLAN (which no longer counts the excluded layers: the
    offsets differ from a long_rows=True plan of the same model, by design)."""
    import mhaq_amd as M
    from mhaq_amd import ops
    from mhaq_amd.multi import MultiTensorWeightQuant
    torch.manual_seed(23)
    kw = dict(qscheme=M.QScheme.PER_CHANNEL, log_s_init=-6, qnmethod=M.QNMethod[method])
    net = torch.nn.ModuleList([
        M.NoisyConv2d(64, 24, 3, bias=False, **kw),          # 576
        M.NoisyLinear(16384, 6, **kw),                       # 16 K floats per row: excluded
        M.NoisyConv2d(512, 12, 3, bias=False, **kw),         # 4608
        M.NoisyLinear(9216, 5, **kw),                        # 9 K: excluded
        M.NoisyConv2d(128, 16, 3, bias=False, **kw)]).to(DEV)
    with torch.no_grad():
        for m in net:
            m.log_wght_s.add_(torch.randn_like(m.log_wght_s) * 0.3)
    Gs = [torch.randn_like(m.weight) for m in net]
    hs = [torch.randn(m.weight.shape[0], device=DEV) for m in net]
    plan = MultiTensorWeightQuant(net, joint_backward=False, backward_group_elems=1 << 30)
    planned = [i for i, m in enumerate(net) if any(m is p for p in plan.layers)]
    assert planned == [0, 2, 4] and max(plan.row) == 4608 and len(plan.groups) == 1
    seed = 91
    ops.manual_seed(seed)
    plan.run()
    outs = [_quantized(m) for m in net]
    (sum((wq * G).sum() for (wq, _), G in zip(outs, Gs)) + sum((l * h).sum() for (_, l), h in zip(outs, hs))).backward()
    got = [(m.weight.grad.clone(), m.log_wght_s.grad.clone()) for m in net]
    g = plan.groups[0]

    def layer_matches(i, offset):
        m = net[i]
        m.weight.grad = m.log_wght_s.grad = None
        n = m.weight.numel()
        if method == "LSQ":
            r = None
        elif i in planned:
            k = planned.index(i)
            e0 = plan.elem_off[k] - g.elem0
            r = ops.fill_r(g.elems, seed, offset, DEV)[e0:e0 + n]
        else:
            r = ops.fill_r(n, seed, offset, DEV)
        wq, zp, s, lwq = ops.fake_quant_weight_layer(m.weight, m.log_wght_s, method, r_sign=r)
        assert torch.equal(wq, outs[i][0]) and torch.equal(lwq, outs[i][1]), i
        ((wq * Gs[i]).sum() + (lwq * hs[i]).sum()).backward()
        return torch.equal(m.weight.grad, got[i][0]) and torch.equal(m.log_wght_s.grad, got[i][1])

    used = []
    for unit in (planned, [1], [3]):
        hits = [o for o in (1, 2, 3) if all(layer_matches(i, o) for i in unit)]
        assert hits, (method, unit)
        used.append(hits)
    if method != "LSQ":                                    # three launches, three streams
        assert sorted(h[0] for h in used) == [1, 2, 3] and all(len(h) == 1 for h in used), used
