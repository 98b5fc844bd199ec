"""GPU (-m gpu): the activation gradient hub (mhaq_amd/act_hub.py) -- one finalize launch per backward pass for
every NoisyAct quantizer (mhaq_fq_act_bwd_partials + mhaq_fq_act_bwd_finalize_multi) -- must give exactly the
bits of the per-quantizer finalize (mhaq_fq_act_bwd): same partition, same order."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _stack(methods, signed):
    import mhaq_amd as M
    acts = torch.nn.ModuleList([M.NoisyAct(init_s=-4 - 0.25 * i, init_q=1 + 0.5 * i, signed=sg,
                                           qnmethod=M.QNMethod[m]) for i, (m, sg) in enumerate(zip(methods, signed))])
    return acts.to(DEV).train()


def _run(acts, xs, gs, hub, seed=11, twice=()):
    from mhaq_amd import ops
    ops.manual_seed(seed)
    for p in acts.parameters():
        p.grad = None
    if hub is not None:
        hub.begin()
    outs, gouts = [], []
    leaves = []
    for i, (a, x, g) in enumerate(zip(acts, xs, gs)):
        xi = x.detach().clone().requires_grad_(True)
        leaves.append(xi)
        outs.append(a(xi))
        gouts.append(g)
        if i in twice:                 # the same module a second time in one forward
            outs.append(a(xi * 0.5))
            gouts.append(g * 2.0)
    if hub is not None:
        hub.end()
    torch.autograd.backward(outs, gouts)
    return [p.grad.clone() if p.grad is not None else None for p in acts.parameters()], [v.grad.clone() for v in leaves]


@pytest.mark.parametrize("methods", [("LSQ",) * 5, ("STE",) * 5, ("STE", "LSQ", "EWGS", "STE", "LSQ")])
def test_joint_finalize_equals_per_quantizer_finalize(methods):
    from mhaq_amd.act_hub import ActGradHub
    gen = torch.Generator().manual_seed(3)
    shapes = [(4, 16, 33, 31), (3, 5, 7), (2, 8, 64, 64), (1000,), (6, 40, 28, 28)]
    xs = [(torch.randn(s, generator=gen) * 2).to(DEV) for s in shapes]
    gs = [torch.randn(s, generator=gen).to(DEV) for s in shapes]
    acts = _stack(methods, (True, False, True, True, False))
    ref_p, ref_x = _run(acts, xs, gs, None)
    hub = ActGradHub(acts)
    assert len(hub) == 5
    for rep in range(2):            # second pass reuses the hub's workspaces and its device table
        got_p, got_x = _run(acts, xs, gs, hub)
        for a, b in zip(ref_p, got_p):
            assert (a is None and b is None) or torch.equal(a, b)
        for a, b in zip(ref_x, got_x):
            assert torch.equal(a, b)
    st = hub.state()
    assert st["has_table"] and st["pending"] == 0 and st["tables"] == 1 and st["retired"] == 0
    unsigned = [a for a in acts if not a.signed]
    assert all(a.act_b.grad is None for a in unsigned)


def test_module_called_twice_and_partial_backward():
    """A quantizer used twice in one forward keeps the immediate finalize for its second call (its hub workspace
    holds one set of partials); a quantizer whose output does not reach the loss gets no gradient."""
    from mhaq_amd.act_hub import ActGradHub
    gen = torch.Generator().manual_seed(4)
    shapes = [(2, 6, 10, 10)] * 3
    xs = [(torch.randn(s, generator=gen) * 2).to(DEV) for s in shapes]
    gs = [torch.randn(s, generator=gen).to(DEV) for s in shapes]
    acts = _stack(("LSQ",) * 3, (True,) * 3)
    ref_p, ref_x = _run(acts, xs, gs, None, twice=(1,))
    hub = ActGradHub(acts)
    got_p, got_x = _run(acts, xs, gs, hub, twice=(1,))
    for a, b in zip(ref_p, got_p):
        assert torch.allclose(a, b, rtol=1e-6, atol=0)      # two partial results are summed by autograd: order may differ
    for a, b in zip(ref_x, got_x):
        assert torch.equal(a, b)
    # only quantizer 0 reaches the loss
    for p in acts.parameters():
        p.grad = None
    hub.begin()
    y0 = acts[0](xs[0].clone().requires_grad_(True))
    _ = acts[1](xs[1])
    hub.end()
    y0.backward(gs[0])
    assert acts[0].log_act_s.grad is not None and acts[1].log_act_s.grad is None and acts[2].log_act_s.grad is None
    # outside begin()/end() and under no_grad the layers take the ordinary path
    y = acts[2](xs[2].clone().requires_grad_(True))
    y.backward(gs[2])
    assert acts[2].log_act_s.grad is not None
    with torch.no_grad():
        hub.begin()
        assert hub.take(0) is None
        hub.end()


def test_finalize_multi_c_abi_matches_act_bwd():
    from mhaq_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator().manual_seed(8)
    sizes = [50_001, 4_096, 2_000_003]
    descs, grads_ref, keep = [], [], []
    for n in sizes:
        x = (torch.randn(n, generator=gen) * 2).to(DEV)
        g = torch.randn(n, generator=gen).to(DEV)
        y, gx1, gx2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        ls, lq, b = (torch.tensor([v], device=DEV) for v in (-3.0, 1.5, -1.2))
        params = torch.empty(5, device=DEV)
        assert L.mhaq_fq_act_fwd(x.data_ptr(), y.data_ptr(), n, ls.data_ptr(), lq.data_ptr(), b.data_ptr(),
                                 params.data_ptr(), None, None, None, 0, st) == 0
        nb = L.mhaq_fq_act_bwd_workspace_bytes(n)
        ws1 = torch.empty(nb, dtype=torch.uint8, device=DEV)
        ws2 = torch.empty(nb, dtype=torch.uint8, device=DEV)
        gr = torch.empty(3, device=DEV)
        assert L.mhaq_fq_act_bwd(x.data_ptr(), g.data_ptr(), gx1.data_ptr(), n, params.data_ptr(), 0, None, 5, 9,
                                 None, gr.data_ptr(), ws1.data_ptr(), nb, st) == 0
        nparts = ctypes.c_int32(0)
        assert L.mhaq_fq_act_bwd_partials(x.data_ptr(), g.data_ptr(), gx2.data_ptr(), n, params.data_ptr(), 0, None,
                                          5, 9, None, ws2.data_ptr(), nb, ctypes.byref(nparts), st) == 0
        assert torch.equal(gx1, gx2) and nparts.value > 0
        descs.append((ws2.data_ptr(), nparts.value))
        grads_ref.append(gr)
        keep += [ws2, params]
    import numpy as np
    table = torch.from_numpy(np.array(descs, dtype=np.int64).reshape(-1)).to(DEV)
    out = torch.empty(len(sizes), 3, device=DEV)
    assert L.mhaq_fq_act_bwd_finalize_multi(table.data_ptr(), len(sizes), out.data_ptr(), st) == 0
    assert L.mhaq_fq_act_bwd_finalize_multi(None, 0, None, st) == 0
    assert L.mhaq_fq_act_bwd_finalize_multi(None, 2, out.data_ptr(), st) == -1
    for i, gr in enumerate(grads_ref):
        assert torch.equal(out[i], gr)


def test_trainer_with_and_without_joint_finalize_agree_bit_for_bit():
    import mhaq_amd as M
    from mhaq_amd import nets, ops
    from mhaq_amd.qat import QATConfig, QATTrainer
    det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        res = []
        for joint in (False, True):
            torch.manual_seed(5)
            ops.manual_seed(5)
            cfg = QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.AEWGS, act_bit=4, weight_bit=4,
                            excluded_layers=("features.init_block.conv", "output"), warmup=2, distillation=True,
                            learning_rate=1e-3, joint_act_finalize=joint)
            g = torch.Generator().manual_seed(2)
            calib = torch.randn(8, 3, 32, 32, generator=g).to(DEV)
            tr = QATTrainer(nets.resnet20_cifar(10), cfg, DEV, calib_batches=[calib], distributed=False)
            assert (tr.act_hub is not None) == joint
            x = torch.randn(8, 3, 32, 32, generator=g).to(DEV)
            y = torch.randint(0, 10, (8,), generator=g).to(DEV)
            losses = [float(tr.train_step(x, y)) for _ in range(4)]
            res.append((losses, [p.detach().clone() for p in tr.net.parameters()]))
        assert res[0][0] == res[1][0]
        for a, b in zip(res[0][1], res[1][1]):
            assert torch.equal(a, b)
    finally:
        torch.backends.cudnn.deterministic = det


# ------------------------------------------------------------------ forward-only multi-tensor weight quantization
@pytest.mark.parametrize("channels_last", [False, True])
def test_model_wide_weight_forward_equals_per_layer_forward(channels_last):
    """MultiTensorWeightQuant(joint_backward=False): one launch quantizes every per-channel weight before the
    forward pass; each layer then only picks its slice up and keeps its OWN backward launch.  Same bits as the
    per-layer forward, same gradients; a weight touched after run() makes that layer fall back to its own launch."""
    import mhaq_amd as M
    from mhaq_amd.multi import MultiTensorWeightQuant
    torch.manual_seed(3)
    det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True      # MIOpen's default NHWC wrw kernels use atomics: not run-to-run exact
    try:
        _weight_forward_case(M, MultiTensorWeightQuant, channels_last)
    finally:
        torch.backends.cudnn.deterministic = det


def _weight_forward_case(M, MultiTensorWeightQuant, channels_last):
    net = torch.nn.Sequential(
        M.NoisyConv2d(3, 8, 3, padding=1, qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.LSQ),
        M.NoisyConv2d(8, 6, 3, padding=1, qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.STE),
        M.NoisyConv2d(6, 5, 1, qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.AEWGS),
        M.NoisyConv2d(5, 4, 3, padding=1, qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.LSQ)).to(DEV)
    if channels_last:
        net = net.to(memory_format=torch.channels_last)
    with torch.no_grad():
        for m in net:
            m.log_wght_s.fill_(-5.3)
    x = torch.randn(2, 3, 9, 9, device=DEV)
    go = torch.randn(2, 4, 9, 9, device=DEV)

    def run(multi, touch=False):
        from mhaq_amd import ops
        ops.manual_seed(9)
        for p in net.parameters():
            p.grad = None
        if multi is not None:
            multi.run()
            assert all(m._pre_fwd is not None for m in net)
        if touch:
            with torch.no_grad():
                net[1].weight.mul_(1.0)               # bumps the version: layer 1 must not use its stale slice
        out = net(x)
        lw = torch.cat([m.regulariser_input() for m in net])
        (out * go).sum().add(lw.sum()).backward()
        return out.detach().clone(), [p.grad.clone() for p in net.parameters() if p.grad is not None]
    ref_out, ref_g = run(None)
    multi = MultiTensorWeightQuant(net, joint_backward=False)      # mixed estimators are fine: only forwards batch
    assert multi.nlayers == 4
    for touch in (False, True, False):
        out, g = run(multi, touch)
        assert torch.equal(out, ref_out)
        assert len(g) == len(ref_g) and all(torch.equal(a, b) for a, b in zip(g, ref_g))
        assert all(m._pre_fwd is None for m in net)                # consumed (or discarded) by the forward
    assert len(multi._tables) == 1                                 # the pointer table is uploaded once


def test_trainer_with_and_without_model_wide_weight_forward_agree_bit_for_bit():
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
            cfg = QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.AEWGS, act_bit=4, weight_bit=4,
                            excluded_layers=("features.init_block.conv", "output"), warmup=2, distillation=True,
                            learning_rate=1e-3, multi_weight_forward=on)
            g = torch.Generator().manual_seed(2)
            calib = torch.randn(8, 3, 32, 32, generator=g).to(DEV)
            tr = QATTrainer(nets.resnet20_cifar(10).to(memory_format=torch.channels_last), cfg, DEV,
                            calib_batches=[calib], distributed=False, capture_graph=False)
            assert (tr.weight_forward is not None) == on
            x = torch.randn(8, 3, 32, 32, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
            y = torch.randint(0, 10, (8,), generator=g).to(DEV)
            losses = [float(tr.train_step(x, y)) for _ in range(4)]
            res.append((losses, [p.detach().clone() for p in tr.net.parameters()]))
        assert res[0][0] == res[1][0]
        for a, b in zip(res[0][1], res[1][1]):
            assert torch.equal(a, b)
    finally:
        torch.backends.cudnn.deterministic = det


def test_hub_retention_is_bounded_over_many_batch_shapes_and_replays_survive():
    """A long run over many batch shapes (variable-size eval batches, a long-lived service) must not grow the hub:
    eager descriptor tables live in a 4-entry LRU, an outgrown eager workspace is dropped, and only what a CAPTURED graph
    has baked into its launches is held -- until release_captured().  200 eager steps over 50 batch sizes around a
    captured graph: table count, retired workspaces and allocated device memory stay flat, and the graph (captured
    before its workspaces were outgrown) still replays the bits of the eager pass.
    (Round 2 kept every table and every outgrown workspace for ever -- its fix for a freed-workspace abort under replay,
    gpurun_out/r02_t14.log; docs/NOTEBOOK.md section 6, "A process abort".)"""
    from mhaq_amd import ops
    from mhaq_amd.act_hub import ActGradHub
    torch.manual_seed(0)
    acts = _stack(("LSQ",) * 4, (True, False, True, True))        # LSQ: nothing random, replays compare bit for bit
    hub = ActGradHub(acts)
    small = [torch.randn(6, 8, 10, 10, device=DEV) * 2 for _ in acts]
    g_small = [torch.randn_like(x) for x in small]

    def eager(xs, gs):
        return _run(acts, xs, gs, hub)

    ref_p, ref_x = eager(small, g_small)
    # capture the pass at the small shape (static inputs, the hub's workspaces of this size baked in)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eager(small, g_small)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    leaves = [x.detach().clone().requires_grad_(True) for x in small]
    for p in acts.parameters():
        p.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        hub.begin()
        outs = [a(v) for a, v in zip(acts, leaves)]
        hub.end()
        torch.autograd.backward(outs, g_small)
    static = [p.grad for p in acts.parameters()], [v.grad for v in leaves]
    st0 = hub.state()
    assert st0["captured_tables"] == 1

    def replay_equals_eager():
        graph.replay()
        torch.cuda.synchronize()
        for a, b in zip(ref_p, static[0]):
            assert (a is None and b is None) or torch.equal(a, b)
        for a, b in zip(ref_x, static[1]):
            assert torch.equal(a, b)
    replay_equals_eager()

    # 200 eager steps over 50 batch sizes, most of them larger than the captured one (workspaces are outgrown)
    sizes = [2 + 3 * i for i in range(50)]
    gen = torch.Generator().manual_seed(5)
    order = [sizes[int(i)] for i in torch.randint(0, 50, (200,), generator=gen)]
    order[:50] = sizes[::-1]                           # every size early on, the largest first
    data = {n: ([torch.randn(n, 8, 10, 10, device=DEV) for _ in acts], [torch.randn(n, 8, 10, 10, device=DEV) for _ in acts])
            for n in sizes}
    marks = []
    for step, n in enumerate(order):
        eager(*data[n])
        if step in (99, 199):
            torch.cuda.synchronize()
            marks.append((hub.state(), torch.cuda.memory_allocated()))
    (s1, m1), (s2, m2) = marks
    assert s1["tables"] <= 4 + 1 and s2["tables"] <= 4 + 1            # the eager LRU + the captured table
    assert s2["captured_tables"] == 1
    assert s2["retired"] == s1["retired"] <= len(acts)                # only the captured workspaces, outgrown once
    assert s2["workspace_bytes"] == s1["workspace_bytes"]
    assert m2 <= m1                                                   # allocated device memory is flat
    replay_equals_eager()                                             # the graph's workspaces and table are intact
    del graph
    hub.release_captured()
    s3 = hub.state()
    assert s3["captured_tables"] == 0 and s3["retired"] == 0 and s3["tables"] <= 4
    got_p, got_x = eager(small, g_small)                              # and the eager path carries on
    for a, b in zip(ref_p, got_p):
        assert (a is None and b is None) or torch.equal(a, b)


def test_eager_only_hub_never_retires_a_workspace():
    from mhaq_amd.act_hub import ActGradHub
    acts = _stack(("STE", "LSQ"), (True, True))
    hub = ActGradHub(acts)
    for n in (4, 64, 8, 256, 16, 300, 2):
        xs = [torch.randn(n, 4, 9, 9, device=DEV) for _ in acts]
        _run(acts, xs, [torch.randn_like(x) for x in xs], hub)
    st = hub.state()
    assert st["retired"] == 0 and st["captured_tables"] == 0 and st["tables"] <= 4
