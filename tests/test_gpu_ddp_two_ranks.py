"""GPU (-m gpu): two ranks sharing the one GPU of the test box, process group over gloo (RCCL refuses two
ranks on one device): the HIP layers under torch DDP -- per-rank data, packed AEWGS statistics all-reduce from
inside backward on device tensors, gradient averaging -- i.e. the N>1 path of bench.py minus the transport."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _spawn(fn, world=2):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, fn, ret)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return dict(ret)


def _w_aewgs(rank, world):
    from mhaq_amd import ops
    dev = "cuda:0"
    torch.manual_seed(0)
    w = (torch.randn(8, 4, 3, 3) * 0.2).to(dev).requires_grad_(True)
    ls = torch.full((8, 1, 1, 1), -4.0, device=dev, requires_grad=True)
    g = torch.Generator().manual_seed(100 + rank)
    G = torch.randn(8, 4, 3, 3, generator=g).to(dev)
    r8 = torch.ones(8, 4, 3, 3, dtype=torch.int8, device=dev)
    wq, zp, s, lwq = ops.fake_quant_weight_layer(w, ls, "AEWGS", r_sign=r8)
    wq.backward(G)
    return w.grad.cpu().tolist(), G.cpu().tolist(), ls.grad.cpu().tolist()


def _w_aewgs_group(rank, world):
    """Two layers whose backward is ONE group launch: one statistics launch, ONE packed all-reduce."""
    import mhaq_amd as M
    from mhaq_amd import ops
    from mhaq_amd.multi import MultiTensorWeightQuant
    dev = "cuda:0"
    torch.manual_seed(0)
    shapes = [(8, 4, 3, 3), (6, 8, 3, 3)]
    net = torch.nn.ModuleList([M.NoisyConv2d(s[1], s[0], 3, bias=False, qscheme=M.QScheme.PER_CHANNEL,
                                             log_s_init=-4, qnmethod=M.QNMethod.AEWGS) for s in shapes]).to(dev)
    g = torch.Generator().manual_seed(100 + rank)
    Gs = [torch.randn(s, generator=g).to(dev) for s in shapes]
    plan = MultiTensorWeightQuant(net, joint_backward=False, backward_group_elems=1 << 20)
    assert len(plan.groups) == 1 and plan.groups[0].n == 2
    calls = []
    real = ops._allreduce_avg_
    ops._allreduce_avg_ = lambda t: (calls.append(tuple(t.shape)), real(t))[1]
    plan.run()
    wqs = [m._quantized_weight()[0] for m in net]
    sum((wq * G).sum() for wq, G in zip(wqs, Gs)).backward()
    ops._allreduce_avg_ = real
    assert calls == [(3, 14)], calls            # one message for both layers
    return ([m.weight.detach().cpu().tolist() for m in net], [G.cpu().tolist() for G in Gs],
            [m.weight.grad.cpu().tolist() for m in net])


def _w_trainer(rank, world):
    import mhaq_amd as M
    from mhaq_amd import nets
    from mhaq_amd.qat import QATConfig, QATTrainer
    dev = "cuda:0"
    torch.manual_seed(1)
    net = nets.resnet20_cifar(10)
    cfg = QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.AEWGS, act_bit=4, weight_bit=4,
                    excluded_layers=("features.init_block.conv", "output"), warmup=2, sync_batchnorm=False)
    g = torch.Generator().manual_seed(50 + rank)
    x = torch.randn(8, 3, 32, 32, generator=g).to(dev)
    y = torch.randint(0, 10, (8,), generator=g).to(dev)
    calib = torch.randn(8, 3, 32, 32, generator=torch.Generator().manual_seed(9)).to(dev)
    tr = QATTrainer(net, cfg, dev, calib_batches=[calib])
    assert tr.distributed
    losses = [float(tr.train_step(x, y)) for _ in range(3)]
    flat = torch.cat([p.detach().flatten() for p in tr.net.parameters()]).double()
    return losses, float(flat.sum()), float(flat.abs().sum())


def _equiv_cfg():
    import mhaq_amd as M
    from mhaq_amd.qat import QATConfig
    return QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.LSQ, act_bit=4, weight_bit=4,
                     excluded_layers=("features.init_block.conv", "output"), warmup=1, distillation=False,
                     learning_rate=1e-2, sync_batchnorm=True)


def _equiv_data():
    g = torch.Generator().manual_seed(77)
    return (torch.randn(16, 3, 32, 32, generator=g), torch.randint(0, 10, (16,), generator=g),
            torch.randn(8, 3, 32, 32, generator=g))


def _equiv_run(x, y, calib, dev):
    import mhaq_amd as M
    from mhaq_amd import nets, ops
    from mhaq_amd.qat import QATTrainer
    torch.backends.cudnn.deterministic = True
    torch.manual_seed(3)
    ops.manual_seed(3)
    sgd = lambda params, lr: torch.optim.SGD(params, lr=lr)       # noqa: E731  (linear in the gradient)
    net = nets.resnet20_cifar(10).to(dev).train()
    with torch.no_grad():         # a "pretrained" network: BatchNorm running statistics that describe its activations
        warm = _equiv_data()[0].to(dev)      # (calibration runs in eval mode; on a fresh net it would fit ranges to
        for _ in range(40):                  # un-normalised activations and the first train-mode batch would clip)
            net(warm)
    tr = QATTrainer(net, _equiv_cfg(), dev, calib_batches=[calib.to(dev)], optimizer_factory=sgd,
                    capture_graph=False)
    for m in tr.net.modules():
        if isinstance(m, M.NoisyAct):
            m.Q.qnmethod = M.QNMethod.LSQ
    flat = lambda: torch.cat([p.detach().flatten() for p in tr.net.parameters()]).cpu()      # noqa: E731
    quant = torch.cat([torch.full((p.numel(),), ("log_" in n) or n.endswith("act_b"), dtype=torch.bool)
                       for n, p in tr.net.named_parameters()])
    p_init = flat()
    # two calls = ONE update: the first optimizer step runs at rate 0 (temperature_adjust.py:28-33), the second at lr
    losses = [float(tr.train_step(x.to(dev), y.to(dev))) for _ in range(2)]
    return losses, p_init, flat(), quant


def _w_equiv(rank, world):
    x, y, calib = _equiv_data()
    sl = slice(rank * 8, rank * 8 + 8)
    losses, p_init, p_final, _ = _equiv_run(x[sl], y[sl], calib, "cuda:0")
    return losses, p_init.tolist(), p_final.tolist()


def test_aewgs_packed_allreduce_on_device_tensors():
    from oracle import fq_closed_form as CF
    out = _spawn(_w_aewgs)
    torch.manual_seed(0)
    w = torch.randn(8, 4, 3, 3) * 0.2
    s = torch.exp2(torch.full((8,), -4.0))
    Gs = [torch.tensor(out[r][1]) for r in (0, 1)]
    zp = w.amin((1, 2, 3), keepdim=True)
    v = (w - zp) / s.reshape(8, 1, 1, 1)
    e = torch.round(v) - v
    num = sum(((G * s.reshape(8, 1, 1, 1)).sign() * e).mean((1, 2, 3)) for G in Gs) / 2
    stats = (num, e.square().mean((1, 2, 3)), e.mean((1, 2, 3)))
    for r in (0, 1):
        cf = CF.per_channel(w, Gs[r], torch.full_like(w, 0.5), s, "AEWGS", stats=stats)
        assert torch.allclose(torch.tensor(out[r][0]), cf["gw"], rtol=1e-5, atol=1e-6)


def test_aewgs_group_exchange_is_one_packed_allreduce():
    """gW of every layer of the group == the closed form with the statistics averaged over both ranks."""
    from oracle import fq_closed_form as CF
    out = _spawn(_w_aewgs_group)
    for layer in range(2):
        w = torch.tensor(out[0][0][layer])
        co = w.shape[0]
        s = torch.exp2(torch.full((co,), -4.0))
        Gs = [torch.tensor(out[r][1][layer]) for r in (0, 1)]
        zp = w.amin((1, 2, 3), keepdim=True)
        v = (w - zp) / s.reshape(co, 1, 1, 1)
        e = torch.round(v) - v
        num = sum(((G * s.reshape(co, 1, 1, 1)).sign() * e).mean((1, 2, 3)) for G in Gs) / 2
        stats = (num, e.square().mean((1, 2, 3)), e.mean((1, 2, 3)))
        for r in (0, 1):
            cf = CF.per_channel(w, Gs[r], torch.full_like(w, 0.5), s, "AEWGS", stats=stats)
            assert torch.allclose(torch.tensor(out[r][2][layer]), cf["gw"], rtol=1e-5, atol=1e-6), (layer, r)


def test_hip_layers_under_ddp_two_ranks_stay_in_sync():
    out = _spawn(_w_trainer)
    (l0, s0, a0), (l1, s1, a1) = out[0], out[1]
    assert all(map(lambda v: v == v, l0 + l1))          # finite
    assert l0 != l1
    assert abs(s0 - s1) <= 1e-6 * a0 and abs(a0 - a1) <= 1e-6 * a0


def test_two_ranks_equal_one_process_on_the_concatenated_batch():
    """Data parallelism of the path: two ranks with 8 samples each -- per-rank activation shards,
    replicated scalar parameters whose partial gradient sums ride DDP's averaging all-reduce, the grouped weight
    backward, SyncBatchNorm -- take the steps one process takes on all 16 samples (LSQ everywhere: nothing random,
    every gradient linear in the loss; plain SGD).  Not bit for bit: SyncBatchNorm's kernels differ from the
    single-process BatchNorm's in the last bit, a rounding decision of the next NoisyAct flips, and 18 quantized
    layers with batch-statistics BatchNorm between them amplify that (measured layer by layer: 1e-6 after the first
    BatchNorm, one grid step after the next NoisyAct, 1e-3 .. 1e-2 of the logits at the end); what a wrong reduction (SUM for AVG, a parameter left out of the all-reduce, rank-local statistics)
    would change is the size and direction of the update, and those are pinned."""
    out = _spawn(_w_equiv)
    x, y, calib = _equiv_data()
    losses, p_init, p_full, quant = _equiv_run(x, y, calib, "cuda:0")
    i0, i1 = torch.tensor(out[0][1]), torch.tensor(out[1][1])
    f0, f1 = torch.tensor(out[0][2]), torch.tensor(out[1][2])
    assert torch.equal(i0, p_init) and torch.equal(i1, p_init)       # same calibrated start everywhere
    assert torch.equal(f0, f1)                                        # the ranks hold the same parameters
    assert abs((out[0][0][0] + out[1][0][0]) / 2 - losses[0]) <= 2e-4 * abs(losses[0])   # mean of the shard losses
    for name, mask in (("all parameters", torch.ones_like(quant)), ("quantizer parameters", quant)):
        d_ddp, d_full = (f0 - p_init)[mask].double(), (p_full - p_init)[mask].double()
        cos = float(d_ddp @ d_full / (d_ddp.norm() * d_full.norm()))
        ratio = float(d_ddp.norm() / d_full.norm())
        assert cos > 0.97 and 0.95 < ratio < 1.05, (name, cos, ratio)      # measured: 0.992 / 0.999 and 0.987 / 0.986


# ------------------------------------------------------------------------------ four ranks (BASELINE configs[4] is a 4-GPU run)
def _w_trainer_counting(rank, world):
    """The DDP trainer with the exchanges counted: (losses, parameter sum, |sum|, packed all-reduces per backward,
    backward groups, seeds drawn)."""
    import mhaq_amd as M
    from mhaq_amd import nets, ops
    from mhaq_amd.qat import QATConfig, QATTrainer
    dev = "cuda:0"
    torch.manual_seed(1)
    ops.manual_seed(1)
    net = nets.resnet20_cifar(10)
    cfg = QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.AEWGS, act_bit=4, weight_bit=4,
                    excluded_layers=("features.init_block.conv", "output"), warmup=2, sync_batchnorm=False,
                    weight_backward_group_elems=60000)
    g = torch.Generator().manual_seed(50 + rank)
    x = torch.randn(8, 3, 32, 32, generator=g).to(dev)
    y = torch.randint(0, 10, (8,), generator=g).to(dev)
    calib = torch.randn(8, 3, 32, 32, generator=torch.Generator().manual_seed(9)).to(dev)
    tr = QATTrainer(net, cfg, dev, calib_batches=[calib])
    assert tr.distributed and tr.weight_forward is not None
    calls = []
    real = ops._allreduce_avg_
    ops._allreduce_avg_ = lambda t: (calls.append(tuple(t.shape)), real(t))[1]
    losses = []
    per_step = []
    for _ in range(3):
        calls.clear()
        losses.append(float(tr.train_step(x, y)))
        per_step.append(list(calls))
    ops._allreduce_avg_ = real
    wf = tr.weight_forward
    expected = sorted([(3, g_.co) for g_ in wf.groups] +
                      [(3, wf.co[i]) for i in range(wf.nlayers) if wf.group_of[i] is None])
    flat = torch.cat([p.detach().flatten() for p in tr.net.parameters()]).double()
    seed, _ = ops.rng.next()
    return losses, float(flat.sum()), float(flat.abs().sum()), [sorted(c) for c in per_step], expected, seed


@pytest.mark.parametrize("world", [4])
def test_four_ranks_parameters_in_sync_one_exchange_per_backward_group_distinct_sign_streams(world):
    """World size 4 over gloo on the one GPU of the box (RCCL wants one device per rank): parameters identical on every
    rank after 3 AEWGS steps, exactly ONE packed [3, group_co] statistics message per backward group (and one per
    ungrouped layer) per step -- the same message list on every rank, so the collectives line up --, and a different
    Philox seed per rank (ranks draw different streams)."""
    out = _spawn(_w_trainer_counting, world=world)
    sums = [out[r][1] for r in range(world)]
    scale = out[0][2]
    assert all(abs(v - sums[0]) <= 1e-6 * scale for v in sums), sums
    assert len({tuple(out[r][0]) for r in range(world)}) == world        # different data -> different losses
    for r in range(world):
        per_step, expected = out[r][3], out[r][4]
        assert len(expected) >= 2
        assert all(step == expected for step in per_step), (r, per_step, expected)
    assert len({out[r][5] for r in range(world)}) == world               # per-rank sign streams


def test_aewgs_group_exchange_at_world_size_four():
    from oracle import fq_closed_form as CF
    world = 4
    out = _spawn(_w_aewgs_group, world=world)
    for layer in range(2):
        w = torch.tensor(out[0][0][layer])
        co = w.shape[0]
        s = torch.exp2(torch.full((co,), -4.0))
        Gs = [torch.tensor(out[r][1][layer]) for r in range(world)]
        zp = w.amin((1, 2, 3), keepdim=True)
        v = (w - zp) / s.reshape(co, 1, 1, 1)
        e = torch.round(v) - v
        num = sum(((G * s.reshape(co, 1, 1, 1)).sign() * e).mean((1, 2, 3)) for G in Gs) / world
        stats = (num, e.square().mean((1, 2, 3)), e.mean((1, 2, 3)))
        for r in range(world):
            cf = CF.per_channel(w, Gs[r], torch.full_like(w, 0.5), s, "AEWGS", stats=stats)
            assert torch.allclose(torch.tensor(out[r][2][layer]), cf["gw"], rtol=1e-5, atol=1e-6), (layer, r)


# ------------------------------------------------------------------------------ a captured step under data parallelism
def _w_flat_sync(rank, world):
    """BASELINE configs[4]'s kind of model (RFDN: no BatchNorm, LSQ: no statistics exchange) on two ranks: the trainer
    that replays forward + backward as a hipGraph and all-reduces the flattened gradients once per step, against the
    eager torch-DDP trainer on the same data."""
    import mhaq_amd as M
    from mhaq_amd import nets, ops
    from mhaq_amd.qat import QATConfig, QATTrainer
    dev = "cuda:0"
    g = torch.Generator().manual_seed(60 + rank)
    x = (torch.rand(4, 3, 24, 24, generator=g) * 255.0).to(dev)
    y = (torch.rand(4, 3, 96, 96, generator=g) * 255.0).to(dev)
    calib = (torch.rand(4, 3, 24, 24, generator=torch.Generator().manual_seed(9)) * 255.0).to(dev)
    out = {}
    for name, capture in (("graph", True), ("ddp", False)):
        torch.manual_seed(1)
        ops.manual_seed(1)
        cfg = QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.LSQ, act_bit=4, weight_bit=4,
                        distillation=False, excluded_layers=("fea_conv", "upsampler.0"), learning_rate=5e-4,
                        warmup=2, criterion=torch.nn.L1Loss())
        tr = QATTrainer(nets.rfdn(), cfg, dev, calib_batches=[calib], capture_graph=capture)
        for m in tr.net.modules():
            if isinstance(m, M.NoisyAct):
                m.Q.qnmethod = M.QNMethod.LSQ                 # nothing random: the two forms must agree
        assert tr.distributed and tr._flat_sync == capture
        assert isinstance(tr.module, torch.nn.parallel.DistributedDataParallel) != capture
        flat0 = torch.cat([p.detach().flatten() for p in tr.net.parameters()]).cpu()
        losses = [float(tr.train_step(x, y)) for _ in range(7)]
        assert (tr._graph is not None) == capture
        flat = torch.cat([p.detach().flatten() for p in tr.net.parameters()]).cpu()
        out[name] = (losses, flat.tolist(), flat0.tolist())
    return out


def test_captured_step_with_one_flat_gradient_allreduce_equals_torch_ddp():
    """Data parallelism for the host-bound configurations: no collective inside the step, the step replayed as a
    hipGraph, ONE all-reduce over the flattened gradients after it.  Two ranks: both ranks hold the same parameters after
    7 steps, and the update they took is the one torch DDP's bucketed all-reduce leads to.  Not bit for bit between the two
    RUNS: MIOpen's weight-gradient kernels accumulate with atomics, so two trainings differ in the last bits of their
    gradients and RAdam's normalised step turns that into ~1e-2 of a step where a gradient is near zero; what a wrong
    reduction (SUM for AVG, a gradient left out, a stale replay buffer) would change is the direction and size of the
    update, and those are pinned."""
    out = _spawn(_w_flat_sync)
    for name in ("graph", "ddp"):
        assert out[0][name][1] == out[1][name][1], f"{name}: ranks out of sync"
        assert out[0][name][0] != out[1][name][0]                       # different data per rank
    pg, pd = torch.tensor(out[0]["graph"][1]), torch.tensor(out[0]["ddp"][1])
    p0g, p0d = torch.tensor(out[0]["graph"][2]), torch.tensor(out[0]["ddp"][2])
    assert torch.equal(p0g, p0d)                                            # the same calibrated, broadcast start
    dg, dd = (pg - p0g).double(), (pd - p0d).double()
    cos = float(dg @ dd / (dg.norm() * dd.norm()))
    ratio = float(dg.norm() / dd.norm())
    assert cos > 0.999 and 0.99 < ratio < 1.01, (cos, ratio)
    assert float((pg - pd).abs().max()) <= 0.05 * float(dd.abs().max())
    lg, ld = out[0]["graph"][0], out[0]["ddp"][0]
    assert all(abs(a - b) <= 1e-5 * abs(b) for a, b in zip(lg, ld)), (lg, ld)


def _w_flat_sync_refused(rank, world):
    import mhaq_amd as M
    from mhaq_amd import nets
    from mhaq_amd.qat import QATConfig, QATTrainer
    dev = "cuda:0"
    calib = torch.randn(4, 3, 32, 32).to(dev)
    cfg = QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.LSQ, act_bit=4, weight_bit=4,
                    excluded_layers=("features.init_block.conv", "output"))
    try:
        QATTrainer(nets.resnet20_cifar(10), cfg, dev, calib_batches=[calib], capture_graph=True)     # SyncBatchNorm inside
    except ValueError as e:
        msg = str(e)
    else:
        msg = ""
    tr = QATTrainer(nets.resnet20_cifar(10), cfg, dev, calib_batches=[calib], capture_graph="auto")
    # plain BatchNorm (sync_batchnorm off): no collective inside the step, but BUFFERS -- torch DDP re-broadcasts the
    # running statistics from rank 0 every forward, the flat form has no such exchange, so such a model keeps DDP too
    cfg_bn = QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.LSQ, act_bit=4, weight_bit=4,
                       excluded_layers=("features.init_block.conv", "output"), sync_batchnorm=False)
    tr_bn = QATTrainer(nets.resnet20_cifar(10), cfg_bn, dev, calib_batches=[calib], capture_graph="auto")
    bn = (tr_bn._flat_sync, bool(tr_bn.capture_graph), isinstance(tr_bn.module, torch.nn.parallel.DistributedDataParallel),
          any(isinstance(m, torch.nn.SyncBatchNorm) for m in tr_bn.net.modules()))
    return msg, tr._flat_sync, bool(tr.capture_graph), isinstance(tr.module, torch.nn.parallel.DistributedDataParallel), bn


def test_a_step_with_collectives_of_its_own_keeps_torch_ddp():
    out = _spawn(_w_flat_sync_refused)
    for r in (0, 1):
        msg, flat, capture, is_ddp, bn = out[r]
        assert "SyncBatchNorm" in msg
        assert not flat and not capture and is_ddp
        assert bn == (False, False, True, False)       # buffers without SyncBatchNorm: still torch DDP, never the flat form


def _w_auto_moves_to_ddp(rank, world):
    """capture_graph="auto" on a BatchNorm-free model under data parallelism settles with the flat all-reduce; a step that
    turns out GPU-bound (forced here: the host-share record is pre-filled) moves to torch DDP after the third step."""
    import mhaq_amd as M
    from mhaq_amd import nets, ops
    from mhaq_amd.qat import QATConfig, QATTrainer
    dev = "cuda:0"
    torch.manual_seed(1)
    ops.manual_seed(1)
    g = torch.Generator().manual_seed(70 + rank)
    x = (torch.rand(4, 3, 24, 24, generator=g) * 255.0).to(dev)
    y = (torch.rand(4, 3, 96, 96, generator=g) * 255.0).to(dev)
    calib = (torch.rand(4, 3, 24, 24, generator=torch.Generator().manual_seed(9)) * 255.0).to(dev)
    cfg = QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.LSQ, act_bit=4, weight_bit=4, distillation=False,
                    excluded_layers=("fea_conv", "upsampler.0"), learning_rate=5e-4, warmup=2, criterion=torch.nn.L1Loss())
    tr = QATTrainer(nets.rfdn(), cfg, dev, calib_batches=[calib], capture_graph="auto")
    assert tr._flat_sync and tr.capture_graph == "auto"
    tr._host_share = [0.0, 0.0]                   # "the host needed none of the step's wall time": GPU-bound
    states = []
    for _ in range(6):
        tr.train_step(x, y)
        states.append((tr._flat_sync, isinstance(tr.module, torch.nn.parallel.DistributedDataParallel), tr._graph is not None))
    flat = torch.cat([p.detach().flatten() for p in tr.net.parameters()]).double()
    return states, float(flat.sum()), float(flat.abs().sum())


def test_auto_mode_moves_a_gpu_bound_step_to_torch_ddp():
    out = _spawn(_w_auto_moves_to_ddp)
    for r in (0, 1):
        states = out[r][0]
        assert states[0] == (True, False, False) and states[1] == (True, False, False)      # settling with the flat all-reduce
        assert all(s == (False, True, False) for s in states[2:]), states                    # then torch DDP, never a graph
    assert abs(out[0][1] - out[1][1]) <= 1e-9 * out[0][2]                                     # ranks in sync throughout


def _rfdn_auto_trainer(rank, capture="auto", net_hook=None):
    import mhaq_amd as M
    from mhaq_amd import nets, ops
    from mhaq_amd.qat import QATConfig, QATTrainer
    dev = "cuda:0"
    torch.manual_seed(1)
    ops.manual_seed(1)
    g = torch.Generator().manual_seed(70 + rank)
    x = (torch.rand(4, 3, 24, 24, generator=g) * 255.0).to(dev)
    y = (torch.rand(4, 3, 96, 96, generator=g) * 255.0).to(dev)
    calib = (torch.rand(4, 3, 24, 24, generator=torch.Generator().manual_seed(9)) * 255.0).to(dev)
    cfg = QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.LSQ, act_bit=4, weight_bit=4, distillation=False,
                    excluded_layers=("fea_conv", "upsampler.0"), learning_rate=5e-4, warmup=2, criterion=torch.nn.L1Loss())
    net = nets.rfdn()
    if net_hook is not None:
        net_hook(net)
    tr = QATTrainer(net, cfg, dev, calib_batches=[calib], capture_graph=capture)
    assert tr._flat_sync and tr.capture_graph == capture
    return tr, x, y


def _six_steps(tr, x, y):
    import warnings
    states = []
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        for _ in range(6):
            tr.train_step(x, y)
            states.append((tr._flat_sync, isinstance(tr.module, torch.nn.parallel.DistributedDataParallel),
                           tr._graph is not None, tr.capture_graph))
    flat = torch.cat([p.detach().flatten() for p in tr.net.parameters()]).cpu()
    return states, flat.tolist(), [str(w.message)[:200] for w in caught if issubclass(w.category, RuntimeWarning)]


def _w_ranks_disagree_on_gpu_bound(rank, world):
    tr, x, y = _rfdn_auto_trainer(rank)
    tr._step_is_gpu_bound = lambda: rank == 0        # rank 0 measures "GPU-bound", rank 1 "host-bound"
    return _six_steps(tr, x, y)


def test_ranks_on_either_side_of_the_gpu_bound_threshold_take_one_decision():
    """Round 3's auto mode decided per rank from a local wall clock: a rank below the 0.8 threshold built torch DDP while its
    peer captured and issued one flat all-reduce -- mismatched collectives, a hang.  The verdicts are all-reduced (MAX) at
    the third settling step: one GPU-bound rank moves the whole job to DDP, and both ranks stay in step."""
    out = _spawn(_w_ranks_disagree_on_gpu_bound)
    for r in (0, 1):
        states = out[r][0]
        assert states[0][:3] == (True, False, False) and states[1][:3] == (True, False, False), states
        assert all(s == (False, True, False, False) for s in states[2:]), states        # both ranks: torch DDP, no graph
    assert out[0][1] == out[1][1]                                                       # identical parameters


def _w_capture_fails_on_one_rank(rank, world):
    tr, x, y = _rfdn_auto_trainer(rank, capture=True)
    if rank == 1:
        inner = tr._forward_backward

        def failing(*a):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("descriptor pool ran dry (injected)")
            return inner(*a)
        tr._forward_backward = failing
    return _six_steps(tr, x, y)


def test_a_capture_that_fails_on_one_rank_takes_every_rank_to_the_eager_loop():
    out = _spawn(_w_capture_fails_on_one_rank)
    for r in (0, 1):
        states, _, warned = out[r]
        assert all(s[:3] == (True, False, False) for s in states), states               # flat all-reduce, never a graph
        assert states[2][3] is True and all(s[3] is False for s in states[3:]), states   # capture given up at step 4, together
        assert any("capture of the training step failed" in m for m in warned), warned
    assert "injected" in " ".join(out[1][2]) and "another rank" in " ".join(out[0][2])
    assert out[0][1] == out[1][1]


def _w_host_sync_on_one_rank(rank, world):
    class SyncsOnRankOne(torch.nn.Module):
        def forward(self, x):
            if rank == 1 and float(x.abs().max()) > 1e30:      # a host sync in forward, on one rank only
                raise RuntimeError("overflow")
            return x

    def hook(net):
        net.add_module("guard", SyncsOnRankOne())
        inner = net.forward
        net.forward = lambda x: net.guard(inner(x))
    tr, x, y = _rfdn_auto_trainer(rank, net_hook=hook)
    tr._step_is_gpu_bound = lambda: False
    return _six_steps(tr, x, y)


def test_a_host_sync_seen_by_one_rank_keeps_every_rank_eager():
    out = _spawn(_w_host_sync_on_one_rank)
    for r in (0, 1):
        states, _, warned = out[r]
        assert all(s[:3] == (True, False, False) for s in states), states
        assert all(s[3] is False for s in states[2:]), states
        assert any("synchronises with the host" in m for m in warned), warned
    assert "another rank" in " ".join(out[0][2])
    assert out[0][1] == out[1][1]
