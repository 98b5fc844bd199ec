"""CPU, world_size 2 over gloo: the N>1 path of the QAT loop (mhaq_amd/qat.py) -- DDP with
find_unused_parameters, per-rank data shards, the AEWGS statistics all-reduce inside backward
(gdnsq.py:126-129, packed into one message in mhaq_amd/ops.py) -- using the oracle's CPU layers
in place of the HIP ones (the trainer takes the layer classes as an argument)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
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
    [p.join(180) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return dict(ret)


# ------------------------------------------------------------------ workers (module level: picklable)
def _w_allreduce_avg(rank, world):
    from mhaq_amd import ops
    t = torch.full((3, 4), float(rank + 1))
    ops._allreduce_avg_(t)
    seed_a = ops.rng.next()
    return t.tolist(), seed_a


def _w_aewgs_stats(rank, world):
    """AEWGS weight backward under DDP: every rank holds the same weight, different upstream grads;
    the all-reduced statistics must equal the mean of the per-rank statistics."""
    from oracle import fq_eager as O
    torch.manual_seed(0)
    w = torch.randn(6, 4, 3, 3) * 0.2
    ls = torch.full((6, 1, 1, 1), -4.0)
    g = torch.Generator().manual_seed(100 + rank)
    G = torch.randn(6, 4, 3, 3, generator=g)
    r = torch.full_like(w, 0.5)
    ws = w.clone().requires_grad_(True)
    wq, _, _ = O.weight_fake_quant(ws, ls, True, "AEWGS", r=r)
    wq.backward(G)
    return ws.grad.tolist(), G.tolist()


def _w_trainer(rank, world):
    from mhaq_amd import nets
    from mhaq_amd.enums import QNMethod, QScheme
    from mhaq_amd.qat import QATConfig, QATTrainer
    from oracle.loss import LOSS_CLASSES
    from oracle.ref_layers import ORACLE_LAYERS
    torch.manual_seed(1)                       # same initial weights on every rank
    net = nets.resnet20_cifar(10)
    cfg = QATConfig(qscheme=QScheme.PER_CHANNEL, qnmethod=QNMethod.AEWGS, act_bit=4, weight_bit=4,
                    excluded_layers=("features.init_block.conv", "output"), warmup=2)
    g = torch.Generator().manual_seed(50 + rank)     # different data per rank
    x = torch.randn(4, 3, 32, 32, generator=g)
    y = torch.randint(0, 10, (4,), generator=g)
    calib = torch.randn(4, 3, 32, 32, generator=torch.Generator().manual_seed(9))
    tr = QATTrainer(net, cfg, "cpu", calib_batches=[calib], layers=ORACLE_LAYERS, loss_classes=LOSS_CLASSES,
                    minmax_fn=lambda t: torch.stack(list(t.aminmax())))
    assert tr.distributed
    losses = [float(tr.train_step(x, y)) for _ in range(3)]
    flat = torch.cat([p.detach().flatten() for p in tr.net.parameters()])
    return losses, float(flat.double().sum()), float(flat.double().abs().sum()), tr.schedule.total_batch


def _w_trainer_quant_bias(rank, world):
    """quantize_bias=True under DDP: log_b_s never receives a gradient (Q_b.scale is overwritten every forward,
    gdnsq_conv2d.py:86-88), so the trainer must keep it out of the reducer or step 2 raises."""
    from mhaq_amd.enums import QNMethod, QScheme
    from mhaq_amd.qat import QATConfig, QATTrainer
    from oracle.loss import LOSS_CLASSES
    from oracle.ref_layers import ORACLE_LAYERS
    torch.manual_seed(2)
    nn = torch.nn
    net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1, bias=True), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1, bias=True),
                        nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(8, 5))
    cfg = QATConfig(qscheme=QScheme.PER_CHANNEL, qnmethod=QNMethod.AEWGS, act_bit=4, weight_bit=4,
                    excluded_layers=("5",), quantize_bias=True, distillation=False, warmup=1)
    g = torch.Generator().manual_seed(60 + rank)
    x = torch.randn(4, 3, 12, 12, generator=g)
    y = torch.randint(0, 5, (4,), generator=g)
    tr = QATTrainer(net, cfg, "cpu", calib_batches=[x], layers=ORACLE_LAYERS, loss_classes=LOSS_CLASSES,
                    minmax_fn=lambda t: torch.stack(list(t.aminmax())))
    assert tr.distributed and all(m.quant_bias for m in tr.net.modules() if hasattr(m, "log_b_s"))
    assert all(not m.log_b_s.requires_grad for m in tr.net.modules() if hasattr(m, "log_b_s"))
    losses = [float(tr.train_step(x, y)) for _ in range(3)]        # step 2 is where an unused parameter raises
    flat = torch.cat([p.detach().flatten() for p in tr.net.parameters()])
    return losses, float(flat.double().sum()), float(flat.double().abs().sum())


# ------------------------------------------------------------------ tests
def test_allreduce_avg_gloo_and_rank_seeds():
    out = _spawn(_w_allreduce_avg)
    for r in (0, 1):
        assert out[r][0] == [[1.5] * 4] * 3           # mean of 1 and 2 on both ranks
    assert out[0][1][0] != out[1][1][0]               # ranks draw different sign streams
    assert out[0][1][1] == out[1][1][1] == 1          # same call counter


def test_aewgs_statistics_are_averaged_across_ranks():
    from oracle import fq_closed_form as CF
    out = _spawn(_w_aewgs_stats)
    torch.manual_seed(0)
    w = torch.randn(6, 4, 3, 3) * 0.2
    s = torch.exp2(torch.full((6,), -4.0))
    Gs = [torch.tensor(out[r][1]) for r in (0, 1)]
    # expected: per-rank statistics averaged (e2, me are rank-invariant; num differs with sign(G))
    zp = w.amin((1, 2, 3), keepdim=True)
    v = (w - zp) / s.reshape(6, 1, 1, 1)
    e = torch.round(v) - v
    num = sum(((G * s.reshape(6, 1, 1, 1)).sign() * e).mean((1, 2, 3)) for G in Gs) / 2
    stats = (num, e.square().mean((1, 2, 3)), e.mean((1, 2, 3)))
    for r in (0, 1):
        cf = CF.per_channel(w, Gs[r], torch.full_like(w, 0.5), s, "AEWGS", stats=stats)
        assert torch.allclose(torch.tensor(out[r][0]), cf["gw"], rtol=1e-5, atol=1e-6)


def test_ddp_qat_trainer_two_ranks_stay_in_sync():
    out = _spawn(_w_trainer)
    (l0, s0, a0, n0), (l1, s1, a1, n1) = out[0], out[1]
    assert n0 == n1 == 3
    assert all(torch.isfinite(torch.tensor(l0))) and all(torch.isfinite(torch.tensor(l1)))
    assert l0 != l1                                   # different shards -> different local losses
    assert abs(s0 - s1) <= 1e-6 * a0 and abs(a0 - a1) <= 1e-6 * a0   # identical parameters after 3 steps


def test_ddp_trainer_with_quantized_bias_two_ranks():
    out = _spawn(_w_trainer_quant_bias)
    (l0, s0, a0), (l1, s1, a1) = out[0], out[1]
    assert all(torch.isfinite(torch.tensor(l0 + l1)))
    assert abs(s0 - s1) <= 1e-6 * a0 and abs(a0 - a1) <= 1e-6 * a0


# ------------------------------------------------------------------ per-rank sign streams (SURVEY.md 8e determinism note)
def _w_rank_seeds(rank, world):
    from mhaq_amd import ops
    ops.manual_seed(2024)                      # the SAME user seed on every rank, as a training script would set it
    a = ops.rng.next()
    b = ops.rng.next()
    return a, b


def test_ranks_draw_different_and_independent_sign_streams():
    """The reference draws randint_like on each rank's own generator, for weights too (gdnsq.py:54): ranks must not
    share a sign stream.  Same user seed on both ranks -> different Philox keys (seed ^ rank * golden ratio), the same
    offset sequence; the two ranks' streams for a weight tensor agree on ~half of the elements (independent fair
    coins), and so do consecutive offsets of one rank."""
    import numpy as np
    from tests.philox_ref import signs
    out = _spawn(_w_rank_seeds)
    (s0, o0), (s0b, o0b) = out[0]
    (s1, o1), _ = out[1]
    assert s0 != s1 and s0 == s0b                         # per-rank key, constant within a rank
    assert (o0, o0b) == (1, 2) and o1 == 1                # every backward call takes the next offset
    assert s0 == 2024 and s1 == 2024 ^ 0x9E3779B97F4A7C15
    n = 1 << 16
    r0, r1, r0b = signs(n, s0, o0), signs(n, s1, o1), signs(n, s0, o0b)
    for a, b in ((r0, r1), (r0, r0b)):
        agree = float(np.mean(a == b))
        assert abs(agree - 0.5) < 4 * 0.5 / np.sqrt(n), agree      # 4 sigma of a fair coin
    for r in (r0, r1):
        assert abs(float(np.mean(r.astype(np.float64)))) < 4 / np.sqrt(n)
