"""GPU (-m gpu): the device work of a QAT step captured in a hipGraph (QATTrainer(capture_graph=True)): teacher and
student forward, fused PotentialLoss and backward replayed as one graph launch, the optimizer stepping eagerly on
the graph's static gradients.  Everything a replay must see
fresh lives on the device: the loss state {loss_sum, cnt, t}, the learning rate, and the offset of the random sign
streams (a captured launch's host (seed, offset) is frozen; the backward kernels add a device-resident uint64
-- `offset_dev`, include/mhaq_fq.h -- that the captured step advances at its end)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _restore_rng_mode():
    from mhaq_amd import ops
    det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True      # MIOpen's default wrw kernels use atomics: not run-to-run exact
    yield
    torch.backends.cudnn.deterministic = det
    assert ops.rng.offset_base is None          # a capturing trainer scopes its device offset to its own steps


def _make(capture, distillation, act_method):
    import mhaq_amd as M
    from mhaq_amd import nets, ops
    from mhaq_amd.qat import QATConfig, QATTrainer
    torch.manual_seed(5)
    ops.manual_seed(5)
    cfg = QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.LSQ, act_bit=4, weight_bit=4,
                    excluded_layers=("features.init_block.conv", "output"), warmup=3, distillation=distillation,
                    learning_rate=1e-3)
    g = torch.Generator().manual_seed(2)
    calib = torch.randn(8, 3, 32, 32, generator=g).to(DEV)
    factory = lambda params, lr: torch.optim.RAdam(   # noqa: E731  -- the same capturable optimizer on both sides
        params, torch.tensor(float(lr), device=DEV), capturable=True)
    tr = QATTrainer(nets.resnet20_cifar(10), cfg, DEV, calib_batches=[calib], distributed=False,
                    optimizer_factory=factory, capture_graph=capture)
    for m in tr.net.modules():
        if hasattr(m, "log_act_s"):
            m.Q.qnmethod = M.QNMethod[act_method]
    return tr


def _set_weight_method(tr, name):
    import mhaq_amd as M
    for m in tr.net.modules():
        if hasattr(m, "log_wght_s"):
            m.Q.qnmethod = M.QNMethod[name]


@pytest.mark.parametrize("distillation,act_method,w_method", [(False, "LSQ", "LSQ"), (True, "LSQ", "LSQ"),
                                                               (True, "STE", "AEWGS"), (False, "STE", "STE")])
def test_graphed_steps_equal_eager_steps_bit_for_bit(distillation, act_method, w_method):
    """3 eager + 5 replayed steps leave exactly the parameters, loss values and loss state that 8 eager steps
    leave -- with the random estimators too: replay k adds k * stride to the frozen host offsets through the
    device-resident word, i.e. draws the sign streams eager step k draws."""
    gen = torch.Generator().manual_seed(9)
    batches = [(torch.randn(8, 3, 32, 32, generator=gen).to(DEV), torch.randint(0, 10, (8,), generator=gen).to(DEV))
               for _ in range(8)]
    eager = _make(False, distillation, act_method)          # (re)seeds the sign stream
    _set_weight_method(eager, w_method)
    le = [float(eager.train_step(x, y)) for x, y in batches]
    graphed = _make(True, distillation, act_method)
    _set_weight_method(graphed, w_method)
    lg = [float(graphed.train_step(x, y)) for x, y in batches]
    assert graphed._graph is not None and graphed._eager_steps == 3
    if act_method != "LSQ":
        assert graphed._rng_stride > 0 and int(graphed._rng_base) == 5 * graphed._rng_stride
    assert le == lg
    for (n, a), (_, b) in zip(eager.net.named_parameters(), graphed.net.named_parameters()):
        assert torch.equal(a, b), n
    assert eager.loss.cnt == graphed.loss.cnt == 9                    # 1 + 8 training steps, counted on the device
    assert float(eager.loss.loss_sum) == float(graphed.loss.loss_sum)
    assert eager.loss.t == graphed.loss.t > 0                         # warm-up of 3 steps is over: t ramps
    assert float(graphed.loss._state[2]) == pytest.approx(graphed.loss.t)
    lr_e = float(eager.optimizer.param_groups[0]["lr"])
    assert float(graphed.optimizer.param_groups[0]["lr"]) == lr_e


def test_replay_k_uses_the_sign_stream_of_offset0_plus_k():
    """STE activation backward under capture with a device-resident offset word: replay k must equal an eager
    backward given the materialised stream fill_r(seed, offset0 + k) -- and the deterministic gradients repeat."""
    from mhaq_amd import ops
    ops.manual_seed(77)
    x = (torch.randn(4, 8, 12, 12, device=DEV) * 2)
    g = torch.randn_like(x)
    ls = torch.tensor([-3.0], device=DEV, requires_grad=True)
    lq = torch.tensor([1.0], device=DEV, requires_grad=True)
    b = torch.tensor([-1.0], device=DEV, requires_grad=True)
    base = torch.zeros(1, dtype=torch.int64, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), ops.rng.device_offset(base):
        for _ in range(2):
            y, _ = ops.fake_quant_act_layer(x, ls, lq, b, "STE")
            y.backward(g)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ls.grad = lq.grad = b.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side), ops.rng.device_offset(base):
        y, _ = ops.fake_quant_act_layer(x, ls, lq, b, "STE")
        y.backward(g)
        base.add_(1)
    seed, offset0 = ops.rng.next()[0], ops.rng.drawn() - 1      # the captured launch holds host offset `offset0`
    assert ops.rng.offset_base is None
    got = []
    for k in range(4):
        graph.replay()
        got.append((float(ls.grad), float(lq.grad), float(b.grad)))
    assert int(base) == 4
    for k in range(4):
        l2, q2, b2 = (t.detach().clone().requires_grad_(True) for t in (ls, lq, b))
        r8 = ops.fill_r(x.numel(), seed, offset0 + k, DEV)
        y2, _ = ops.fake_quant_act_layer(x, l2, q2, b2, "STE", r_sign=r8)
        y2.backward(g)
        assert got[k] == (float(l2.grad), float(q2.grad), float(b2.grad)), k
    assert len({v[0] for v in got}) == 4                  # fresh signs per replay ...
    assert len({v[1:] for v in got}) == 1                 # ... the deterministic gradients repeat exactly


def test_graphed_trainer_runs_the_default_estimators():
    tr = _make(True, True, "STE")
    import mhaq_amd as M
    for m in tr.net.modules():
        if hasattr(m, "log_wght_s"):
            m.Q.qnmethod = M.QNMethod.AEWGS
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(8, 3, 32, 32, generator=gen).to(DEV)
    y = torch.randint(0, 10, (8,), generator=gen).to(DEV)
    before = copy.deepcopy([p.detach().clone() for p in tr.net.parameters()])
    losses = [float(tr.train_step(x, y)) for _ in range(7)]
    assert all(v == v for v in losses) and tr._graph is not None
    assert any(not torch.equal(a, b) for a, b in zip(before, tr.net.parameters()))
    with pytest.raises(ValueError):
        from mhaq_amd import nets
        from mhaq_amd.qat import QATConfig, QATTrainer
        QATTrainer(nets.resnet20_cifar(10), QATConfig(excluded_layers=("features.init_block.conv", "output")), DEV,
                   multi_tensor_weights=True, capture_graph=True)


def test_sign_tensor_codings_are_equivalent():
    """r_sign: a positive value is +0.5, zero or negative -0.5 -> the +-1 coding of mhaq_fq_fill_r and the 0/1
    coding of torch.randint(0, 2) give the same backward (per-tensor and per-channel kernels)."""
    from mhaq_amd import ops
    gen = torch.Generator().manual_seed(1)
    x = (torch.randn(5, 7, 9, generator=gen) * 2).to(DEV)
    g = torch.randn(5, 7, 9, generator=gen).to(DEV)
    bits = torch.randint(0, 2, (5, 7, 9), generator=gen, dtype=torch.int8).to(DEV)
    res = []
    for r8 in (bits, bits * 2 - 1, bits * 77 - 5 * (1 - bits)):
        ls, lq, b = (torch.tensor([v], device=DEV, requires_grad=True) for v in (-3.0, 1.0, -1.0))
        xr = x.clone().requires_grad_(True)
        y, _ = ops.fake_quant_act_layer(xr, ls, lq, b, "STE", r_sign=r8.contiguous())
        y.backward(g)
        w = x.reshape(5, 63).clone().requires_grad_(True)
        lws = torch.full((5, 1), -4.0, device=DEV, requires_grad=True)
        wq, _, _, _ = ops.fake_quant_weight_layer(w, lws, "AEWGS", r_sign=r8.reshape(5, 63).contiguous())
        wq.backward(g.reshape(5, 63))
        res.append((ls.grad.clone(), lq.grad.clone(), b.grad.clone(), xr.grad.clone(), w.grad.clone(), lws.grad.clone()))
    for other in res[1:]:
        assert all(torch.equal(a, b) for a, b in zip(res[0], other))


def test_capture_graph_auto_captures_a_host_bound_step_and_matches_eager():
    """capture_graph="auto": a ResNet-20 step at batch 8 is all host time -> the trainer captures after the settling
    steps; results equal the eager trainer's bit for bit (LSQ)."""
    gen = torch.Generator().manual_seed(9)
    batches = [(torch.randn(8, 3, 32, 32, generator=gen).to(DEV), torch.randint(0, 10, (8,), generator=gen).to(DEV))
               for _ in range(6)]
    eager = _make(False, True, "LSQ")
    le = [float(eager.train_step(x, y)) for x, y in batches]
    auto = _make("auto", True, "LSQ")
    la = [float(auto.train_step(x, y)) for x, y in batches]
    assert auto._graph is not None and auto.capture_graph == "auto" and len(auto._host_share) == 3
    assert le == la
    for (n, a), (_, b) in zip(eager.net.named_parameters(), auto.net.named_parameters()):
        assert torch.equal(a, b), n


@pytest.mark.parametrize("act_method,w_method", [("LSQ", "LSQ"), ("STE", "AEWGS")])
def test_a_batch_of_another_shape_runs_eagerly_and_the_graph_survives(act_method, w_method):
    """Graph mode with a smaller last batch in the middle: that step runs eagerly (same stream, same sign-stream
    bookkeeping), the following replays read the graph's own gradient tensors again; results equal the eager
    trainer's on the same sequence bit for bit -- with the random estimators too: the eager stand-in for replay R draws
    the streams replay R would have drawn and moves the device word on, so no two steps share a sign stream (round 2
    re-used the streams of the odd-shaped step two replays later; only LSQ, which draws none, was tested)."""
    gen = torch.Generator().manual_seed(13)
    sizes = [8, 8, 8, 8, 8, 4, 8, 8, 3, 8]
    batches = [(torch.randn(n, 3, 32, 32, generator=gen).to(DEV), torch.randint(0, 10, (n,), generator=gen).to(DEV))
               for n in sizes]
    eager = _make(False, True, act_method)
    _set_weight_method(eager, w_method)
    le = [float(eager.train_step(x, y)) for x, y in batches]
    graphed = _make(True, True, act_method)
    _set_weight_method(graphed, w_method)
    lg = [float(graphed.train_step(x, y)) for x, y in batches]
    assert graphed._graph is not None
    assert le == lg
    for (n, a), (_, b) in zip(eager.net.named_parameters(), graphed.net.named_parameters()):
        assert torch.equal(a, b), n


_FALLBACK_SCRIPT = r"""
import sys, warnings
sys.path.insert(0, {root!r})
import torch
import mhaq_amd as M
from mhaq_amd import nets, ops
from mhaq_amd.qat import QATConfig, QATTrainer

class Syncing(torch.nn.Module):        # a user layer with a host sync in its forward: not capturable
    def forward(self, x):
        if float(x.abs().max()) > 1e30:
            raise RuntimeError("overflow")
        return x

torch.manual_seed(5); ops.manual_seed(5)
net = nets.resnet20_cifar(10)
net.features.add_module("guard", Syncing())
cfg = QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.LSQ, act_bit=4, weight_bit=4,
                excluded_layers=("features.init_block.conv", "output"), warmup=3, distillation=True, learning_rate=1e-3)
g = torch.Generator().manual_seed(2)
dev = "cuda:0"
calib = torch.randn(8, 3, 32, 32, generator=g).to(dev)
tr = QATTrainer(net, cfg, dev, calib_batches=[calib], distributed=False, capture_graph={mode!r})
x = torch.randn(8, 3, 32, 32, generator=g).to(dev); y = torch.randint(0, 10, (8,), generator=g).to(dev)
losses = []
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    for _ in range(8):
        losses.append(float(tr.train_step(x, y)))
torch.cuda.synchronize()
assert all(v == v for v in losses), losses
assert tr._graph is None and tr.capture_graph is False, (tr._graph, tr.capture_graph)
assert any("captured" in str(m.message) or "capture" in str(m.message) for m in w), [str(m.message) for m in w]
assert losses[-1] != losses[0]
st = tr.act_hub.state()
assert st["captured_tables"] == 0 and st["retired"] == 0
print("FALLBACK_OK", losses[0], losses[-1])
"""


@pytest.mark.parametrize("mode", [True, "auto"])
def test_a_step_that_cannot_be_captured_falls_back_to_the_eager_loop(mode):
    """capture_graph="auto" is the default of the stock trainer: a model with a host sync in its forward (a user layer
    calling .item()) must keep training when the capture fails -- the graph is dropped with a warning, the trainer
    carries on eagerly on its settling stream and nothing stays pinned for a graph that does not exist.  In a child
    process under a time limit: a failed capture is exactly the kind of thing that must not take the test run down."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, "-c", _FALLBACK_SCRIPT.format(root=root, mode=mode)], capture_output=True,
                          text=True, timeout=300)
    assert proc.returncode == 0 and "FALLBACK_OK" in proc.stdout, proc.stdout[-2000:] + proc.stderr[-3000:]
