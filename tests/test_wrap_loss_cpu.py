"""CPU: host-side logic of the product -- the wrapping rule (mhaq_amd/wrap.py), regulariser gathering and
PotentialLoss (mhaq_amd/loss.py), TemperatureSchedule (mhaq_amd/qat.py) -- with the oracle's CPU layers
plugged in, against the golden vectors recorded from the reference's ModelHelper / PotentialLoss
(tests/golden/model_cases.npz) and the reference's documented wrapping outcomes."""
import numpy as np
import pytest
import torch

from mhaq_amd import nets, wrap
from mhaq_amd.loss import SymmetricalKL
from oracle.loss import LOSS_CLASSES, PotentialLoss, PotentialLossNoPred
from mhaq_amd.qat import TemperatureSchedule
from oracle import ref_layers as RL
from oracle.ref_layers import ORACLE_LAYERS
from tests.golden_util import T, load_cases

MODEL = load_cases("model_cases.npz")


def _toy(c, pc):
    qs = 1 if pc else 0
    net = torch.nn.Sequential(
        RL.NoisyAct(signed=True), RL.NoisyConv2d(3, 6, 3, padding=1, qscheme=qs, qnmethod="LSQ"),
        torch.nn.ReLU(),
        RL.NoisyAct(signed=False), RL.NoisyConv2d(6, 4, 3, padding=1, qscheme=qs, qnmethod="LSQ"))
    convs = [m for m in net if isinstance(m, RL.NoisyConv2d)]
    acts = [m for m in net if isinstance(m, RL.NoisyAct)]
    with torch.no_grad():
        for i, m in enumerate(convs):
            m.weight.copy_(T(c[f"w{i}"]))
            m.log_wght_s.copy_(T(c[f"log_wght_s{i}"]).view_as(m.log_wght_s))
        for i, a in enumerate(acts):
            a.log_act_s.copy_(T(c[f"log_act_s{i}"]))
            a.log_act_q.copy_(T(c[f"log_act_q{i}"]))
    return net, convs, acts


@pytest.mark.parametrize("name", sorted(MODEL))
def test_get_model_values_and_potential_loss_match_reference(name):
    c = MODEL[name]
    pc = bool(c["per_channel"])
    net, convs, acts = _toy(c, pc)
    las, laq, lws, lwq = wrap.get_model_values(net, 1 if pc else 0)
    for got, key in ((las, "las"), (laq, "laq"), (lws, "lws"), (lwq, "lwq")):
        assert np.array_equal(got.detach().numpy(), c[key]), key
    if name.endswith("nopred"):
        L = PotentialLossNoPred(None, p=1, a=int(c["a_bits"]), w=int(c["w_bits"]))
        L.t, L.loss_sum, L.cnt = float(c["t"]), torch.tensor(float(c["loss_sum"])), int(c["cnt"])
        base = torch.tensor(float(c["base"]), requires_grad=True) * 1.0
        ploss = L((base, las, laq, lws, lwq))
    else:
        L = PotentialLoss(torch.nn.MSELoss(), p=1, a=int(c["a_bits"]), w=int(c["w_bits"]))
        L.t, L.loss_sum, L.cnt = float(c["t"]), torch.tensor(float(c["loss_sum"])), int(c["cnt"])
        prd = torch.linspace(-1, 1, 12).view(3, 4).requires_grad_(True)
        tgt = torch.linspace(1, -1, 12).view(3, 4) * 0.5
        ploss = L((prd, las, laq, lws, lwq), tgt)
    assert np.allclose(ploss.detach().numpy(), c["ploss"], rtol=1e-6, atol=1e-7)
    assert L.cnt == int(c["cnt"]) + 1                     # running state advanced in training mode
    ploss.backward()
    for i, m in enumerate(convs):
        assert np.allclose(m.weight.grad.numpy(), c[f"gw{i}"], rtol=1e-6, atol=1e-7)
        assert np.allclose(m.log_wght_s.grad.numpy(), c[f"g_log_wght_s{i}"].reshape(m.log_wght_s.shape), rtol=1e-6, atol=1e-7)
    for i, a in enumerate(acts):
        assert np.allclose(a.log_act_s.grad.numpy(), c[f"g_log_act_s{i}"], rtol=1e-6, atol=1e-7)
        assert np.allclose(a.log_act_q.grad.numpy(), c[f"g_log_act_q{i}"], rtol=1e-6, atol=1e-7)


def test_wrapping_rule_on_the_three_baseline_nets():
    r18 = nets.resnet18(10)
    wrap.quantize_model(r18, 1, "AEWGS", ("conv1", "fc"), layers=ORACLE_LAYERS)
    acts = [(n, m.signed) for n, m in r18.named_modules() if isinstance(m, RL.NoisyAct)]
    assert len(acts) == 16 and all(s == (".conv1." in n) for n, s in acts)
    assert sum(m.weight.numel() for m in r18.modules() if isinstance(m, RL.NoisyConv2d)) == 10_985_472
    assert sum(m.weight.shape[0] for m in r18.modules() if isinstance(m, RL.NoisyConv2d)) == 3840
    # shared, not copied, parameters (gdnsq_quant.py:492-495)
    r20 = nets.resnet20_cifar(100)
    w_before = r20.features.stage1.unit1.body.conv1.conv.weight
    wrap.quantize_model(r20, 0, "STE", ("features.init_block.conv", "output"), layers=ORACLE_LAYERS)
    assert r20.features.stage1.unit1.body.conv1.conv[1].weight is w_before
    acts = [m for m in r20.modules() if isinstance(m, RL.NoisyAct)]
    assert len(acts) == 18 and all(a.signed for a in acts)
    assert sum(m.weight.numel() for m in r20.modules() if isinstance(m, RL.NoisyConv2d)) == 267_264
    assert isinstance(r20.features.stage2.unit1.identity_conv.conv, torch.nn.Conv2d)      # 1x1 skipped
    rf = nets.rfdn()
    wrap.quantize_model(rf, 1, "LSQ", ("fea_conv", "upsampler.0"), layers=ORACLE_LAYERS)
    acts = [m for m in rf.modules() if isinstance(m, RL.NoisyAct)]
    assert len(acts) == 33 and all(a.signed for a in acts)
    assert sum(m.weight.numel() for m in rf.modules() if isinstance(m, RL.NoisyConv2d)) == 358_236
    with pytest.raises(AttributeError):
        wrap.quantize_model(nets.rfdn(), 1, "LSQ", ("not_a_layer",), layers=ORACLE_LAYERS)
    # act_bit == -1 disables the activation quantizers (gdnsq_quant.py:502)
    r = nets.resnet20_cifar(10)
    wrap.quantize_model(r, 0, "STE", ("features.init_block.conv", "output"), act_bit=-1, layers=ORACLE_LAYERS)
    assert all(m.disable for m in r.modules() if isinstance(m, RL.NoisyAct))


def test_temperature_schedule_matches_callback():
    # temperature_adjust.py:36-54 with lr=3e-4, warmup=3, scale_t=2
    sch = TemperatureSchedule(3e-4, warmup=3, scale_lr=1.0, scale_t=2.0)
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=3e-4)
    crit = PotentialLossNoPred(None)
    sch.start(opt)            # on_train_start: change_lr(..., 0) -- the first optimizer step runs at rate 0
    assert opt.param_groups[0]["lr"] == 0.0
    lrs = [sch.step(crit, opt) for _ in range(5)]
    assert np.allclose(lrs[:3], [1e-4, 2e-4, 3e-4]) and np.allclose(lrs[3:], [3e-4, 3e-4])
    assert abs(crit.t - 2 * 3e-4 * 2.0) < 1e-12 and opt.param_groups[0]["lr"] == lrs[-1]


def test_symmetrical_kl():
    a, b = torch.randn(5, 7), torch.randn(5, 7)
    la, lb = a.log_softmax(1), b.log_softmax(1)
    want = ((lb.exp() * (lb - la)).sum() + (la.exp() * (la - lb)).sum()) / 5
    assert torch.allclose(SymmetricalKL()(a, b), want, atol=1e-6)


def test_trainer_first_step_runs_at_rate_zero_like_on_train_start():
    """temperature_adjust.py:28-33: the reference's first optimizer step has lr == 0 (ramp k/warmup after it)."""
    from mhaq_amd.qat import QATConfig, QATTrainer
    torch.manual_seed(0)
    net = nets.resnet20_cifar(10)
    cfg = QATConfig(qscheme=1, qnmethod="LSQ", act_bit=4, weight_bit=4, distillation=False, warmup=4,
                    excluded_layers=("features.init_block.conv", "output"))
    x, y = torch.randn(2, 3, 32, 32), torch.randint(0, 10, (2,))
    tr = QATTrainer(net, cfg, "cpu", calib_batches=[x], layers=ORACLE_LAYERS, loss_classes=LOSS_CLASSES, distributed=False,
                    minmax_fn=lambda t: torch.stack(list(t.aminmax())))
    assert all(g["lr"] == 0.0 for g in tr.optimizer.param_groups)
    before = torch.cat([p.detach().flatten().clone() for p in tr.net.parameters()])
    tr.train_step(x, y)
    after = torch.cat([p.detach().flatten() for p in tr.net.parameters()])
    assert torch.equal(before, after)                       # lr 0: step 1 moves nothing
    assert np.isclose(tr.optimizer.param_groups[0]["lr"], cfg.learning_rate * 1 / 4)
    tr.train_step(x, y)
    assert not torch.equal(before, torch.cat([p.detach().flatten() for p in tr.net.parameters()]))


def test_multi_tensor_weights_refused_for_data_parallel_trainer():
    from mhaq_amd.qat import QATConfig, QATTrainer
    cfg = QATConfig(distillation=False, excluded_layers=("features.init_block.conv", "output"))
    with pytest.raises(ValueError, match="single-GPU"):
        QATTrainer(nets.resnet20_cifar(10), cfg, "cpu", layers=ORACLE_LAYERS, loss_classes=LOSS_CLASSES, distributed=True,
                   multi_tensor_weights=True)


def test_fuse_and_freeze_batchnorm_switches_of_the_wrapping_rule():
    """config.quantization.fuse_batchnorm / freeze_batchnorm (gdnsq_quant.py:129-190): the BatchNorm that FOLLOWS a
    wrapped conv is folded into it before the layer is replaced (the quantizer then sees the folded weight, the
    BatchNorm becomes Identity); freezing puts every remaining BatchNorm in eval mode without gradients."""
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3, padding=1, bias=False), torch.nn.BatchNorm2d(5), torch.nn.ReLU(),
                              torch.nn.Conv2d(5, 4, 1), torch.nn.BatchNorm2d(4))
    with torch.no_grad():
        for m in net:
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(), m.running_var.uniform_(0.5, 2), m.weight.normal_(), m.bias.normal_()
    x = torch.randn(2, 3, 6, 6)
    want = net.eval()(x)
    w0 = net[0].weight.detach().clone()
    scale = (net[1].weight / torch.sqrt(net[1].running_var + net[1].eps)).detach()
    wrap.quantize_model(net, 1, "LSQ", (), layers=ORACLE_LAYERS, fuse_batchnorm=True, freeze_batchnorm=True)
    assert isinstance(net[1], torch.nn.Identity) and isinstance(net[4], torch.nn.BatchNorm2d)   # 1x1 conv: not wrapped
    q = net[0][1]
    assert isinstance(q, RL.NoisyConv2d) and q.bias is not None
    assert torch.allclose(q.weight, w0 * scale.view(-1, 1, 1, 1))
    assert not net[4].training and not net[4].weight.requires_grad and not net[4].bias.requires_grad
    net[0][0].disable = True                                  # activation quantizer off, weight grid fine: function kept
    with torch.no_grad():
        q.log_wght_s.fill_(-20.0)
    got = net.eval()(x)
    assert torch.allclose(got, want, atol=1e-4)
