"""GPU (-m gpu): calibration on the HIP path.  ops.minmax / ops.row_minmax (the fused
min/max sweeps of mhaq_fq_minmax / mhaq_fq_row_minmax) against torch; calibrate_activations / calibrate_weights
over the HIP layers against the vectors recorded from the reference (tests/golden/calib_cases.npz) and, at
BASELINE sizes, against the oracle's restatement of gdnsq/calib/minmaxobserver.py:39-88."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import calib as OC  # noqa: E402
from tests.calib_util import Feed, assert_matches_reference, build  # noqa: E402
from tests.golden_util import bit_equal, load_cases  # noqa: E402

DEV = "cuda:0"
CALIB = load_cases("calib_cases.npz")


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from mhaq_amd import _lib, ops
    _lib.lib()
    return ops


# ------------------------------------------------------------------ min / max sweeps
@pytest.mark.parametrize("n", [1, 3, 4, 5, 1023, 4096, 4099, 1 << 20, (1 << 20) + 3, 50_176_000])
def test_minmax_matches_torch(ops, n):
    gen = torch.Generator(device=DEV).manual_seed(n % 1000)
    x = torch.randn(n, device=DEV, generator=gen) * 3
    mm = ops.minmax(x)
    ref = x.aminmax()
    assert float(mm[0]) == float(ref.min) and float(mm[1]) == float(ref.max)


def test_minmax_misaligned_views_nan_and_inf(ops):
    gen = torch.Generator(device=DEV).manual_seed(5)
    base = torch.randn(4 * 4099 + 9, device=DEV, generator=gen)
    for off in (1, 2, 3, 5):                      # 4-byte aligned only: the dword path
        v = base[off:off + 4 * 4099 + 1]
        assert v.data_ptr() % 16 != 0
        mm = ops.minmax(v)
        assert float(mm[0]) == float(v.min()) and float(mm[1]) == float(v.max())
    t = torch.randn(3, 5, 7, device=DEV).transpose(0, 2)            # non-contiguous: the op makes it dense
    mm = ops.minmax(t)
    assert float(mm[0]) == float(t.min()) and float(mm[1]) == float(t.max())
    x = torch.randn(10_001, device=DEV)
    for pos in (0, 4097, 10_000):
        y = x.clone()
        y[pos] = float("nan")                     # torch.amin / amax propagate NaN
        mm = ops.minmax(y)
        assert math.isnan(float(mm[0])) and math.isnan(float(mm[1]))
    y = x.clone()
    y[17], y[9000] = float("inf"), float("-inf")
    mm = ops.minmax(y)
    assert float(mm[0]) == -math.inf and float(mm[1]) == math.inf
    mm = ops.minmax(torch.full((777,), -0.75, device=DEV))          # zero-width range (the "pruned" branch's input)
    assert float(mm[0]) == float(mm[1]) == -0.75


def test_minmax_past_the_int32_range(ops):
    n = (1 << 31) + 4099
    x = torch.empty(n, device=DEV)
    x.normal_()
    x[n - 3], x[1 << 31] = 77.0, -91.0            # extremes beyond index 2^31
    mm = ops.minmax(x)
    assert float(mm[0]) == -91.0 and float(mm[1]) == 77.0
    del x
    torch.cuda.empty_cache()


@pytest.mark.parametrize("shape", [(64, 64, 3, 3), (512, 512, 3, 3), (50, 50, 3, 3), (5, 1, 1, 1), (7, 13), (3, 40001),
                                   (1000, 512), (2, 40000)])
def test_row_minmax_matches_torch(ops, shape):
    gen = torch.Generator(device=DEV).manual_seed(shape[0])
    w = torch.randn(*shape, device=DEV, generator=gen)
    dims = tuple(range(1, len(shape)))
    mn, mx = ops.row_minmax(w)
    assert torch.equal(mn, w.amin(dims)) and torch.equal(mx, w.amax(dims))
    if len(shape) == 4:                           # channels_last: every output channel is a permuted contiguous row
        wc = w.contiguous(memory_format=torch.channels_last)
        mn2, mx2 = ops.row_minmax(wc)
        assert torch.equal(mn2, mn) and torch.equal(mx2, mx)
    w2 = w.clone()
    w2.view(shape[0], -1)[0, -1] = float("nan")
    mn, mx = ops.row_minmax(w2)
    assert math.isnan(float(mn[0])) and math.isnan(float(mx[0]))
    if shape[0] > 1:
        assert torch.equal(mn[1:], w.amin(dims)[1:]) and torch.equal(mx[1:], w.amax(dims)[1:])


# ------------------------------------------------------------------ calibrate_* vs the reference's vectors
@pytest.mark.parametrize("name", sorted(CALIB))
def test_calibration_on_hip_layers_matches_reference(ops, name):
    import mhaq_amd as M
    from mhaq_amd import qat
    c = CALIB[name]
    acts, convs = build(c, M.NoisyAct, M.NoisyConv2d, M.QScheme.PER_CHANNEL, DEV)
    model = torch.nn.ModuleDict({"feed": Feed(acts, c, DEV), "convs": convs})
    qat.calibrate_weights(model, int(c["wbits"]))                         # mhaq_fq_row_minmax
    qat.calibrate_activations(model["feed"], list(range(3)), int(c["abits"]))   # mhaq_fq_minmax in the observer hook
    assert_matches_reference(c, acts, convs)


def test_calibration_at_resnet18_sizes_matches_oracle(ops):
    """Full-size tensors: the layer-1 activation of ResNet-18 at batch 250 (50.2 M elements, two observed batches)
    and its widest weight [512,512,3,3]; expectations from oracle/calib.py on the CPU."""
    import mhaq_amd as M
    from mhaq_amd import qat
    gen = torch.Generator().manual_seed(3)
    act = M.NoisyAct(signed=True).to(DEV)
    frozen = M.NoisyAct(signed=True).to(DEV)
    frozen.log_act_s.requires_grad_(False), frozen.log_act_q.requires_grad_(False)
    xs = [torch.randn(250, 64, 56, 56, generator=gen) * 2 + 0.1 * b for b in range(2)]

    class Two(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.f = act, frozen

        def forward(self, x):
            self.a(x)
            self.f(x[:4] * 0.5)
    qat.calibrate_activations(Two(), [x.to(DEV).contiguous(memory_format=torch.channels_last) for x in xs], 10)
    mn, mx = OC.observe(xs)
    mn2, mx2 = OC.observe([x[:4] * 0.5 for x in xs])
    exp = OC.mean_stats_activations([dict(min=mn, max=mx, grad_s=True, grad_q=True, grad_b=True),
                                     dict(min=mn2, max=mx2, grad_s=False, grad_q=False, grad_b=True)], abits=10)
    for m, e in ((act, exp[0]), (frozen, exp[1])):
        for k in ("log_act_s", "log_act_q", "act_b"):
            assert bit_equal(getattr(m, k).detach().cpu().numpy(), e[k].numpy()), k
    assert float(frozen.log_act_q - frozen.log_act_s) == 24.0
    conv = M.NoisyConv2d(512, 512, 3, qscheme=M.QScheme.PER_CHANNEL).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(512, 512, 3, 3, generator=gen) * math.sqrt(2.0 / 4608))
    qat.calibrate_weights(conv, 4)
    expw = OC.quantile_weights_s([dict(weight=conv.weight.detach().cpu(), log_wght_s=torch.full((512, 1, 1, 1), -12.0),
                                       grad=True)], wbits=4)[0]
    assert bit_equal(conv.log_wght_s.detach().cpu().numpy(), expw.numpy())
    assert float(conv.log_wght_s.min()) > -12.0          # every channel was raised to its 4-bit floor


def test_trainer_calibration_state_is_the_post_calibration_state_of_the_survey():
    """QATTrainer(calib_batches=...) leaves every quantizer in the state SURVEY.md 8d describes: log_act_q =
    log_act_s + 10, act_b = observed minimum, log_wght_s >= log2(range / 1023)."""
    import mhaq_amd as M
    from mhaq_amd import nets
    from mhaq_amd.qat import QATConfig, QATTrainer
    torch.manual_seed(1)
    cfg = QATConfig(excluded_layers=("features.init_block.conv", "output"), distillation=False)
    calib = torch.randn(16, 3, 32, 32, device=DEV)
    tr = QATTrainer(nets.resnet20_cifar(10), cfg, DEV, calib_batches=[calib], distributed=False)
    for m in tr.net.modules():
        if isinstance(m, M.NoisyAct):
            assert abs(float(m.log_act_q - m.log_act_s) - 10.0) < 1e-5
        if isinstance(m, M.NoisyConv2d):
            span = (m.weight.amax((1, 2, 3)) - m.weight.amin((1, 2, 3))).detach()
            assert torch.all(m.log_wght_s.detach().ravel() >= torch.log2(span / 1023) - 1e-5)
