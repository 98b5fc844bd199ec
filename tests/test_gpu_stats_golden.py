"""GPU (-m gpu): mhaq_amd.stats and NoisyAct.bw on the HIP path against the vectors recorded
from the reference's own utils/model_stats.py functions (tests/golden/stats_cases.npz).  Level counts are integers,
so log2(count) agrees to the last bit or the neighbouring ulp of the device's log2 (1e-6)."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.golden_util import T, load_cases  # noqa: E402

DEV = "cuda:0"
STATS = load_cases("stats_cases.npz")
CONVS = ((3, 6, 3), (6, 4, 3), (4, 8, 1))


@pytest.mark.parametrize("name", sorted(STATS))
def test_statistics_on_hip_layers_match_reference(name):
    import mhaq_amd as M
    from mhaq_amd import stats
    from mhaq_amd.gdnsq import check_model_integrity
    c = STATS[name]
    qs = M.QScheme.PER_CHANNEL if bool(c["per_channel"]) else M.QScheme.PER_TENSOR
    convs = torch.nn.ModuleList([M.NoisyConv2d(ci, co, k, qscheme=qs) for ci, co, k in CONVS])
    acts = torch.nn.ModuleList([M.NoisyAct(signed=bool(c[f"act{i}_signed"])) for i in range(3)])
    with torch.no_grad():
        for i, m in enumerate(convs):
            m.weight.copy_(T(c[f"conv{i}_w"]))
            m.log_wght_s.copy_(T(c[f"conv{i}_log_wght_s"]).view_as(m.log_wght_s))
        for i, a in enumerate(acts):
            ls, lq, b = (float(v) for v in c[f"act{i}_params"])
            a.log_act_s.fill_(ls), a.log_act_q.fill_(lq), a.act_b.fill_(b)
    model = torch.nn.ModuleDict({"convs": convs, "acts": acts}).to(DEV).eval()
    with torch.no_grad():
        for i, a in enumerate(acts):
            a(T(c[f"act{i}_x"]).to(DEV))
    check_model_integrity(model)                                  # the eval asserts, one sync: nothing flagged
    for i, m in enumerate(convs):
        assert abs(stats.get_true_layer_bit_width(m, max=True) - float(c[f"conv{i}_bw_max"])) < 1e-6
        assert abs(stats.get_true_layer_bit_width(m, max=False) - float(c[f"conv{i}_bw_mean"])) < 1e-6
        wnb = stats.get_layer_wnb_bit_width(m.weight.detach(), m.log_wght_s.detach(), m.qscheme)
        assert abs(float(wnb) - float(c[f"conv{i}_wnb"])) < 1e-5
    for i, a in enumerate(acts):
        assert abs(float(a.bw) - float(c[f"act{i}_bw"])) < 1e-6
    assert abs(stats.get_true_weights_width(model, max=True) - float(c["true_weights_width_max"])) < 1e-6
    assert abs(stats.get_true_weights_width(model, max=False) - float(c["true_weights_width_mean"])) < 1e-6
    assert abs(float(stats.get_weights_bit_width_mean(model)) - float(c["weights_bit_width_mean"])) < 1e-5
    assert abs(float(stats.get_activations_bit_width_mean(model)) - float(c["activations_bit_width_mean"])) < 1e-6
    assert abs(stats.get_true_activations_width(model, max=True) - float(c["true_activations_width_max"])) < 1e-6
    assert abs(stats.get_true_activations_width(model, max=False) - float(c["true_activations_width_mean"])) < 1e-6
    crit = types.SimpleNamespace(wt=float(c["true_weights_width_max"]) + 1e-6,      # the device's log2 may sit one ulp up
                                 at=float(c["true_activations_width_max"]) + 1e-6)
    assert stats.is_converged(model, crit)
    crit.wt -= 0.5
    assert not stats.is_converged(model, crit)
