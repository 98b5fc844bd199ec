"""CPU: calibration.  (1) the oracle's restatement (oracle/calib.py) against the vectors
recorded from the reference's own apply_mean_stats_activations / apply_quantile_weights_s / MinMaxObserver._hook
(tests/golden/calib_cases.npz): bit-identical, incl. the zero-width "pruned" branch, the frozen-parameter
max_bits branch and its stickiness; (2) the product's host logic (mhaq_amd.qat.calibrate_*) over the oracle's CPU
layers with CPU min/max providers against the same vectors."""
import numpy as np
import pytest
import torch

from mhaq_amd import qat
from oracle import calib as OC
from oracle import ref_layers as RL
from tests.calib_util import Feed, assert_matches_reference, build, oracle_states
from tests.golden_util import bit_equal, load_cases

CALIB = load_cases("calib_cases.npz")


@pytest.mark.parametrize("name", sorted(CALIB))
def test_oracle_calibration_matches_reference(name):
    c = CALIB[name]
    acts, convs = oracle_states(c)
    for a in acts:
        a["min"], a["max"] = OC.observe(a["batches"])
    new_a = OC.mean_stats_activations(acts, abits=int(c["abits"]))
    for i, st in enumerate(new_a):
        for k in ("log_act_s", "log_act_q", "act_b"):
            assert bit_equal(st[k].numpy().reshape(-1), c[f"act{i}_{k}"].reshape(-1)), (i, k)
        assert [st["grad_s"], st["grad_q"], st["grad_b"]] == [bool(v) for v in c[f"act{i}_grad_out"]]
    new_w = OC.quantile_weights_s(convs, wbits=int(c["wbits"]))
    for i, ls in enumerate(new_w):
        assert bit_equal(ls.numpy().reshape(-1), c[f"conv{i}_log_wght_s"].reshape(-1)), i
    # the fixtures do exercise the branches
    assert float(new_a[2]["log_act_s"]) == 0.0 and not new_a[2]["grad_s"]                 # pruned
    bits = [float(a["log_act_q"] - a["log_act_s"]) for a in new_a]
    assert bits[0] == bits[1] == float(c["abits"]) and bits[3] == bits[4] == bits[5] == 24.0   # frozen, then sticky
    assert float(new_w[0][2]) == -12.0                                                  # constant channel keeps its scale


@pytest.mark.parametrize("name", sorted(CALIB))
def test_product_calibration_host_logic_matches_reference(name):
    c = CALIB[name]
    acts, convs = build(c, RL.NoisyAct, RL.NoisyConv2d, 1, "cpu")
    model = torch.nn.ModuleDict({"feed": Feed(acts, c, "cpu"), "convs": convs})
    qat.calibrate_weights(model, int(c["wbits"]), row_minmax_fn=qat._cpu_row_minmax)
    qat.calibrate_activations(model["feed"], list(range(3)), int(c["abits"]),
                              minmax_fn=lambda t: torch.stack(list(t.aminmax())))
    assert_matches_reference(c, acts, convs)


def test_per_tensor_weight_layer_uses_the_global_range():
    """The reference's function raises for PER_TENSOR layers (its [Co] range cannot be reshaped to [1]); the
    product calibrates them from the whole-tensor range."""
    conv = RL.NoisyConv2d(3, 5, 3, qscheme=0)
    with torch.no_grad():
        conv.weight.copy_(torch.linspace(-1, 2, conv.weight.numel()).view_as(conv.weight))
    qat.calibrate_weights(conv, 4, row_minmax_fn=qat._cpu_row_minmax)
    assert np.isclose(float(conv.log_wght_s), np.log2(3.0 / 15.0))
