"""CPU: the oracle's restatement of the reference's bit-width statistics (oracle/stats.py) against the vectors
recorded from the reference's own utils/model_stats.py functions (tests/golden/stats_cases.npz)."""
import numpy as np
import pytest

from oracle import fq_eager as O
from oracle import stats as OS
from tests.golden_util import T, bit_equal, load_cases

STATS = load_cases("stats_cases.npz")


@pytest.mark.parametrize("name", sorted(STATS))
def test_oracle_statistics_match_reference(name):
    c = STATS[name]
    pc = bool(c["per_channel"])
    per_layer = []
    for i in range(3):
        w, ls = T(c[f"conv{i}_w"]), T(c[f"conv{i}_log_wght_s"])
        assert OS.true_layer_bit_width(w, ls, pc, max=True) == float(c[f"conv{i}_bw_max"])
        assert OS.true_layer_bit_width(w, ls, pc, max=False) == float(c[f"conv{i}_bw_mean"])
        assert bit_equal(OS.layer_wnb_bit_width(w, ls, pc).numpy(), c[f"conv{i}_wnb"])
        per_layer.append(OS.true_layer_bit_width(w, ls, pc, max=True))
    assert np.max(per_layer) == float(c["true_weights_width_max"])
    assert np.mean(per_layer) == float(c["true_weights_width_mean"])
    bws = []
    for i in range(3):
        ls, lq, b = (T(c[f"act{i}_params"][k:k + 1]) for k in range(3))
        y, q = O.act_fake_quant(T(c[f"act{i}_x"]), ls, lq, b, method="LSQ")
        bw = O.act_bit_width(q)
        assert bit_equal(bw.numpy(), c[f"act{i}_bw"])
        bws.append(float(bw))
    assert np.max(bws) == float(c["true_activations_width_max"])
    assert abs(np.mean(bws) - float(c["true_activations_width_mean"])) < 1e-6
