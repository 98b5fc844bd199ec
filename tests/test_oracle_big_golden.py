"""CPU: the eager oracle against the FULL-SIZE vectors recorded from the real reference (tests/golden/big_cases.npz,
oracle/gen_golden_big.py): 21 M-element activation tensors, outputs compared through checksums of their bit patterns."""
import numpy as np
import pytest
import torch

from oracle import fq_eager as O
from tests.golden_util import big_inputs, bits_checksum, load_cases

BIG = load_cases("big_cases.npz")


@pytest.mark.parametrize("name", ["act_big_lsq", "act_big_ste"])
def test_oracle_reproduces_the_reference_at_full_size(name):
    c = BIG[name]
    n, method = int(c["n"]), O.METHODS[int(c["method"])]
    x, g = big_inputs(c["seed"], n, float(c["scale"]))
    r = None
    if method == "STE":
        torch.manual_seed(int(c["seed"]))
        r = torch.randint_like(torch.empty(n), 2) - 0.5
    xr = torch.from_numpy(x).requires_grad_(True)
    ps = [torch.tensor([float(c[k])], requires_grad=True) for k in ("log_act_s", "log_act_q", "act_b")]
    y, _ = O.act_fake_quant(xr, *ps, r=r, method=method)
    y.backward(torch.from_numpy(g))
    assert np.array_equal(bits_checksum(y.detach().numpy()), c["y_sum"])
    assert np.array_equal(bits_checksum(xr.grad.numpy() + np.float32(0.0)), c["gx_sum"])
    s = 2.0 ** float(c["log_act_s"])
    yard = (float(c["abs_s"]) + float(c["abs_g"])) * s * np.log(2.0)
    assert abs(float(ps[0].grad) - float(c["g_log_act_s"])) <= 1e-6 * yard
