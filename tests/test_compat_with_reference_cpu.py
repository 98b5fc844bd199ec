"""CPU, build container only (skipped where /root/reference is absent, e.g. on the GPU box): after
mhaq_amd.compat.install() the REFERENCE's own ModelHelper and PotentialLoss modules -- imported unchanged --
operate on this package's layer classes: isinstance checks, parameter names and shapes line up."""
import os
import sys
import types

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference not mounted")


def test_reference_model_helper_runs_on_our_layers():
    import mhaq_amd
    from mhaq_amd import compat, wrap
    saved = {k: v for k, v in sys.modules.items() if k == "src" or k.startswith("src.")}
    path0 = list(sys.path)
    try:
        sys.dont_write_bytecode = True        # never leave __pycache__ in the (read-only) reference tree
        sys.path.insert(0, REF)
        import src  # noqa: F401
        pkg = types.ModuleType("src.quantization")
        pkg.__path__ = [os.path.join(REF, "src/quantization")]
        sys.modules["src.quantization"] = pkg
        compat.install()
        from src.quantization.gdnsq.utils.model_helper import ModelHelper      # the reference's file
        from src.quantization.gdnsq.gdnsq_loss import PotentialLossNoPred      # the reference's file
        from src.aux.types import QScheme as RefQScheme
        import src.quantization.gdnsq.layers.gdnsq_conv2d as shim
        assert shim.NoisyConv2d is mhaq_amd.NoisyConv2d
        net = torch.nn.Sequential(
            mhaq_amd.NoisyAct(signed=True),
            mhaq_amd.NoisyConv2d(3, 6, 3, qscheme=RefQScheme.PER_CHANNEL, qnmethod=mhaq_amd.QNMethod.LSQ),
            torch.nn.ReLU(),
            mhaq_amd.NoisyAct(signed=False),
            mhaq_amd.NoisyConv2d(6, 4, 3, qscheme=RefQScheme.PER_CHANNEL, qnmethod=mhaq_amd.QNMethod.LSQ))
        ref_vals = ModelHelper.get_model_values(net, RefQScheme.PER_CHANNEL)
        our_vals = wrap.get_model_values(net, mhaq_amd.QScheme.PER_CHANNEL)
        for a, b in zip(ref_vals, our_vals):
            assert torch.equal(a, b)
        crit = PotentialLossNoPred(None, p=1, a=4, w=4)
        crit.t, crit.loss_sum, crit.cnt = 0.5, torch.tensor(1.0), 1
        loss = crit((torch.tensor(2.0, requires_grad=True) * 1.0, *ref_vals))
        loss.backward()
        assert net[1].log_wght_s.grad is not None and net[0].log_act_s.grad is not None
    finally:
        sys.path[:] = path0
        for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
            del sys.modules[k]
        sys.modules.update(saved)
