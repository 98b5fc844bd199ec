"""GPU (-m gpu): QATTrainer.validate_step -- the eval-mode pass of the reference's validation step
(gdnsq_quant.py:234-301, 385-420): criterion on the quantized prediction, bit-width statistics, converged
flag, and the eval asserts of gdnsq.py:211-217 as one lazy device-flag check."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _trainer(distillation=False):
    import mhaq_amd as M
    from mhaq_amd import nets, ops
    from mhaq_amd.qat import QATConfig, QATTrainer
    torch.manual_seed(3)
    ops.manual_seed(3)
    cfg = QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.LSQ, act_bit=4, weight_bit=4,
                    excluded_layers=("features.init_block.conv", "output"), warmup=2, distillation=distillation)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(16, 3, 32, 32, generator=g).to(DEV)
    y = torch.randint(0, 10, (16,), generator=g).to(DEV)
    tr = QATTrainer(nets.resnet20_cifar(10), cfg, DEV, calib_batches=[x], distributed=False)
    return tr, x, y


def test_validate_step_matches_the_oracle_layers_in_eval_mode():
    from mhaq_amd import stats, wrap
    from oracle.ref_layers import ORACLE_LAYERS
    tr, x, y = _trainer()
    for _ in range(2):
        tr.train_step(x, y)
    out = tr.validate_step(x, y)
    assert tr.module.training                        # back in training mode afterwards
    # the same network rebuilt from the oracle's eager layers, same parameters, eval mode, on the device
    ref = copy.deepcopy(tr.net)
    wrap_state = ref.state_dict()
    from mhaq_amd import nets
    onet = nets.resnet20_cifar(10).to(DEV)
    wrap.quantize_model(onet, tr.cfg.qscheme, tr.cfg.qnmethod, tr.cfg.excluded_layers, tr.cfg.quantize_bias,
                        tr.cfg.act_bit, layers=ORACLE_LAYERS)
    onet.to(DEV).load_state_dict(wrap_state)
    onet.eval()
    with torch.no_grad():
        logits = onet(x)
    ref_loss = torch.nn.functional.cross_entropy(logits, y)
    assert torch.allclose(out["val_loss"], ref_loss, rtol=1e-5, atol=1e-6)
    assert float(out["top1"]) == float((logits.argmax(1) == y).float().mean())
    # statistics: the same numbers the stand-alone functions give, finite, and consistent with each other
    assert out["actual_weights_max_bit_width"] == stats.get_true_weights_width(tr.net)
    assert out["actual_weights_bit_width"] <= out["actual_weights_max_bit_width"] + 1e-6
    assert out["actual_activations_bit_width"] <= out["actual_activations_max_bit_width"] + 1e-6
    ref_bw = max(float(m.bw) for m in onet.modules() if hasattr(m, "log_act_s"))
    assert abs(out["actual_activations_max_bit_width"] - ref_bw) < 1e-6      # gdnsq_act.py:51-54 on both sides
    assert torch.isfinite(out["mean_weights_bit_width"]) and torch.isfinite(out["mean_activations_bit_width"])
    assert isinstance(out["converged"], bool)


def test_validate_step_raises_the_reference_assertion_on_nan_input():
    tr, x, y = _trainer()
    bad = x.clone()
    bad[0, 0, 0, 0] = float("nan")
    with pytest.raises(AssertionError, match="Not all elements in the tensor"):
        tr.validate_step(bad, y)
    assert tr.module.training
