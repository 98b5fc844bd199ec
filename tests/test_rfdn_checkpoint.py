"""The reference's one real checkpoint (data/models/RFDN_AIM.pth, the starting point of
config/gdnsq_config_rfdn_lsq_w2a2.yaml) as a data fixture: tests/golden/rfdn_aim_weights.npz, written by
oracle/gen_rfdn_fixture.py.  CPU: it loads into this repo's RFDN with strict=True (checkpoint compatibility of
mhaq_amd/nets.py and of the wrapped model's parameter names) and super-resolves.  GPU: the pretrained model wrapped
with the HIP layers and with the oracle's eager layers, calibrated on the same batch, gives the same image."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
WEIGHTS = os.path.join(HERE, "golden", "rfdn_aim_weights.npz")


def _state():
    with np.load(WEIGHTS) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


def _smooth_image(h=96, w=128, seed=0):
    """A band-limited RGB test image in [0, 1] (no image files needed)."""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(1, 3, h // 8, w // 8, generator=g)
    return F.interpolate(low, size=(h, w), mode="bicubic", align_corners=False).clamp(0, 1)


def test_reference_checkpoint_loads_strictly_and_super_resolves():
    from mhaq_amd import nets
    sd = _state()
    assert len(sd) == 128 and sum(v.numel() for v in sd.values()) == 433448
    net = nets.rfdn()
    net.load_state_dict(sd, strict=True)
    net.eval()
    hr = _smooth_image()
    lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", antialias=True, align_corners=False).clamp(0, 1)
    with torch.no_grad():
        sr = (net(lr * 255.0) / 255.0).clamp(0, 1)      # LVisionSR.forward scales by 255 (vision_sr_module.py:49-53)
        bic = F.interpolate(lr, scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)
    psnr = lambda a: float(10 * torch.log10(1.0 / (a - hr).square().mean()))
    assert sr.shape == hr.shape
    assert psnr(sr) > psnr(bic) - 0.5 and psnr(sr) > 30.0     # a trained x4 network, not noise


def test_wrapped_pretrained_model_keeps_the_checkpoint_names():
    """GDNSQQuant.quantize replaces a conv by Sequential(activations_quantizer, "0") sharing weight and bias: the
    reference's noisy checkpoints carry `<name>.0.weight` (gdnsq_quant.py:501-518); the 1x1 and excluded convs keep
    their names."""
    from mhaq_amd import nets, wrap
    from oracle.ref_layers import ORACLE_LAYERS
    net = nets.rfdn()
    sd = _state()
    net.load_state_dict(sd, strict=True)
    wrap.quantize_model(net, 1, "LSQ", ("fea_conv", "upsampler.0"), layers=ORACLE_LAYERS)
    keys = set(net.state_dict().keys())
    assert "B1.c1_r.0.weight" in keys and "B1.c1_r.activations_quantizer.log_act_s" in keys
    assert "fea_conv.weight" in keys and "B1.c1_d.weight" in keys and "upsampler.0.weight" in keys
    assert torch.equal(net.state_dict()["B1.c1_r.0.weight"], sd["B1.c1_r.weight"])
    assert sum(1 for k in keys if k.endswith("activations_quantizer.log_act_s")) == 33


@pytest.mark.gpu
def test_hip_and_oracle_layers_agree_on_the_pretrained_model():
    import copy
    from mhaq_amd import nets, wrap
    from mhaq_amd.qat import calibrate_activations, calibrate_weights, _cpu_row_minmax
    from oracle.ref_layers import ORACLE_LAYERS
    dev = "cuda:0"
    det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        base = nets.rfdn()
        base.load_state_dict(_state(), strict=True)
        hr = _smooth_image(96, 128, 1).to(dev)
        lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", antialias=True, align_corners=False).clamp(0, 1)
        x = lr * 255.0
        outs = []
        for layers in (None, ORACLE_LAYERS):
            net = copy.deepcopy(base).to(dev)
            wrap.quantize_model(net, 1, "LSQ", ("fea_conv", "upsampler.0"), layers=layers)
            net.to(dev)
            calibrate_weights(net, 4, row_minmax_fn=None if layers is None else _cpu_row_minmax)
            calibrate_activations(net, [x], 4, minmax_fn=None if layers is None else
                                  (lambda t: torch.stack(list(t.aminmax()))))
            net.eval()
            with torch.no_grad():
                outs.append((net(x) / 255.0).clamp(0, 1))
        a, b = outs
        # every quantizer is bit-exact on equal inputs and both models run the same convolutions on the same device
        assert float((a - b).abs().max()) <= 1e-6, float((a - b).abs().max())
        psnr = lambda t: float(10 * torch.log10(1.0 / (t - hr).square().mean()))
        assert abs(psnr(a) - psnr(b)) <= 1e-4 and psnr(a) > 20.0
    finally:
        torch.backends.cudnn.deterministic = det


@pytest.mark.gpu
def test_qat_from_the_pretrained_checkpoint_tracks_the_oracle():
    """BASELINE configs[4]'s recipe (per-channel LSQ, L1, RAdam, batch 24 of 24x24 LR crops, inputs x255) for 25 steps
    from the reference's pretrained RFDN, LSQ activations (nothing random): the HIP-layer trainer and the oracle-layer
    trainer see the same crops; their losses agree step for step while the trajectories are still close, and the
    end-task metric (PSNR on luminance) ends within 0.1 dB."""
    import copy
    import mhaq_amd as M
    from mhaq_amd import nets, ops
    from mhaq_amd.qat import QATConfig, QATTrainer
    from oracle.ref_layers import ORACLE_LAYERS
    dev = torch.device("cuda:0")
    det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True

    class L1On255(torch.nn.Module):
        def forward(self, out, target):
            return F.l1_loss(out / 255.0, target)

    def lum(t):
        c = torch.tensor([65.738, 129.057, 25.064], device=t.device).reshape(1, 3, 1, 1) / 256
        return t.mul(c).sum(dim=1, keepdim=True)

    try:
        base = nets.rfdn()
        base.load_state_dict(_state(), strict=True)
        hrs = torch.cat([_smooth_image(96, 96, s) for s in range(48)]).to(dev)
        lrs = F.interpolate(hrs, scale_factor=0.25, mode="bicubic", antialias=True, align_corners=False).clamp(0, 1)
        test_hr = _smooth_image(128, 160, 99).to(dev)
        test_lr = F.interpolate(test_hr, scale_factor=0.25, mode="bicubic", antialias=True,
                                align_corners=False).clamp(0, 1)
        res = []
        for layers in (None, ORACLE_LAYERS):
            torch.manual_seed(3)
            ops.manual_seed(3)
            cfg = QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.LSQ, act_bit=8, weight_bit=8,
                            calib_act_bit=8, calib_weight_bit=8, excluded_layers=("fea_conv", "upsampler.0"),
                            distillation=False, learning_rate=5e-4, warmup=10, criterion=L1On255())
            mm = (lambda t: torch.stack(list(t.aminmax()))) if layers is not None else None
            tr = QATTrainer(copy.deepcopy(base), cfg, dev, calib_batches=[lrs[:24] * 255.0], layers=layers,
                            minmax_fn=mm, distributed=False, capture_graph=False)
            for m in tr.net.modules():
                if hasattr(m, "log_act_s"):
                    if hasattr(m, "Q"):
                        m.Q.qnmethod = M.QNMethod.LSQ
                    else:
                        m.qnmethod = "LSQ"
            losses = []
            for k in range(25):
                idx = torch.arange(24) + (k % 2) * 24
                losses.append(float(tr.train_step(lrs[idx] * 255.0, hrs[idx])))
            tr.net.eval()
            with torch.no_grad():
                sr = (tr.net(test_lr * 255.0) / 255.0).clamp(0, 1)
            res.append((losses, float(10 * torch.log10(1.0 / (lum(sr) - lum(test_hr)).square().mean()))))
        (l_hip, p_hip), (l_ref, p_ref) = res
        for k in range(5):       # the first steps: same state, same batch -> the same loss to fp32 summation order
            assert abs(l_hip[k] - l_ref[k]) <= 2e-4 * abs(l_ref[k]), (k, l_hip[k], l_ref[k])
        assert abs(l_hip[-1] - l_ref[-1]) <= 0.1 * abs(l_ref[-1])
        assert abs(p_hip - p_ref) <= 0.1 and p_hip > 25.0, (p_hip, p_ref)
    finally:
        torch.backends.cudnn.deterministic = det
