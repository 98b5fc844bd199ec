"""GPU (-m gpu), needs TWO visible devices (skipped on the one-GPU test boxes): the cross-device guards.  The C ABI
launches on the stream it is handed and wants that stream's device current (include/mhaq_fq.h "Devices"); the host side
switches to its input's device per call -- `MHAQ_ON_DEVICE_OF` / the guards of plan_forward and plan_group_apply in
torch_binding.cpp, `ops._on_device` for the ctypes ops.  Here a model lives on cuda:1 while cuda:0 is current: the
activation layer, the weight layer, the model-wide plan (forward + grouped backward) and a ctypes op must give the bits
of the same run on cuda:0.  Until a box with two devices has run this file, the multi-device-per-process claim is
UNTESTED on hardware (said so in include/mhaq_fq.h and INTEGRATION.md)."""
import pytest
import torch

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible GPUs in one process")]


def _run(dev):
    import mhaq_amd as M
    from mhaq_amd import ops
    from mhaq_amd.multi import MultiTensorWeightQuant
    torch.manual_seed(3)
    ops.manual_seed(3)
    kw = dict(qscheme=M.QScheme.PER_CHANNEL, log_s_init=-6, qnmethod=M.QNMethod.STE)
    act = M.NoisyAct(init_s=-3, init_q=2, signed=True).to(dev)
    convs = torch.nn.ModuleList([M.NoisyConv2d(16, 8, 3, bias=False, **kw), M.NoisyConv2d(8, 12, 3, bias=False, **kw)]).to(dev)
    x = (torch.randn(4, 16, 9, 9) * 2).to(dev).requires_grad_(True)
    g = torch.randn(4, 16, 9, 9).to(dev)
    Gs = [torch.randn(m.weight.shape).to(dev) for m in convs]
    assert torch.cuda.current_device() == 0                     # the point: the inputs' device is NOT the current one
    y = act(x)
    y.backward(g)
    plan = MultiTensorWeightQuant(convs, joint_backward=False, backward_group_elems=1)
    plan.run()
    wqs = [m._quantized_weight()[0] for m in convs]
    torch.autograd.backward(wqs, Gs)
    wq1, _ = ops.fake_quant_weight_pc(convs[0].weight.detach(), torch.exp2(convs[0].log_wght_s.detach()), "LSQ")
    r = ops.fill_r(1000, 5, 1, device=dev)
    mm = ops.minmax(x.detach())
    outs = [y, x.grad, act.log_act_s.grad, act.log_act_q.grad, act.act_b.grad, *wqs, *[m.weight.grad for m in convs],
            *[m.log_wght_s.grad for m in convs], wq1, r, mm]
    assert all(t.device == torch.device(dev) for t in outs)
    return [t.detach().cpu() for t in outs]


def test_a_model_on_cuda1_with_cuda0_current_gives_the_bits_of_cuda0():
    torch.cuda.set_device(0)
    a = _run("cuda:0")
    b = _run("cuda:1")
    assert torch.cuda.current_device() == 0
    for i, (p, q) in enumerate(zip(a, b)):
        assert torch.equal(p, q), i
