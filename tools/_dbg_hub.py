import torch, sys
sys.path.insert(0, ".")
import mhaq_amd as M
from mhaq_amd import ops
from mhaq_amd.act_hub import ActGradHub
DEV="cuda:0"
acts = torch.nn.ModuleList([M.NoisyAct(init_s=-4, init_q=1, signed=sg, qnmethod=M.QNMethod.LSQ) for sg in (True, False, True)]).to(DEV).train()
xs = [torch.randn(4, 8, 9, 9, device=DEV) for _ in acts]
hub = ActGradHub(acts)
for rep in range(2):
    for p in acts.parameters(): p.grad = None
    hub.begin()
    print("outs", [(o.requires_grad, o.grad_fn) for o in hub._outs][:4])
    outs = [a(x.clone().requires_grad_(True)) for a, x in zip(acts, xs)]
    hub.end()
    print("y grad_fn", outs[0].grad_fn, "state", hub.state())
    torch.autograd.backward(outs, [torch.randn_like(o) for o in outs])
    print("state after", hub.state())
    for n, p in acts.named_parameters(): print(rep, n, None if p.grad is None else p.grad.tolist())
