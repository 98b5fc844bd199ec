"""GPU (-m gpu): the C-ABI entry points are stream-ordered, allocation-free and sync-free, so a forward +
backward of the fake-quant ops can be captured into a hipGraph and replayed on new data (SURVEY.md 8b:
'capture-safe ... so they can sit inside DDP's overlapped backward')."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def test_act_and_weight_ops_capture_and_replay():
    assert torch.cuda.is_available()
    from mhaq_amd import ops
    torch.manual_seed(0)
    x = torch.randn(8, 32, 28, 28, device=DEV, requires_grad=True)
    g = torch.randn(8, 32, 28, 28, device=DEV)
    ls = torch.tensor([-4.2], device=DEV, requires_grad=True)
    lq = torch.tensor([2.3], device=DEV, requires_grad=True)
    b = torch.tensor([-2.1], device=DEV, requires_grad=True)
    w = (torch.randn(32, 32, 3, 3, device=DEV) * 0.1).requires_grad_(True)
    lws = torch.full((32, 1, 1, 1), -7.3, device=DEV, requires_grad=True)
    G = torch.randn(32, 32, 3, 3, device=DEV)

    def step():
        y, _ = ops.fake_quant_act_layer(x, ls, lq, b, "LSQ")
        wq, _, _, lwq = ops.fake_quant_weight_layer(w, lws, "LSQ")
        return torch.autograd.grad([y, wq, lwq], [x, ls, lq, b, w, lws], [g, G, torch.ones_like(lwq)])

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):            # warm-up on the side stream, as torch.cuda.graph requires
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = step()
    # new data in the static input buffers, then replay
    with torch.no_grad():
        x.copy_(torch.randn_like(x) * 1.7)
        g.copy_(torch.randn_like(g))
        w.copy_(torch.randn_like(w) * 0.2)
        G.copy_(torch.randn_like(G))
        ls.fill_(-3.9)
    graph.replay()
    torch.cuda.synchronize()
    replayed = [t.clone() for t in captured]
    eager = step()
    for a, e in zip(replayed, eager):
        assert torch.equal(a, e)          # deterministic kernels: bit-identical to a direct call
