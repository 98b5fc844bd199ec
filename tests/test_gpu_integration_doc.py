"""GPU (-m gpu): the binding INTEGRATION.md section 3 tells a maintainer of the reference to paste next to gdnsq.py -- a ctypes
stub over include/mhaq_fq.h plus one torch.autograd.Function -- is taken out of the document, executed as it stands
(only the library's path is filled in) and held to the product's own op: a sample that drifts from the ABI fails here."""
import os
import re

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def _sample():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 3. Bind the C ABI directly"):text.index("## 4.")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) == 1 and "class FakeQuantAct(torch.autograd.Function)" in blocks[0]
    return blocks[0]


def test_the_documented_ctypes_binding_runs_and_equals_the_product_op():
    from mhaq_amd import _lib, ops
    _lib.lib()
    code = _sample().replace('C.CDLL("libmhaq_fq.so")', f'C.CDLL("{_lib.LIB_PATH}")')
    offsets = iter(range(41, 1000))
    ns = {"SEED": 77, "next_offset": lambda: next(offsets)}
    exec(compile(code, "INTEGRATION.md section 3", "exec"), ns)
    FakeQuantAct = ns["FakeQuantAct"]
    torch.manual_seed(0)
    x = (torch.randn(5, 7, 9, 11, device=DEV) * 2)
    g = torch.randn_like(x)
    s = torch.tensor([0.2371], device=DEV)
    b = torch.tensor([-1.9], device=DEV)
    hi = b + 16 * s - s
    leaves = [t.clone().requires_grad_(True) for t in (x, s, b, b, hi)]
    y = FakeQuantAct.apply(*leaves)
    y.backward(g)
    # the product's op on the same tensors with the signs of the stream the sample drew: (SEED, offset 41)
    r = ops.fill_r(x.numel(), 77, 41, DEV)
    ref = [t.clone().requires_grad_(True) for t in (x, s, b, b, hi)]
    y2 = ops.fake_quant_per_tensor(*ref, "STE", r_sign=r)
    y2.backward(g)
    assert torch.equal(y, y2)
    for a, c in zip(leaves, ref):
        assert torch.equal(a.grad, c.grad)
    # ... and the same numbers as the reference's own op chain (gdnsq.py:189-229) on these tensors, from the eager oracle
    from oracle import fq_eager as O
    o = [t.detach().cpu().clone().requires_grad_(True) for t in (x, s, b, b, hi)]
    yo = O.dequantize(O.quantize(o[0], o[1], o[2], o[3], o[4], "STE", r.cpu().float().reshape(x.shape) * 0.5), o[1], o[2])
    yo.backward(g.cpu())
    assert torch.equal(y.detach().cpu(), yo.detach()) and torch.equal(leaves[0].grad.cpu(), o[0].grad)
