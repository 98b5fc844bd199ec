#!/usr/bin/env python3
"""Why is the hipGraph replay of the GPU-bound ResNet-18 step slower than the eager step?  Same step four ways:
eager / eager with the capturable optimizer / graph / graph without the teacher's side stream."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402,F401  (seeds the MIOpen user db before torch loads MIOpen)
import torch  # noqa: E402

from mhaq_amd import nets, ops  # noqa: E402
from mhaq_amd.enums import QNMethod, QScheme  # noqa: E402
from mhaq_amd.qat import QATConfig, QATTrainer  # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
x = torch.randn(250, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
y = torch.randint(0, 1000, (250,), device=dev)


def run(name, graph, capturable, overlap):
    torch.manual_seed(0)
    ops.manual_seed(0)
    cfg = QATConfig(qscheme=QScheme.PER_CHANNEL, qnmethod=QNMethod.AEWGS, overlap_teacher=overlap)
    net = nets.resnet18(1000).to(memory_format=torch.channels_last)
    fac = None
    if capturable and not graph:
        fac = lambda p, lr: torch.optim.RAdam(p, torch.tensor(float(lr), device=dev), capturable=True)  # noqa: E731
    tr = QATTrainer(net, cfg, dev, calib_batches=[x[:64]], capture_graph=graph, optimizer_factory=fac)
    for _ in range(6):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter() - t0) * 100:.2f} ms/step", flush=True)
    del tr
    torch.cuda.empty_cache()


which = sys.argv[1:] or ["eager", "eager_cap", "graph", "graph_serial"]
for w in which:
    run(w, graph=w.startswith("graph"), capturable=w in ("eager_cap",), overlap=not w.endswith("serial"))
