#!/usr/bin/env python3
"""Peak device memory of one QAT step of the bench workload (ResNet-18, batch 250, channels_last, distillation) on the
HIP layers (backward recomputes the quantizer from x: 4 B/elem saved per quantizer) and on the oracle's eager layers
on the same GPU (the reference's chain saves ~4 full-size tensors per quantizer)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402,F401  (seeds the MIOpen user db before torch loads MIOpen)
import torch  # noqa: E402

from mhaq_amd import nets, ops  # noqa: E402
from mhaq_amd.enums import QNMethod, QScheme  # noqa: E402
from mhaq_amd.qat import QATConfig, QATTrainer  # noqa: E402
from oracle.ref_layers import ORACLE_LAYERS  # noqa: E402  (comparison leg only)

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 250
for name, layers in (("HIP layers", None), ("eager oracle layers on the GPU", ORACLE_LAYERS)):
    torch.manual_seed(0)
    ops.manual_seed(0)
    cfg = QATConfig(qscheme=QScheme.PER_CHANNEL, qnmethod=QNMethod.AEWGS)
    net = nets.resnet18(1000).to(memory_format=torch.channels_last)
    x = torch.randn(B, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 1000, (B,), device=dev)
    mm = (lambda t: torch.stack(list(t.aminmax()))) if layers is not None else None
    tr = QATTrainer(net, cfg, dev, calib_batches=[x[:32]], layers=layers, minmax_fn=mm, distributed=False,
                    capture_graph=False)
    for _ in range(2):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    import time
    t0 = time.perf_counter()
    for _ in range(5):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name:32s} batch {B}: peak allocated {torch.cuda.max_memory_allocated() / 2**30:6.2f} GiB, "
          f"{dt * 1e3:7.1f} ms/step", flush=True)
    del tr, net, x, y
    torch.cuda.empty_cache()
