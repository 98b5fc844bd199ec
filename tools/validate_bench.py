#!/usr/bin/env python3
"""QATTrainer.validate_step on the bench workload (ResNet-18, batch 250, channels_last): eval-mode forward through the
fused kernels with their integrity flags, the criterion, the six bit-width statistics and the converged flag
(gdnsq_quant.py:234-301, 385-420) -- ms per validation batch."""
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
torch.manual_seed(0)
ops.manual_seed(0)
cfg = QATConfig(qscheme=QScheme.PER_CHANNEL, qnmethod=QNMethod.AEWGS)
net = nets.resnet18(1000).to(memory_format=torch.channels_last)
x = torch.randn(250, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
y = torch.randint(0, 1000, (250,), device=dev)
tr = QATTrainer(net, cfg, dev, calib_batches=[x[:64]], capture_graph=False)
for _ in range(3):
    tr.train_step(x, y)
for _ in range(3):
    rec = tr.validate_step(x, y)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec = tr.validate_step(x, y)
    float(rec["val_loss"])
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print(f"validate_step: {ts[5]:.2f} ms per batch of 250 ({250 / ts[5] * 1e3:.0f} images/s); "
      f"record: { {k: (round(float(v), 4) if not isinstance(v, bool) else v) for k, v in rec.items()} }")
