#!/usr/bin/env python3
"""How long does the HOST need to enqueue one QAT step of the bench workload (ResNet-18, batch 250), versus the
GPU time of the step?  host << gpu means the step is GPU-bound and extra Python (DDP hooks, SyncBatchNorm) fits."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (seeds the MIOpen user db before torch loads MIOpen)
import torch  # noqa: E402

from mhaq_amd import nets, ops  # noqa: E402
from mhaq_amd.enums import QNMethod, QScheme  # noqa: E402
from mhaq_amd.qat import QATConfig, QATTrainer  # noqa: E402

dev = torch.device("cuda:0")
if os.environ.get("MHAQ_FORCE_COLLECTIVES") == "1":      # rehearse the data-parallel trainer on one GPU: RCCL, world size 1
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
ops.manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else "resnet18"
if which == "resnet18":
    cfg = QATConfig(qscheme=QScheme.PER_CHANNEL, qnmethod=QNMethod.AEWGS)
    net = nets.resnet18(1000).to(memory_format=torch.channels_last)
    x = torch.randn(250, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 1000, (250,), device=dev)
elif which == "rfdn":   # BASELINE configs[4] at the reference's training shape: batch 24 of 24x24 LR crops, L1, LSQ, no distillation
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    hw = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    cfg = QATConfig(qscheme=QScheme.PER_CHANNEL, qnmethod=QNMethod.LSQ, act_bit=2, weight_bit=2, distillation=False,
                    excluded_layers=("fea_conv", "upsampler.0"), learning_rate=5e-4, criterion=torch.nn.L1Loss())
    net = nets.rfdn()
    x = torch.rand(B, 3, hw, hw, device=dev) * 255.0
    y = torch.rand(B, 3, hw * 4, hw * 4, device=dev) * 255.0
else:       # resnet20 <batch>: the CIFAR configs (W4A4 STE, batch 128 or 1000)
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    cfg = QATConfig(qscheme=QScheme.PER_CHANNEL, qnmethod=QNMethod.STE, act_bit=4, weight_bit=4,
                    excluded_layers=("features.init_block.conv", "output"))
    net = nets.resnet20_cifar(100).to(memory_format=torch.channels_last)
    x = torch.randn(B, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 100, (B,), device=dev)
if os.environ.get("MHAQ_STEP_QSCHEME") == "0":              # the per-tensor override of SURVEY.md 8d config 2
    cfg.qscheme = QScheme.PER_TENSOR
if os.environ.get("MHAQ_STEP_NO_MULTI_FWD") == "1":          # A/B: every weight layer launches for itself
    cfg.multi_weight_forward = False
if os.environ.get("MHAQ_STEP_GROUP_ELEMS") is not None:      # A/B: 0 = per-layer weight backward launches
    cfg.weight_backward_group_elems = int(os.environ["MHAQ_STEP_GROUP_ELEMS"])
graph = {"1": True, "auto": "auto"}.get(os.environ.get("MHAQ_STEP_GRAPH"), False)   # QATTrainer(capture_graph=...)
tr = QATTrainer(net, cfg, dev, calib_batches=[x[:64]], capture_graph=graph)
for _ in range(5):
    tr.train_step(x, y)
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step(x, y)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    total.append((t2 - t0) * 1e3)
if tr.distributed:
    print(f"[data parallel, {'one flat gradient all-reduce per step' if tr._flat_sync else 'torch DDP'}]", end=" ")
print({True: "hipGraph replay:", False: "eager:", "auto": f"auto (-> {'graph' if tr._graph is not None else 'eager'}):"}[graph], end=" ")
print(f"host enqueue {sorted(host)[5]:.1f} ms/step, step (sync to sync) {sorted(total)[5]:.1f} ms")
