#!/usr/bin/env python3
"""Top-1 parity proxy (SURVEY.md 8 row g: the reference's Top-1 needs checkpoints and datasets that are LFS stubs /
absent offline).  A synthetic 10-class image task that ResNet-20 can learn: (1) train the FP network, (2) run the
same W4A4 QAT recipe from it -- per-channel weights, STE activations, Sym-KL distillation from the FP teacher,
PotentialLoss, RAdam, 4-bit calibration -- once on the HIP layers and once on the oracle's eager layers (same
device, same data order, same initial state; the random sign streams differ by construction), (3) Top-1 of all
three on a held-out set.  One JSON line."""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mhaq_amd as M  # noqa: E402
from mhaq_amd import nets, ops  # noqa: E402
from mhaq_amd.qat import QATConfig, QATTrainer  # noqa: E402
from oracle.ref_layers import ORACLE_LAYERS  # noqa: E402  (the checker: never on the product path)

DEV = torch.device("cuda:0")


class Task:
    """10 classes = 10 smooth random templates; a sample is its class template, randomly shifted, plus noise."""

    def __init__(self, seed=0, noise=1.2):
        g = torch.Generator(device=DEV).manual_seed(seed)
        low = torch.randn(10, 3, 8, 8, device=DEV, generator=g)
        self.templates = torch.nn.functional.interpolate(low, size=(40, 40), mode="bilinear", align_corners=False)
        self.noise = noise

    def batch(self, n, g):
        y = torch.randint(0, 10, (n,), device=DEV, generator=g)
        dx = torch.randint(0, 9, (n,), device=DEV, generator=g)
        dy = torch.randint(0, 9, (n,), device=DEV, generator=g)
        idx = torch.arange(32, device=DEV)
        rows = (dy[:, None] + idx[None, :])                      # [n, 32]
        cols = (dx[:, None] + idx[None, :])
        t = self.templates[y]                                     # [n, 3, 40, 40]
        x = t[torch.arange(n, device=DEV)[:, None, None], :, rows[:, :, None], cols[:, None, :]]   # [n, 32, 32, 3]
        x = x.permute(0, 3, 1, 2).contiguous()
        return x + self.noise * torch.randn(x.shape, device=DEV, generator=g), y


@torch.no_grad()
def top1(net, xs, ys):
    net.eval()
    hit = 0
    for x, y in zip(xs, ys):
        hit += int((net(x).argmax(1) == y).sum())
    net.train()
    return 100.0 * hit / sum(len(y) for y in ys)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fp-steps", type=int, default=400)
    ap.add_argument("--qat-steps", type=int, default=400)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--seeds", type=int, default=2, help="QAT repetitions per side (different sign / data seeds)")
    ap.add_argument("--noise", type=float, default=3.0, help="per-pixel noise sigma (templates have unit scale)")
    ap.add_argument("--qnmethod", default="STE", choices=["STE", "LSQ", "AEWGS"],
                    help="estimator of weights AND activations.  LSQ has no random term: both sides then run the same "
                         "deterministic recipe and differ only by fp32 summation order.  AEWGS: the weight estimator of "
                         "configs[3] (gdnsq_config_resnet18_imagenet_aewgs_w1a1.yaml); activations keep STE")
    ap.add_argument("--test-batches", type=int, default=8, help="held-out batches of 500 samples")
    ap.add_argument("--calib-bits", type=int, default=0,
                    help="calibration bit width; 0 = the target width (a direct low-bit start).  The reference's ResNet "
                         "configs calibrate at 10 bits and let the PotentialLoss bring the widths down")
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    task = Task(noise=args.noise)
    gt = torch.Generator(device=DEV).manual_seed(999)
    test = [task.batch(500, gt) for _ in range(args.test_batches)]  # 4000 held-out samples by default
    xs, ys = [b[0] for b in test], [b[1] for b in test]
    # ---- (1) FP network
    torch.manual_seed(0)
    fp = nets.resnet20_cifar(10).to(DEV)
    opt = torch.optim.AdamW(fp.parameters(), lr=2e-3, weight_decay=1e-4)
    g = torch.Generator(device=DEV).manual_seed(1)
    for i in range(args.fp_steps):
        x, y = task.batch(args.batch, g)
        loss = torch.nn.functional.cross_entropy(fp(x), y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    out = {"task": f"synthetic 10-class 32x32 (shifted smooth templates + noise sigma {args.noise}), ResNet-20, 4000 held-out "
                   f"samples".replace("4000", str(500 * args.test_batches)),
           "fp_top1": round(top1(fp, xs, ys), 2), "fp_steps": args.fp_steps, "qat_steps": args.qat_steps,
           "recipe": f"W{args.bits}A{args.bits}, per-channel {args.qnmethod} weights + {'LSQ' if args.qnmethod == 'LSQ' else 'STE'} activations, Sym-KL distillation, "
                     f"PotentialLoss, RAdam 2e-3, batch {args.batch}, calibrated at {args.calib_bits or args.bits} bits"}
    # ---- (2) the same QAT recipe on both layer sets
    res = {"hip": [], "oracle": []}
    for side, layers in (("hip", None), ("oracle", ORACLE_LAYERS)):
        for rep in range(args.seeds):
            torch.manual_seed(100 + rep)
            ops.manual_seed(100 + rep)
            cfg = QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod[args.qnmethod], act_bit=args.bits,
                            weight_bit=args.bits, calib_act_bit=args.calib_bits or args.bits,
                            calib_weight_bit=args.calib_bits or args.bits,
                            excluded_layers=("features.init_block.conv", "output"), distillation=True,
                            learning_rate=2e-3, warmup=20)
            gq = torch.Generator(device=DEV).manual_seed(7 + rep)
            calib = [task.batch(256, gq)[0]]
            mm = (lambda t: torch.stack(list(t.aminmax()))) if layers is not None else None
            tr = QATTrainer(copy.deepcopy(fp), cfg, DEV, calib_batches=calib, layers=layers, minmax_fn=mm,
                            distributed=False)
            if args.qnmethod == "LSQ":       # the wrap rule builds STE activation quantizers (gdnsq_quant.py:501-518)
                for m in tr.net.modules():
                    if hasattr(m, "log_act_s") and hasattr(m, "Q"):
                        m.Q.qnmethod = M.QNMethod[args.qnmethod]
                    elif hasattr(m, "log_act_s"):                # the oracle's NoisyAct keeps the name
                        m.qnmethod = args.qnmethod
            post_calib = round(top1(tr.net, xs, ys), 2)          # quantized at `bits`, before any QAT step
            t0 = time.perf_counter()
            for i in range(args.qat_steps):
                x, y = task.batch(args.batch, gq)
                last = tr.train_step(x, y)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[side].append({"top1": round(top1(tr.net, xs, ys), 2), "post_calibration_top1": post_calib,
                              "final_loss": round(float(last), 5), "ms_per_step": round(dt / args.qat_steps * 1e3, 2)})
            print(f"[top1_proxy] {side} seed {rep + 1}/{args.seeds}: {res[side][-1]}", file=sys.stderr, flush=True)
            del tr
    out["hip"], out["oracle"] = res["hip"], res["oracle"]
    mh = sum(r["top1"] for r in res["hip"]) / len(res["hip"])
    mo = sum(r["top1"] for r in res["oracle"]) / len(res["oracle"])
    out["mean_top1_hip"], out["mean_top1_oracle"] = round(mh, 2), round(mo, 2)
    out["top1_difference"] = round(mh - mo, 2)
    if args.seeds > 1:      # is the difference distinguishable from the seed-to-seed spread of either side?
        import statistics
        sh, so = (statistics.stdev(r["top1"] for r in res[k]) for k in ("hip", "oracle"))
        out["stdev_top1_hip"], out["stdev_top1_oracle"] = round(sh, 2), round(so, 2)
        out["stderr_of_difference"] = round(((sh ** 2 + so ** 2) / args.seeds) ** 0.5, 2)
        # the same seed on both sides shares the FP start, the calibration batch and the data order: paired differences
        d = [a["top1"] - b["top1"] for a, b in zip(res["hip"], res["oracle"])]
        out["paired_difference_mean_stderr"] = [round(statistics.mean(d), 3),
                                                round(statistics.stdev(d) / len(d) ** 0.5, 3)]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
