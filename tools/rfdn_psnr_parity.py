#!/usr/bin/env python3
"""End-task parity on the reference's own pretrained model (SURVEY.md 8 row g, BASELINE configs[4]).

The one real checkpoint the reference ships is the pretrained RFDN that config/gdnsq_config_rfdn_lsq_w2a2.yaml starts
from (tests/golden/rfdn_aim_weights.npz holds its tensors; it loads into mhaq_amd.nets.rfdn() with strict=True).
The end-task metric of that config is PSNR on the luminance channel (vision_sr_module.py:151-158: clamp to [0, 1],
to_luminance, piq.psnr).  No SR dataset is available offline; the container image holds a handful of real RGB
photographs (scikit-image's sample images), which is enough for a PARITY statement: the config's QAT recipe at
--bits -- per-channel LSQ weights, calibration at that width, L1 loss, RAdam 5e-4, batch 24 of 24x24 LR crops (x4, HR 96x96), inputs
scaled by 255 like LVisionSR.forward -- from the same pretrained state on the same crops, once on the HIP layers
and once on the oracle's eager layers, then PSNR-Y of both students on held-out images.  With LSQ activation
quantizers nothing in the step is random, so the two sides differ only by fp32 summation order.  One JSON line."""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import mhaq_amd as M  # noqa: E402
from mhaq_amd import nets, ops  # noqa: E402
from mhaq_amd.qat import QATConfig, QATTrainer  # noqa: E402
from oracle.ref_layers import ORACLE_LAYERS  # noqa: E402  (the checker: never on the product path)

DEV = torch.device("cuda:0")
IMAGES = os.environ.get("MHAQ_SR_IMAGES", "/opt/conda/lib/python3.9/site-packages/skimage/data")
TRAIN = ("astronaut.png", "coffee.png", "ihc.png", "motorcycle_left.png", "color.png")
HELD_OUT = ("chelsea.png", "motorcycle_right.png", "phantom.png")
WEIGHTS = os.path.join(ROOT, "tests", "golden", "rfdn_aim_weights.npz")


def pretrained_rfdn():
    net = nets.rfdn()
    with np.load(WEIGHTS) as z:
        net.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files}, strict=True)
    return net


def load_rgb(name):
    from PIL import Image
    im = np.asarray(Image.open(os.path.join(IMAGES, name)).convert("RGB"), dtype=np.float32) / 255.0
    t = torch.from_numpy(im).permute(2, 0, 1).contiguous()
    return t[:, :t.shape[1] // 4 * 4, :t.shape[2] // 4 * 4]


def downscale(hr):
    """LR = bicubic x1/4 with antialiasing, clamped to [0, 1] (what an SR data pipeline stores as the LR image)."""
    return F.interpolate(hr, scale_factor=0.25, mode="bicubic", antialias=True, align_corners=False).clamp(0, 1)


def to_luminance(t):            # transforms.py:389-391
    c = torch.tensor([65.738, 129.057, 25.064], device=t.device).reshape(1, 3, 1, 1) / 256
    return t.mul(c).sum(dim=1, keepdim=True)


@torch.no_grad()
def psnr_y(net, pairs):
    """vision_sr_module.py:151-158 on full images: mean over images of PSNR(luminance), data range 1."""
    net.eval()
    vals = []
    for lr, hr in pairs:
        try:
            sr = (net(lr * 255.0) / 255.0).clamp(0, 1)
        except AssertionError as e:      # the reference's eval-mode integrity asserts (gdnsq.py:211-217): a diverged run
            net.train()
            return float("nan"), [repr(e)]
        mse = (to_luminance(sr) - to_luminance(hr)).square().mean()
        vals.append(float(10 * torch.log10(1.0 / mse)))
    net.train()
    return sum(vals) / len(vals), vals


class Crops:
    """Random aligned 96x96 HR crops (24x24 LR) from the training images, with flips: the same stream for a seed."""

    def __init__(self, names):
        self.hr = [load_rgb(n).to(DEV) for n in names]
        self.lr = [downscale(h[None])[0] for h in self.hr]

    def batch(self, n, g):
        xs, ys = [], []
        for _ in range(n):
            k = int(torch.randint(0, len(self.hr), (1,), generator=g))
            h, w = self.lr[k].shape[1:]
            i = int(torch.randint(0, h - 24 + 1, (1,), generator=g))
            j = int(torch.randint(0, w - 24 + 1, (1,), generator=g))
            lr = self.lr[k][:, i:i + 24, j:j + 24]
            hr = self.hr[k][:, 4 * i:4 * i + 96, 4 * j:4 * j + 96]
            if int(torch.randint(0, 2, (1,), generator=g)):
                lr, hr = lr.flip(2), hr.flip(2)
            xs.append(lr)
            ys.append(hr)
        return torch.stack(xs) * 255.0, torch.stack(ys)


class L1On255(torch.nn.Module):
    """LVisionSR.forward divides the model output by 255 before the L1 criterion (vision_sr_module.py:49-53)."""

    def forward(self, out, target):
        return F.l1_loss(out / 255.0, target)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=100, help="LR warm-up steps (the config: 500 of ~1e5)")
    ap.add_argument("--bits", type=int, default=4,
                    help="weight / activation / calibration bit width.  The config's own 2 bits, calibrated directly "
                         "at 2 bits as the YAML says, start from a destroyed image (8 dB; ~10 dB after 3000 steps on either "
                         "side); 4 and 8 bits start at 25 / 32.6 dB and are what a few hundred steps can compare")
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--act-estimator", default="LSQ", choices=["LSQ", "STE"],
                    help="LSQ: nothing random on either side.  STE: what the wrap rule builds (random sign streams, "
                         "different by construction between the two sides)")
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    crops = Crops(TRAIN)
    held = []
    for n in HELD_OUT:
        hr = load_rgb(n).to(DEV)[None]
        held.append((downscale(hr), hr))
    fp = pretrained_rfdn().to(DEV)
    fp_psnr, fp_each = psnr_y(fp, held)
    bic = sum(float(10 * torch.log10(1.0 / (to_luminance(F.interpolate(lr, scale_factor=4, mode="bicubic",
              align_corners=False).clamp(0, 1)) - to_luminance(hr)).square().mean())) for lr, hr in held) / len(held)
    out = {"task": "x4 super-resolution, pretrained RFDN (the reference's data/models/RFDN_AIM.pth), PSNR on luminance "
                   "(vision_sr_module.py:151-158) over 3 held-out photographs; QAT on random 24x24 LR crops of 5 others",
           "recipe": f"config/gdnsq_config_rfdn_lsq_w2a2.yaml at W{args.bits}A{args.bits}: per-channel LSQ weights, "
                     f"{args.act_estimator} activations, {args.bits}-bit calibration, L1, RAdam 5e-4, batch 24, "
                     f"{args.steps} steps (warm-up {args.warmup})",
           "pretrained_fp_psnr_y": round(fp_psnr, 4), "bicubic_psnr_y": round(bic, 4),
           "held_out": list(HELD_OUT)}
    res = {"hip": [], "oracle": []}
    for side, layers in (("hip", None), ("oracle", ORACLE_LAYERS)):
        for rep in range(args.seeds):
            torch.manual_seed(100 + rep)
            ops.manual_seed(100 + rep)
            cfg = QATConfig(qscheme=M.QScheme.PER_CHANNEL, qnmethod=M.QNMethod.LSQ, act_bit=args.bits,
                            weight_bit=args.bits, calib_act_bit=args.bits, calib_weight_bit=args.bits,
                            excluded_layers=("fea_conv", "upsampler.0"), distillation=False, learning_rate=5e-4,
                            warmup=args.warmup, criterion=L1On255())
            g = torch.Generator().manual_seed(7 + rep)
            calib = [crops.batch(24, g)[0] for _ in range(2)]
            mm = (lambda t: torch.stack(list(t.aminmax()))) if layers is not None else None
            tr = QATTrainer(copy.deepcopy(fp), cfg, DEV, calib_batches=calib, layers=layers, minmax_fn=mm,
                            distributed=False)
            for m in tr.net.modules():           # the wrap rule builds STE activation quantizers
                if hasattr(m, "log_act_s"):
                    if hasattr(m, "Q"):
                        m.Q.qnmethod = M.QNMethod[args.act_estimator]
                    else:
                        m.qnmethod = args.act_estimator
            post_calib, _ = psnr_y(tr.net, held)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                x, y = crops.batch(24, g)
                last = tr.train_step(x, y)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            final, each = psnr_y(tr.net, held)
            res[side].append({"psnr_y": round(final, 4), "per_image": [round(v, 4) if isinstance(v, float) else v for v in each],
                              "post_calibration_psnr_y": round(post_calib, 4), "final_loss": round(float(last), 6),
                              "ms_per_step": round(dt / args.steps * 1e3, 2)})
            print(f"[rfdn_psnr_parity] {side} seed {rep + 1}/{args.seeds}: {res[side][-1]['psnr_y']} dB", file=sys.stderr, flush=True)
            del tr
    out["hip"], out["oracle"] = res["hip"], res["oracle"]
    mh = sum(r["psnr_y"] for r in res["hip"]) / args.seeds
    mo = sum(r["psnr_y"] for r in res["oracle"]) / args.seeds
    out["mean_psnr_y_hip"], out["mean_psnr_y_oracle"] = round(mh, 4), round(mo, 4)
    out["psnr_difference_db"] = round(mh - mo, 4)
    out["per_seed_difference_db"] = [round(a["psnr_y"] - b["psnr_y"], 4) for a, b in zip(res["hip"], res["oracle"])]
    if args.seeds > 1:       # the same seed on both sides shares crops and calibration: a paired comparison
        import statistics
        d = out["per_seed_difference_db"]
        out["paired_difference_db_mean_stderr"] = [round(statistics.mean(d), 4),
                                                   round(statistics.stdev(d) / len(d) ** 0.5, 4)]
        out["stdev_psnr_y_hip"] = round(statistics.stdev(r["psnr_y"] for r in res["hip"]), 4)
        out["stdev_psnr_y_oracle"] = round(statistics.stdev(r["psnr_y"] for r in res["oracle"]), 4)
    out["post_calibration_difference_db"] = [round(a["post_calibration_psnr_y"] - b["post_calibration_psnr_y"], 4)
                                             for a, b in zip(res["hip"], res["oracle"])]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
