#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only): turns the one real checkpoint the reference ships --
/root/reference/data/models/RFDN_AIM.pth, the pretrained RFDN that config/gdnsq_config_rfdn_lsq_w2a2.yaml starts
from (`cpt_url: file://data/models/RFDN_AIM.pth`; every other checkpoint in that directory is a 130-byte LFS stub)
-- into a data fixture: tests/golden/rfdn_aim_weights.npz, one float32 array per state-dict key.  The reference
does not exist on the GPU box, so the end-task parity tool (tools/rfdn_psnr_parity.py) and its test read the fixture.
Data only: tensors, no code."""
import os
import sys

import numpy as np
import torch

SRC = "/root/reference/data/models/RFDN_AIM.pth"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                   "rfdn_aim_weights.npz")


def main():
    sd = torch.load(SRC, map_location="cpu", weights_only=True)
    arrays = {k: v.detach().to(torch.float32).numpy() for k, v in sd.items()}
    np.savez_compressed(DST, **arrays)
    n = sum(a.size for a in arrays.values())
    print(f"{DST}: {len(arrays)} tensors, {n} parameters, {os.path.getsize(DST) / 1e6:.2f} MB", file=sys.stderr)


if __name__ == "__main__":
    main()
