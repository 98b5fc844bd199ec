#!/usr/bin/env python3
"""Fake-quant path only: the quantizer set of every BASELINE config -- all NoisyAct tensors and
all weight tensors, fused forward + backward, 20 B/elem algorithmic -- through the raw C ABI, through the product path
(modules + compiled autograd nodes, eager from an idle stream) and replayed as a hipGraph.  One JSON line per config
(tools/fq_sets.py does the measuring; bench.py reports the same numbers in its `configs` block)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools.fq_sets import CONFIGS, cpu_fake_quant_set, measure_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default=",".join(CONFIGS))
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--cpu", action="store_true", help="also time the eager CPU oracle on the ResNet-20 batch-128 set")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for key in args.configs.split(","):
        print(json.dumps({"config": key, **measure_config(key, dev, reps=args.reps)}), flush=True)
    if args.cpu:
        print(json.dumps({"config": "cpu_fake_quant_set", **cpu_fake_quant_set()}), flush=True)


if __name__ == "__main__":
    main()
