#!/usr/bin/env python3
"""Four ranks on one GPU over gloo (RCCL refuses several ranks per device): the DDP trainer with the HIP layers and the
grouped AEWGS exchange at a world size above two -- collective order, parameter sync (a rehearsal, not a test)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import tests.test_gpu_ddp_two_ranks as T
    out = T._spawn(T._w_trainer, world=4)
    print("losses per rank:", [out[r][0] for r in range(4)])
    s = [out[r][1] for r in range(4)]
    a = [out[r][2] for r in range(4)]
    print("parameter sums per rank:", s)
    assert all(abs(x - s[0]) <= 1e-6 * a[0] for x in s), "ranks out of sync"
    T._spawn(T._w_aewgs_group, world=4)
    print("grouped AEWGS exchange at world size 4: ok")


if __name__ == "__main__":
    main()
