"""Loader for the committed golden fixtures (tests/golden/*.npz, made by oracle/gen_golden.py)."""
import os
from collections import defaultdict

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_cases(fname):
    z = np.load(os.path.join(GOLDEN, fname))
    cases = defaultdict(dict)
    for k in z.files:
        case, field = k.split("__", 1)
        cases[case][field] = z[k]
    return dict(cases)


def T(a, device="cpu"):
    return torch.from_numpy(np.asarray(a, dtype=np.float32).copy()).to(device)


def r_from_sign(a, device="cpu"):
    """int8 sign fixture (+-1) -> the +-0.5 tensor of gdnsq.py:54."""
    return torch.from_numpy(a.astype(np.float32) * 0.5).to(device)


def bit_equal(a, b):
    """Bit-exact fp32 comparison (treats +0/-0 as different, NaN==NaN if same bits)."""
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    b = np.ascontiguousarray(np.asarray(b, dtype=np.float32))
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def value_equal(a, b):
    """Exact value equality (+0 == -0)."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    return a.shape == b.shape and np.array_equal(a, b)


def max_ulp(a, b):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32)).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(np.asarray(b, dtype=np.float32)).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return int(np.max(np.abs(a - b))) if a.size else 0


def off_extremes_mask(w, per_channel, also_max=False):
    """Elements of a weight tensor that are NOT their group's minimum (nor, with also_max, its maximum): the
    amin (amax) backward adds a share of a REDUCED gradient only at the extremes, everywhere else gW is the
    elementwise (G*s [+ estimator])/s of STE / LSQ / EWGS and must be value-equal to the reference."""
    w = np.asarray(w, dtype=np.float32)
    if per_channel:
        w2 = w.reshape(w.shape[0], -1)
        mask = w2 != w2.min(axis=1, keepdims=True)
        if also_max:
            mask &= w2 != w2.max(axis=1, keepdims=True)
        return mask.reshape(w.shape)
    mask = w != w.min()
    if also_max:
        mask &= w != w.max()
    return mask


def exact_off_extremes(gw, ref, w, per_channel, also_max=False):
    mask = off_extremes_mask(w, per_channel, also_max)
    gw, ref = np.asarray(gw, dtype=np.float32), np.asarray(ref, dtype=np.float32)
    return gw.shape == ref.shape and np.array_equal(gw[mask], ref[mask])


def bits_checksum(a):
    """Three 64-bit checksums (mod 2^64) of the BIT PATTERNS of an fp32 array: sum, xor, position-weighted sum -- a changed,
    missing or permuted element changes at least one of them.  For outputs too large to commit (tests/golden/big_cases.npz)."""
    u = np.ascontiguousarray(np.asarray(a, dtype=np.float32)).view(np.uint32).astype(np.uint64).ravel()
    idx = np.arange(1, u.size + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return np.array([u.sum(dtype=np.uint64), np.bitwise_xor.reduce(u), (u * idx).sum(dtype=np.uint64)], dtype=np.uint64)


def big_inputs(seed, n, scale):
    """The (x, g) / (w, G) pair of a full-size fixture: numpy's default_rng (PCG64) is bit-reproducible on any machine, so the
    GPU box regenerates here what oracle/gen_golden_big.py fed the reference."""
    rng = np.random.default_rng(int(seed))
    x = rng.standard_normal(int(n), dtype=np.float32) * np.float32(scale)
    g = rng.standard_normal(int(n), dtype=np.float32)
    return x, g
