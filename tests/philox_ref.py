"""The sign stream of include/mhaq_fq.h (layout v3) restated in numpy (test infrastructure): Philox4x32-10, counter =
{lo(c), hi(c), lo(offset), hi(offset)}, key = {lo(seed), hi(seed)}, c = i >> 7: one call covers 128 consecutive elements;
element i takes bit (j & 31) of output word (j >> 5), j = i & 127; r = bit ? +0.5 : -0.5."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def words(counter64, offset, seed):
    """The four 32-bit output words of Philox4x32-10 for an array of 64-bit counters: uint32 [4, n]."""
    c = np.asarray(counter64, dtype=np.uint64)
    c0, c1 = c & MASK, c >> np.uint64(32)
    c2 = np.full_like(c0, np.uint64(offset & 0xFFFFFFFF))
    c3 = np.full_like(c0, np.uint64((offset >> 32) & 0xFFFFFFFF))
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & MASK
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3]).astype(np.uint32)


def signs(n, seed, offset):
    """int8 +1 / -1 for elements 0 .. n-1 of the (seed, offset) stream (what mhaq_fq_fill_r writes)."""
    i = np.arange(n, dtype=np.uint64)
    ncalls = (n + 127) >> 7
    w = words(np.arange(ncalls, dtype=np.uint64), offset, seed)          # [4, ncalls]
    j = i & np.uint64(127)
    word = w[(j >> np.uint64(5)).astype(np.int64), (i >> np.uint64(7)).astype(np.int64)].astype(np.uint64)
    bit = (word >> (j & np.uint64(31))) & np.uint64(1)
    return np.where(bit == 1, 1, -1).astype(np.int8)
