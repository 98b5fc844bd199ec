"""The sign stream of include/mhaq_fq.h restated in numpy (test infrastructure): Philox4x32-10, counter =
{lo(c), hi(c), lo(offset), hi(offset)}, key = {lo(seed), hi(seed)}, c = (f >> 9) * 256 + (f & 255) with f = i >> 2;
element i takes bit 4 * ((f >> 8) & 1) + (i & 3) of the first output word; r = bit ? +0.5 : -0.5."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def first_word(counter64, offset, seed):
    """First 32-bit output word of Philox4x32-10 for an array of 64-bit counters."""
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
    return c0.astype(np.uint32)


def signs(n, seed, offset):
    """int8 +1 / -1 for elements 0 .. n-1 of the (seed, offset) stream (what mhaq_fq_fill_r writes)."""
    i = np.arange(n, dtype=np.uint64)
    f = i >> np.uint64(2)
    call = (f >> np.uint64(9)) * np.uint64(256) + (f & np.uint64(255))
    uniq, inv = np.unique(call, return_inverse=True)
    word = first_word(uniq, offset, seed)[inv].astype(np.uint64)
    bit = (word >> (np.uint64(4) * ((f >> np.uint64(8)) & np.uint64(1)) + (i & np.uint64(3)))) & np.uint64(1)
    return np.where(bit == 1, 1, -1).astype(np.int8)
