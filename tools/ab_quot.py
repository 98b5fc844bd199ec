#!/usr/bin/env python3
"""A/B of two builds of the library on the per-channel EWGS / AEWGS backward: every output must be the same bits
(used when the IEEE divisions by the wave-uniform scale became Markstein corrections).  usage:
  python tools/ab_quot.py <libA.so> <libB.so>"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mhaq_amd import _lib


def load(path):
    L = C.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(L, name):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
    return L


A, B = load(sys.argv[1]), load(sys.argv[2])
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
bad = cases = 0
g = torch.Generator(device=dev).manual_seed(0)
for trial in range(120):
    co = int(torch.randint(1, 200, (1,)))
    row = int(torch.randint(1, 3000, (1,)))
    if trial % 3 == 0:
        row = row // 4 * 4 + 4
    if trial % 10 == 0:
        co, row = 512, 4608
    w = torch.randn(co, row, device=dev, generator=g) * 10 ** float(torch.empty(1).uniform_(-3, 1))
    mag = 10 ** torch.empty(co, 1, device=dev).uniform_(-36, 30) if trial % 2 else torch.ones(co, 1, device=dev)
    G = torch.randn(co, row, device=dev, generator=g) * mag
    if trial % 5 == 0:
        G[:, ::7] = 0
    ls = torch.empty(co, device=dev).uniform_(-9, -1)
    if trial % 4 == 0:
        ls = ls.round()                       # power-of-two scales
    if trial % 11 == 0:                       # significand all ones: the degenerate-scale fallback
        ls = torch.log2(torch.full((co,), 0.124999992549419403076171875, device=dev))
    aux = torch.empty(4, co, device=dev)
    wq = torch.empty_like(w)
    assert A.mhaq_fq_wlayer_fwd(w.data_ptr(), wq.data_ptr(), ls.data_ptr(), co, row, aux[0].data_ptr(),
                                aux[1].data_ptr(), aux[2].data_ptr(), aux[3].data_ptr(), st) == 0
    glwq = torch.randn(co, device=dev, generator=g)
    for method in (1, 2):
        for stats_given in (False, True):
            stats = None
            if method == 2 and stats_given:
                stats = torch.empty(3, co, device=dev)
                assert A.mhaq_fq_pc_aewgs_stats(w.data_ptr(), G.data_ptr(), aux[0].data_ptr(), aux[1].data_ptr(), co,
                                                row, stats.data_ptr(), st) == 0
            outs = []
            for L in (A, B):
                gw = torch.full_like(w, float("nan"))
                gls = torch.full((co,), float("nan"), device=dev)
                assert L.mhaq_fq_wlayer_bwd(w.data_ptr(), G.data_ptr(), gw.data_ptr(), gls.data_ptr(),
                                            aux[0].data_ptr(), aux[1].data_ptr(), aux[2].data_ptr(), glwq.data_ptr(),
                                            co, row, method, stats.data_ptr() if stats is not None else None, None,
                                            None, 11, trial + 1, None, st) == 0
                outs.append((gw, gls))
            torch.cuda.synchronize()
            cases += 1
            same = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(*outs))
            if not same:
                bad += 1
                d = (outs[0][0] != outs[1][0]) & ~(outs[0][0].isnan() & outs[1][0].isnan())
                print(f"MISMATCH trial {trial} method {method} stats {stats_given} co {co} row {row}: "
                      f"{int(d.sum())} gw elements, gls equal {torch.equal(outs[0][1], outs[1][1])}", flush=True)
print(f"ab_quot: {cases} cases, {bad} mismatching")
sys.exit(1 if bad else 0)
