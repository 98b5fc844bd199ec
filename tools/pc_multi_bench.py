#!/usr/bin/env python3
"""The multi-tensor weight launches of a training step on their own: ResNet-18's 16 per-channel
weight layers (11.0 M weights, 44 MB) in the trainer's form -- ONE model-wide forward launch (mhaq_fq_wlayer_fwd_multi)
and the backward in groups of layers cut from the end of the model (mhaq_fq_wlayer_bwd_group; AEWGS with its group
statistics launch) -- through the raw C ABI, HIP-event timing per launch kind.
    python3 tools/pc_multi_bench.py [STE|LSQ|AEWGS] [model]          model: resnet18 (default), resnet20, resnet20_pt, rfdn
Under rocprofv3 (program directly after --):
    rocprofv3 --kernel-trace --stats ... / rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY ...
"""
import ctypes
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mhaq_amd import _lib, ops  # noqa: E402
from mhaq_amd.multi import backward_groups  # noqa: E402
from tools.fq_sets import METHOD_ID, _Desc, weight_shapes  # noqa: E402


def trace(L, model, method, tot_c, cho, row, co, fwd, bwd, groups, nset):
    """Phase stamps of every workgroup of ONE cold launch (100 MHz wall clock, first wave): where the time of a launch goes."""
    import numpy as np
    L.mhaq_debug_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]

    def read(n):
        buf = np.zeros((n, 8), dtype=np.uint64)
        assert L.mhaq_debug_trace_read(buf.ctypes.data, n) == 0
        return buf

    def report(name, buf, rows):
        t = buf[:, :6].astype(np.int64)
        t0 = t[:, 0].min()
        rel = (t - t0) * 0.01                                   # us since the first workgroup started
        span = rel[:, 5].max()
        ph = np.diff(rel, axis=1)                               # desc, load, reduce, store issue, store ack
        print(f"== {name}: {len(buf)} workgroups, launch span {span:.2f} us (first entry -> last store acknowledged)")
        names = ["descriptor", "row loads", "compute+reduce", "compute+store issue", "store ack"]
        for k, nm in enumerate(names):
            print(f"   {nm:22s} median {np.median(ph[:, k]):6.2f}  p90 {np.percentile(ph[:, k], 90):6.2f}  max {ph[:, k].max():6.2f} us")
        life = rel[:, 5] - rel[:, 0]
        print(f"   workgroup lifetime     median {np.median(life):6.2f}  p90 {np.percentile(life, 90):6.2f}  max {life.max():6.2f} us")
        # start-time profile: how many workgroups have started / finished by t
        for q in (0.5, 1, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 24):
            if q > span + 2:
                break
            started = (rel[:, 0] <= q).sum()
            done = (rel[:, 5] <= q).sum()
            byt = rows[rel[:, 5] <= q].sum() * 4
            print(f"   t = {q:5.1f} us: started {started:5d}  finished {done:5d}  alive {started - done:5d}  rows finished {byt / 1e6:6.1f} MB")
        hw = buf[:, 6]
        xcc = (hw >> np.uint64(32)) & np.uint64(0xf)
        cu = (hw >> np.uint64(8)) & np.uint64(0xf)
        se = (hw >> np.uint64(13)) & np.uint64(0x7)
        ids = xcc * np.uint64(1000) + se * np.uint64(100) + cu
        per = np.bincount(np.unique(ids, return_inverse=True)[1])
        print(f"   distinct (xcc, se, cu) ids {len(per)}; workgroups per id min {per.min()} median {int(np.median(per))} max {per.max()}")
        print("   first start per XCD (us): " + " ".join(f"{rel[xcc == x, 0].min():.2f}" for x in np.unique(xcc)) +
              "   last end per XCD: " + " ".join(f"{rel[xcc == x, 5].max():.2f}" for x in np.unique(xcc)))
        np.save(f"gpurun_out/trace_{model}_{method}_{name.split()[0]}.npy", buf)

    rows_all = np.concatenate([np.full(c, r) for c, r in zip(co, row)])
    torch.cuda.synchronize()
    read(8192)                                                  # clear
    # the LAST of a train of back-to-back launches (each overwrites the stamps): a launch into an idle GPU starts its
    # XCDs up to 5 us apart, which a training step's stream of kernels does not see
    for k in range(nset):
        fwd(k)
    report("forward", read(tot_c), rows_all)
    for gi, (a_, b_) in enumerate(groups):
        for k in range(nset):
            bwd(k, gi, False)
        n = sum(co[a_:b_])
        report(f"backward{gi} (layers {a_}..{b_ - 1})", read(n), rows_all[cho[a_]:cho[a_] + n])


def main():
    method = sys.argv[1] if len(sys.argv) > 1 else "STE"
    model = sys.argv[2] if len(sys.argv) > 2 else "resnet18"
    per_tensor = model.endswith("_pt")
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    wsh = weight_shapes(model[:-3] if per_tensor else model)
    gen = torch.Generator(device=dev).manual_seed(4)
    # several parameter sets so that back-to-back repetitions do not re-read the same 44 MB out of the Infinity Cache
    nset = int(os.environ.get("MHAQ_PCMB_NSET", "6"))       # 1 = cache-warm: the same 44 MB every launch (in-step-like)
    sets = []
    mid = METHOD_ID[method]
    co = [1 if per_tensor else s[0] for s in wsh]
    row = [math.prod(s) // c for s, c in zip(wsh, co)]
    eo, cho = [], []
    e = c_ = 0
    for a_, b_ in zip(co, row):
        eo.append(e)
        cho.append(c_)
        e += a_ * b_
        c_ += a_
    tot_e, tot_c, max_row = e, c_, max(row)
    groups = backward_groups([a_ * b_ for a_, b_ in zip(co, row)], [mid] * len(wsh), 4 << 20)
    for _ in range(nset):
        Ws = [torch.randn(s, device=dev, generator=gen) * math.sqrt(2.0 / (s[1] * 9)) for s in wsh]
        Gs = [torch.randn(s, device=dev, generator=gen) for s in wsh]
        lss = []
        for w, c in zip(Ws, co):
            mn, mx = ops.row_minmax(w)
            if per_tensor:
                mn, mx = mn.min(), mx.max()
            lss.append(torch.clamp(torch.log2((mx - mn) / 15), min=-12.0).reshape(c).contiguous())
        arr = (_Desc * len(wsh))()
        for i in range(len(wsh)):
            arr[i] = _Desc(Ws[i].data_ptr(), lss[i].data_ptr(), None, None, co[i], row[i], eo[i], cho[i])
        ftable = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        gplans = []
        for a_, b_ in groups:
            garr = (_Desc * (b_ - a_))()
            for k, i in enumerate(range(a_, b_)):
                garr[k] = _Desc(Ws[i].data_ptr(), None, Gs[i].data_ptr(), None, co[i], row[i], eo[i] - eo[a_],
                                cho[i] - cho[a_])
            gco = sum(co[i] for i in range(a_, b_))
            gel = sum(co[i] * row[i] for i in range(a_, b_))
            gplans.append((torch.frombuffer(bytearray(bytes(garr)), dtype=torch.uint8).to(dev), b_ - a_, gco,
                           max(row[i] for i in range(a_, b_)), cho[a_], torch.empty(gel, device=dev),
                           torch.empty(gco, device=dev), torch.empty(3, gco, device=dev), gel))
        sets.append((Ws, Gs, lss, ftable, gplans, torch.empty(tot_e, device=dev), torch.empty(4, tot_c, device=dev)))
    off = [0]

    def fwd(k):
        _, _, _, ftable, _, wq_all, aux_all = sets[k % nset]
        assert L.mhaq_fq_wlayer_fwd_multi(ftable.data_ptr(), len(wsh), tot_c, max_row, wq_all.data_ptr(),
                                          aux_all.data_ptr(), st) == 0

    def bwd(k, gi, with_stats):
        _, _, _, _, gplans, _, aux_all = sets[k % nset]
        tab, n, gco, grow, c0, gwb, glb, stats, _ = gplans[gi]
        off[0] += 1
        if with_stats:      # the data-parallel trainer's form: statistics launch, (all-reduce), apply
            assert L.mhaq_fq_wlayer_aewgs_stats_group(tab.data_ptr(), n, gco, aux_all.data_ptr() + 4 * c0, tot_c,
                                                      stats.data_ptr(), st) == 0
        assert L.mhaq_fq_wlayer_bwd_group(tab.data_ptr(), n, gco, grow, aux_all.data_ptr() + 4 * c0, tot_c,
                                          gwb.data_ptr(), glb.data_ptr(), mid, stats.data_ptr() if with_stats else None,
                                          1234, off[0], None, st) == 0

    def timed(fn, reps=40):
        for k in range(nset):
            fn(k)
        torch.cuda.synchronize()
        out = []
        for r in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for k in range(reps):
                fn(r * reps + k)
            b.record()
            torch.cuda.synchronize()
            out.append(a.elapsed_time(b) / reps * 1e3)
        return sorted(out)[3]
    for k in range(nset):
        fwd(k)
    if os.environ.get("MHAQ_PCMB_TRACE"):       # a -DMHAQ_TRACE build (tools/variants.sh trace "-DMHAQ_TRACE") as MHAQ_FQ_LIB
        trace(L, model, method, tot_c, cho, row, co, fwd, bwd, groups, nset)
        return
    print(f"{model} weights, {method}: {len(wsh)} layers, {tot_c} rows, {tot_e} weights ({tot_e * 4 / 1e6:.1f} MB), rows of "
          f"{min(row)}..{max(row)} floats; backward groups {groups}", flush=True)
    tf = timed(fwd)
    print(f"  forward, one launch           {tf:7.2f} us  {8 * tot_e / tf / 1e3:7.0f} GB/s (8 B/elem)", flush=True)
    tot = tf
    for gi, (a_, b_) in enumerate(groups):
        gel = sets[0][4][gi][8]
        tb = timed(lambda k: bwd(k, gi, False))
        tot += tb
        print(f"  backward group {gi} (layers {a_}..{b_ - 1}, {gel * 4 / 1e6:5.1f} MB) {tb:7.2f} us  {12 * gel / tb / 1e3:7.0f} GB/s (12 B/elem)",
              flush=True)
        if method == "AEWGS":
            ts = timed(lambda k: bwd(k, gi, True))
            print(f"     with the group statistics launch (data-parallel form) {ts:7.2f} us  {20 * gel / ts / 1e3:7.0f} GB/s (20 B/elem)",
                  flush=True)
    print(f"  forward + grouped backward    {tot:7.2f} us  {20 * tot_e / tot / 1e3:7.0f} GB/s (20 B/elem) = "
          f"{20 * tot_e / tot / 1e3 / 8000:.3f} of 8 TB/s", flush=True)


if __name__ == "__main__":
    main()
