#!/usr/bin/env python3
"""bench.py -- QAT images/sec on N MI355X + roofline of the dominant fake-quant kernel.

Contract (one JSON line on rank 0):
  metric/value   QAT images/s: one step = one data-parallel QAT training step (student forward
                 through the HIP fake-quant path, FP-teacher forward, PotentialLoss, backward with
                 DDP bucketed all-reduce over RCCL, RAdam step) of ResNet-18 on synthetic 224x224
                 data, per-GPU batch 250 (config/gdnsq_config_resnet18_imagenet_aewgs_w1a1.yaml;
                 BASELINE.json configs[3]; `--qnmethod STE` gives configs[2]).  Inputs are resident
                 in HBM before the timed region; weak scaling (per-GPU work fixed).
  roofline       the dominant fake-quant kernel, pt_bwd_kernel (activation backward, 12 B/elem
                 algorithmic: read x, read g, write gx), timed alone with HIP events on the launch
                 stream over the ResNet-18 layer-1 activation [250,64,56,56] (50.2 M elements),
                 buffers rotated to defeat the 256 MB Infinity Cache.
  cpu_baseline   the CPU oracle (oracle/ref_layers.py over oracle/fq_eager.py: the eager port of the
                 reference) running the SAME training step on the host cores at the SAME batch (2 timed steps).
  gpu_eager_baseline   the same step, same batch, with the oracle's eager layers on the SAME GPU (N = 1): the
                 op chain the path replaces, one box, one run.  Both are reported baselines, never the target.

Launch: python bench.py --gpus N          N = 1: this process.  N > 1 without WORLD_SIZE in the environment:
                                          this process starts N ranks itself (one per GPU, RCCL) through
                                          `python -m torch.distributed.run` BEFORE anything touches the GPU, relays
                                          rank 0's JSON line and exits with the launcher's status -- what Lightning's
                                          DDP launcher does for the reference (training/trainer.py:92-97).
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
               --master-port P bench.py --gpus N --steps K --warmup W       (the driver's form: WORLD_SIZE is set,
                                          every rank runs main() directly)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _requested_gpus(argv):
    """--gpus from the raw argument list (read before torch is imported)."""
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


def launch_ranks(argv, n):
    """`bench.py --gpus N` as typed by hand or by a driver that does not wrap it in torchrun: start N ranks as
    fresh child processes (this parent has not imported torch, so no process that has initialised the GPU is
    ever replaced or forked), pass stdout's JSON line through, everything else to stderr."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    print(f"[bench launcher] starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    line = None
    for ln in proc.stdout:
        if ln.startswith('{"metric"') or ln.startswith('{"roofline"'):     # the result line (or --roofline-only's)
            line = ln.rstrip("\n")
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
        print("[bench launcher] the ranks exited without a result line", file=sys.stderr, flush=True)
    return rc


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and _requested_gpus(sys.argv[1:]) > 1:
    sys.exit(launch_ranks(sys.argv[1:], _requested_gpus(sys.argv[1:])))



def seed_miopen_user_db():
    """cudnn.benchmark=True (what Lightning gives the reference) makes MIOpen search every convolution shape
    on first use: ~5 minutes on a fresh box for the 40-odd shapes of this step.  mhaq_amd/miopen_db/ holds the
    outcome of that search on an MI355X (MIOpen's own user find-db / perf-db text files: which MIOpen solver
    won per shape); each process gets a private writable copy so the search is skipped.  The convolutions
    themselves stay MIOpen's -- nothing here touches the fake-quant path.  MHAQ_NO_MIOPEN_DB=1 disables it."""
    if os.environ.get("MHAQ_NO_MIOPEN_DB") == "1" or "MIOPEN_USER_DB_PATH" in os.environ:
        return None
    import glob
    import shutil
    import tempfile
    src = os.path.join(ROOT, "mhaq_amd", "miopen_db")
    files = glob.glob(os.path.join(src, "*.txt"))
    if not files:
        return None
    dst = tempfile.mkdtemp(prefix=f"mhaq_miopen_{os.environ.get('LOCAL_RANK', '0')}_")
    for f in files:
        shutil.copy(f, dst)
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    import atexit
    atexit.register(shutil.rmtree, dst, ignore_errors=True)
    return dst


MIOPEN_DB_DIR = seed_miopen_user_db()      # before torch touches MIOpen
# dmabuf IPC: RCCL across processes needs it on this pool (the driver's environment exports it already; a rank started
# by any other launcher must not depend on that).  Read when the HSA runtime initialises, i.e. at the first GPU call.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

T0 = time.perf_counter()


def log(msg):
    """Progress on stderr (stdout carries only the JSON line)."""
    print(f"[bench +{time.perf_counter() - T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def start_heartbeat(period=60.0):
    """A line on stderr every minute: a cold MIOpen cache can keep the first step silent for minutes."""
    import threading

    def beat():
        while True:
            time.sleep(period)
            log("... still running")
    threading.Thread(target=beat, daemon=True).start()


HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_COPY_GBPS = 6290.0  # the same guide's measured float4-copy rate ("6.29 TB/s measured, 79 %"): the practical ceiling
# the dominant kernel: the activation backward exactly as the training step instantiates it (mhaq_fq_act_bwd)
# template arguments: <METHOD = STE, RSIGN = false, ALIGNED, COUNT = false, ACT, BIG (the occupancy form for tensors of kBwdBigElems = 20 Mi elements and more, fq_pt.hip)>
DOMINANT_KERNEL = "mhaq::pt_bwd_kernel<0, false, true, false, true, true>"


def kernel_source_hash():
    """sha256 over the sources of the streaming kernels: ties a PMC measurement to the code it was taken on."""
    from tools.kernel_hash import kernel_source_hash as h
    return h()


def profiled_traffic():
    """HBM bytes per launch of the dominant kernel from the newest profiles/rNN_traffic.json (written by
    tools/summarize_prof.py from the two rocprofv3 --pmc passes: 2 x FETCH_SIZE -- gfx950 counts 16 B/lane
    streaming reads at 1/2 -- + WRITE_SIZE).  None when there is no such file or when the kernel sources have
    changed since it was measured (tests/test_profiles_cpu.py fails in that case: re-run tools/profile_bench.sh)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as fh:
        rec = json.load(fh)
    src = os.path.basename(files[-1])
    if rec.get("kernel_source_hash") != kernel_source_hash():
        return None, f"{src} is stale (kernel sources changed since the PMC passes)"
    ent = rec.get("kernels", {}).get(DOMINANT_KERNEL)
    return (None, f"{src} has no entry for {DOMINANT_KERNEL}") if ent is None else (int(ent["hbm_bytes"]), src)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=250, help="per-GPU batch (reference config: 250)")
    ap.add_argument("--image", type=int, default=224)
    ap.add_argument("--qnmethod", default="AEWGS", choices=["STE", "LSQ", "AEWGS", "EWGS"])
    ap.add_argument("--no-distillation", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=0,
                    help="batch of the CPU baseline step; 0 = the GPU line's own --batch (the same workload: ~8 s/step "
                         "at 250 on a 16-CPU share)")
    ap.add_argument("--no-gpu-eager-baseline", action="store_true",
                    help="skip the leg that times the same step with the eager op chain on the same GPU")
    ap.add_argument("--cpu-steps", type=int, default=2, help="minimum number of timed CPU steps")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="keep timing CPU steps until this much CPU work has been sampled (bounded sample)")
    ap.add_argument("--kernel-reps", type=int, default=30)
    ap.add_argument("--multi-tensor-weights", action="store_true",
                    help="quantize all weights in one launch per direction (single-GPU option)")
    ap.add_argument("--no-cudnn-benchmark", action="store_true",
                    help="disable torch.backends.cudnn.benchmark (MIOpen algorithm search; Lightning enables it)")
    ap.add_argument("--nchw", action="store_true",
                    help="keep the reference's NCHW memory format (default: channels_last, see docs/NOTEBOOK.md section 6, Memory format)")
    ap.add_argument("--no-teacher-overlap", action="store_true",
                    help="run the frozen teacher forward on the main stream instead of a second HIP stream")
    ap.add_argument("--no-multi-weight-forward", action="store_true",
                    help="per-layer weight forward launches instead of one model-wide launch per step (A/B)")
    ap.add_argument("--weight-group-elems", type=int, default=4 << 20,
                    help="weight backward in groups of consecutive layers of at least this many weights (0: one "
                         "launch, and under AEWGS + DDP one statistics all-reduce, per layer)")
    ap.add_argument("--student-high-priority", action="store_true",
                    help="run the step on a priority -1 HIP stream (teacher stream stays at 0): measured option")
    ap.add_argument("--capture-graph", nargs="?", const="on", default="off", choices=["off", "on", "auto"],
                    help="replay forward + loss + backward as one hipGraph (single GPU).  auto: only if the host needs "
                         "> 80 %% of a step's wall time to enqueue it (the ResNet-18 batch-250 step is GPU-bound: "
                         "stays eager); see DESIGN.md section 6 / docs/NOTEBOOK.md section 6 \"hipGraph step\"")
    ap.add_argument("--roofline-only", action="store_true", help="run only the kernel legs (PMC passes)")
    ap.add_argument("--no-roofline-set", action="store_true",
                    help="skip the 16-tensor activation-set leg (6.7 GB of buffers, ~2 s)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the per-BASELINE-config quantizer-set legs (ResNet-20 b128 / b1000 per-tensor, RFDN "
                         "reference and stress shapes; ~10 s)")
    return ap.parse_args()


# ------------------------------------------------------------------------------ roofline leg
def kernel_roofline(dev, reps):
    """The activation quantizer of ResNet-18 layer 1 exactly as the training step runs it: mhaq_fq_act_fwd
    (pt_fwd_kernel<LOGP>), mhaq_fq_act_bwd_partials (pt_bwd_kernel<STE, ACT>: the dominant kernel) and
    mhaq_fq_act_bwd (with its own finalize), each timed alone."""
    import ctypes

    from mhaq_amd import _lib
    L = _lib.lib()
    shape = (250, 64, 56, 56)
    n = 250 * 64 * 56 * 56
    nbuf = 3
    gen = torch.Generator(device=dev).manual_seed(1)
    xs = [torch.randn(shape, device=dev, generator=gen) * 2 for _ in range(nbuf)]
    gs = [torch.randn(shape, device=dev, generator=gen) for _ in range(nbuf)]
    ys = [torch.empty(shape, device=dev) for _ in range(nbuf)]
    import math
    ls = torch.tensor([math.log2(0.2371)], device=dev)     # 16 levels over [-1.9, 1.66]: both clamp sides active
    lq = ls + 4
    b = torch.tensor([-1.9], device=dev)
    params = torch.empty(5, device=dev)
    grads = torch.empty(3, device=dev)
    nb = L.mhaq_fq_act_bwd_workspace_bytes(n)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    nparts = ctypes.c_int32(0)

    def fwd(i):
        k = i % nbuf
        return L.mhaq_fq_act_fwd(xs[k].data_ptr(), ys[k].data_ptr(), n, ls.data_ptr(), lq.data_ptr(), b.data_ptr(),
                                 params.data_ptr(), None, None, None, 0, st)

    def bwd_kernel(i):
        k = i % nbuf
        return L.mhaq_fq_act_bwd_partials(xs[k].data_ptr(), gs[k].data_ptr(), ys[k].data_ptr(), n, params.data_ptr(),
                                          0, None, 1234, i + 1, None, ws.data_ptr(), nb, ctypes.byref(nparts), st)

    def bwd_full(i):
        k = i % nbuf
        return L.mhaq_fq_act_bwd(xs[k].data_ptr(), gs[k].data_ptr(), ys[k].data_ptr(), n, params.data_ptr(), 0, None,
                                 1234, i + 1, None, grads.data_ptr(), ws.data_ptr(), nb, st)

    def timed(fn):
        """Average duration of back-to-back launches between two HIP events on the launch stream."""
        for i in range(10):      # warm-up: clocks ramp for the first few hundred microseconds of work
            assert fn(i) == 0
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(reps):
            fn(i)
        e.record()
        torch.cuda.synchronize()
        return a.elapsed_time(e) / reps  # ms

    # 5 interleaved rounds, median per kernel: single rounds vary by +-5 % with clock / power state
    rounds = [(timed(fwd), timed(bwd_kernel), timed(bwd_full)) for _ in range(5)]
    t_f, t_bk, t_b = (sorted(r[i] for r in rounds)[2] for i in range(3))
    ach = 12.0 * n / t_bk / 1e6
    traffic, traffic_src = profiled_traffic()
    roof = {"bound": "hbm", "kernel": DOMINANT_KERNEL + " (NoisyAct backward, STE, as mhaq_fq_act_bwd launches it)",
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
            "traffic": traffic, "traffic_source": traffic_src, "bytes_per_launch": 12 * n,
            "avg_launch_us": round(t_bk * 1e3, 2),
            "tensor": "resnet18 layer1 activation [250,64,56,56] fp32",
            "measured_copy_ceiling": HBM_COPY_GBPS, "frac_of_copy_ceiling": round(ach / HBM_COPY_GBPS, 4)}
    extra = {"fq_fwd_GBps": round(8.0 * n / t_f / 1e6, 1),
             "fq_bwd_with_finalize_GBps": round(12.0 * n / t_b / 1e6, 1),
             "fq_fused_fwd_bwd_GBps": round(20.0 * n / (t_f + t_b) / 1e6, 1),
             "fq_fused_frac_of_peak": round(20.0 * n / (t_f + t_b) / 1e6 / HBM_PEAK_GBPS, 4)}
    del xs, gs, ys
    torch.cuda.empty_cache()
    return roof, extra


# ------------------------------------------------------------------------------ the whole activation set
RESNET18_ACT_SHAPES = ([(64, 56, 56)] * 5 + [(128, 28, 28)] * 4 + [(256, 14, 14)] * 4 + [(512, 7, 7)] * 3)


def roofline_set(dev, batch=250, reps=10):
    """BASELINE's second metric on its own workload: the fused fake-quant forward + backward over ALL 16
    NoisyAct tensors of ResNet-18 W4A4 at per-GPU batch 250 (420.2 M elements, 20 B/elem
    algorithmic = 8.40 GB per pass), as a training step runs them: 16 forwards in layer order, 16 backwards in
    reverse order, every quantizer with its own parameters and its own tensors (6.7 GB resident, nothing is
    cache-warm), ONE joint finalize for the scalar gradients (act_hub.py).
      per_size   kernel rates per tensor size through the raw C ABI, buffers rotated
      set_capi   the sequence as raw C-ABI calls (device-side rate: ~5 us of host per launch)
      set_autograd  the same sequence through the product's autograd ops and NoisyAct modules, eager, from an idle
                    stream (the GPU starves on the 6 M-element tensors: ~24 / ~60 us of Python + autograd per
                    forward / backward op) -- and `_queued`: with the launch queue pre-filled behind a spin kernel,
                    which is the situation inside a GPU-bound training step
      set_graph  that autograd sequence captured once and replayed as a hipGraph
      weights / set_with_weights_*  the 16 per-channel weight tensors of the config on top: as the trainer runs them
                    (one model-wide forward launch + the backward in 3 groups of consecutive layers), all per-layer,
                    and the multi-tensor form with one launch per direction (single GPU)"""
    import ctypes
    import math

    import mhaq_amd as M
    from mhaq_amd import _lib, ops
    from mhaq_amd.act_hub import ActGradHub
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    shapes = [(batch, *s) for s in RESNET18_ACT_SHAPES]
    gen = torch.Generator(device=dev).manual_seed(3)
    xs = [(torch.randn(s, device=dev, generator=gen) * 2).contiguous(memory_format=torch.channels_last) for s in shapes]
    gs = [torch.randn(s, device=dev, generator=gen).contiguous(memory_format=torch.channels_last) for s in shapes]
    ys = [torch.empty_like(x) for x in xs]
    gxs = [torch.empty_like(x) for x in xs]
    acts = torch.nn.ModuleList([M.NoisyAct(signed=(i % 2 == 0)) for i in range(len(shapes))]).to(dev).train()
    with torch.no_grad():                                   # the post-calibration state (minmaxobserver.py:56-61), 4 bit
        for a, x in zip(acts, xs):
            mn, mx = ops.minmax(x).tolist()
            a.act_b.fill_(mn)
            a.log_act_s.fill_(math.log2((mx - mn) / 15))
            a.log_act_q.fill_(math.log2((mx - mn) / 15) + 4)
    params = [torch.empty(5, device=dev) for _ in shapes]
    grads = [torch.empty(3, device=dev) for _ in shapes]
    wss = [torch.empty(L.mhaq_fq_act_bwd_workspace_bytes(x.numel()), dtype=torch.uint8, device=dev) for x in xs]
    nparts = ctypes.c_int32(0)
    off = [0]

    def fwd(i):
        a = acts[i]
        assert L.mhaq_fq_act_fwd(xs[i].data_ptr(), ys[i].data_ptr(), xs[i].numel(), a.log_act_s.data_ptr(),
                                 a.log_act_q.data_ptr(), a.act_b.data_ptr(), params[i].data_ptr(), None, None, None,
                                 0, st) == 0

    def bwd(i, finalize):
        off[0] += 1
        if finalize:
            rc = L.mhaq_fq_act_bwd(xs[i].data_ptr(), gs[i].data_ptr(), gxs[i].data_ptr(), xs[i].numel(),
                                   params[i].data_ptr(), 0, None, 1234, off[0], None, grads[i].data_ptr(),
                                   wss[i].data_ptr(), wss[i].numel(), st)
        else:
            rc = L.mhaq_fq_act_bwd_partials(xs[i].data_ptr(), gs[i].data_ptr(), gxs[i].data_ptr(), xs[i].numel(),
                                            params[i].data_ptr(), 0, None, 1234, off[0], None, wss[i].data_ptr(),
                                            wss[i].numel(), ctypes.byref(nparts), st)
        assert rc == 0
        return nparts.value

    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return a.elapsed_time(e) / n

    def med(fn, n):
        return sorted(timed(fn, n) for _ in range(3))[1]

    for i in range(len(shapes)):
        fwd(i)
    # ---- per size (rotating over the 5 / 4 / 4 / 3 tensors of that size)
    groups = {}
    for i, s in enumerate(shapes):
        groups.setdefault(s, []).append(i)
    per_size = []
    for s, idx in groups.items():
        n = math.prod(s)
        # rotation far beyond the 256 MB Infinity Cache for EVERY kernel timed here: the 3-5 tensors of a size are cloned
        # until x + g + y + gx of the rotation hold >= 800 MB (a rotation that fits the cache turns a default-policy stream
        # -- torch's yardstick kernels below -- into cache hits while the non-temporal streams of this path still go to HBM)
        need = max(len(idx), int(800e6 // (16 * n)) + 1)
        rx = [xs[i] for i in idx] + [xs[idx[j % len(idx)]].clone() for j in range(need - len(idx))]
        rg = [gs[i] for i in idx] + [gs[idx[j % len(idx)]].clone() for j in range(need - len(idx))]
        ry = [ys[i] for i in idx] + [torch.empty_like(xs[idx[0]]) for _ in range(need - len(idx))]
        rgx = [gxs[i] for i in idx] + [torch.empty_like(xs[idx[0]]) for _ in range(need - len(idx))]
        a0, p0, g0, w0 = acts[idx[0]], params[idx[0]], grads[idx[0]], wss[idx[0]]
        k = [0]

        def f():
            j = k[0] % need; k[0] += 1
            assert L.mhaq_fq_act_fwd(rx[j].data_ptr(), ry[j].data_ptr(), n, a0.log_act_s.data_ptr(),
                                     a0.log_act_q.data_ptr(), a0.act_b.data_ptr(), p0.data_ptr(), None, None, None, 0, st) == 0

        def bk():
            j = k[0] % need; k[0] += 1; off[0] += 1
            assert L.mhaq_fq_act_bwd_partials(rx[j].data_ptr(), rg[j].data_ptr(), rgx[j].data_ptr(), n, p0.data_ptr(), 0,
                                              None, 1234, off[0], None, w0.data_ptr(), w0.numel(), ctypes.byref(nparts),
                                              st) == 0

        def bf():
            j = k[0] % need; k[0] += 1; off[0] += 1
            assert L.mhaq_fq_act_bwd(rx[j].data_ptr(), rg[j].data_ptr(), rgx[j].data_ptr(), n, p0.data_ptr(), 0, None,
                                     1234, off[0], None, g0.data_ptr(), w0.data_ptr(), w0.numel(), st) == 0
        f()
        tf, tk, tb = med(f, 3 * reps), med(bk, 3 * reps), med(bf, 3 * reps)

        # the bare streams of the same size on the same box, timed the same way over the same rotation: torch's own
        # elementwise kernels for one read + one write and two reads + one write -- a live yardstick next to every
        # figure (the hand-written bare streams with the kernels' own access pattern: tools/size_ceilings.hip,
        # profiles/r04_size_ceilings.txt)
        def c1():
            j = k[0] % need; k[0] += 1
            torch.mul(rx[j], 2.0, out=ry[j])

        def c2():
            j = k[0] % need; k[0] += 1
            torch.add(rx[j], rg[j], out=rgx[j])
        t1, t2 = med(c1, 3 * reps), med(c2, 3 * reps)
        per_size.append({"tensor": list(s), "elements": n, "count": len(idx), "rotation": need,
                         "fwd_us": round(tf * 1e3, 2), "bwd_kernel_us": round(tk * 1e3, 2),
                         "bwd_with_own_finalize_us": round(tb * 1e3, 2),
                         "fwd_GBps": round(8 * n / tf / 1e6, 1), "bwd_GBps": round(12 * n / tk / 1e6, 1),
                         "fused_GBps": round(20 * n / (tf + tk) / 1e6, 1),
                         "torch_mul_1r1w_us": round(t1 * 1e3, 2), "torch_add_2r1w_us": round(t2 * 1e3, 2),
                         "fwd_vs_torch_1r1w": round(t1 / tf, 3), "bwd_vs_torch_2r1w": round(t2 / tk, 3)})
        del rx, rg, ry, rgx
        torch.cuda.empty_cache()
    ntot = sum(math.prod(s) for s in shapes)

    # ---- the 16-tensor sequence, raw C ABI: forwards, backwards (partials), one joint finalize
    table = torch.empty(len(shapes), 2, dtype=torch.int64, device=dev)
    slab = torch.empty(len(shapes), 3, device=dev)
    nps = [bwd(i, False) for i in range(len(shapes))]
    table.copy_(torch.tensor([[w.data_ptr(), k] for w, k in zip(wss, nps)], dtype=torch.int64))

    def seq_capi():
        for i in range(len(shapes)):
            fwd(i)
        for i in reversed(range(len(shapes))):
            bwd(i, False)
        assert L.mhaq_fq_act_bwd_finalize_multi(table.data_ptr(), len(shapes), slab.data_ptr(), st) == 0

    def seq_capi_own_finalize():
        for i in range(len(shapes)):
            fwd(i)
        for i in reversed(range(len(shapes))):
            bwd(i, True)
    t_capi, t_own = med(seq_capi, reps), med(seq_capi_own_finalize, reps)

    # ---- the 16 per-channel weight tensors of the same config (10.99 M elements, 0.22 GB per pass: SURVEY.md 8d says
    # "8.40 GB + 0.22 GB => >= 1.54 ms at 5.6 TB/s"): per-layer launches (what the DDP trainer runs: 32 launches of
    # 6-14 us, launch-bound) and the multi-tensor form (2 launches, device pointer table; single-GPU option)
    wshapes = ([(64, 64, 3, 3)] * 4 + [(128, 64, 3, 3)] + [(128, 128, 3, 3)] * 3 + [(256, 128, 3, 3)] +
               [(256, 256, 3, 3)] * 3 + [(512, 256, 3, 3)] + [(512, 512, 3, 3)] * 3)
    ws_ = [torch.randn(sh, device=dev, generator=gen) * math.sqrt(2.0 / (sh[1] * 9)) for sh in wshapes]
    Gs_ = [torch.randn(sh, device=dev, generator=gen) for sh in wshapes]
    wq_ = [torch.empty_like(w) for w in ws_]
    gw_ = [torch.empty_like(w) for w in ws_]
    lss = []
    for w in ws_:                                          # 4-bit per-channel grid (HIP sweep: no torch reduction kernels)
        mn_, mx_ = ops.row_minmax(w)
        lss.append(torch.log2((mx_ - mn_) / 15).contiguous())
    auxs = [torch.empty(4, w.shape[0], device=dev) for w in ws_]
    glss = [torch.empty(w.shape[0], device=dev) for w in ws_]
    wmethod = 0                                            # STE (config 3)

    def weights_per_layer():
        for w, wq, ls, a in zip(ws_, wq_, lss, auxs):
            co, row = w.shape[0], w.numel() // w.shape[0]
            assert L.mhaq_fq_wlayer_fwd(w.data_ptr(), wq.data_ptr(), ls.data_ptr(), co, row, a[0].data_ptr(),
                                        a[1].data_ptr(), a[2].data_ptr(), a[3].data_ptr(), st) == 0
        for w, G, gw, a, gl in reversed(list(zip(ws_, Gs_, gw_, auxs, glss))):
            co, row = w.shape[0], w.numel() // w.shape[0]
            off[0] += 1
            assert L.mhaq_fq_wlayer_bwd(w.data_ptr(), G.data_ptr(), gw.data_ptr(), gl.data_ptr(), a[0].data_ptr(),
                                        a[1].data_ptr(), a[2].data_ptr(), None, co, row, wmethod, None, None, None,
                                        1234, off[0], None, st) == 0
    t_w_layers = med(weights_per_layer, reps)

    class Desc(ctypes.Structure):          # mhaq_wlayer_desc
        _fields_ = [("w", ctypes.c_void_p), ("log_s", ctypes.c_void_p), ("G", ctypes.c_void_p),
                    ("g_lwq", ctypes.c_void_p), ("co", ctypes.c_int64), ("row", ctypes.c_int64),
                    ("elem_offset", ctypes.c_int64), ("chan_offset", ctypes.c_int64)]
    arr = (Desc * len(ws_))()
    eo = co_ = 0
    for i, (w, ls, G) in enumerate(zip(ws_, lss, Gs_)):
        arr[i] = Desc(w.data_ptr(), ls.data_ptr(), G.data_ptr(), None, w.shape[0], w.numel() // w.shape[0], eo, co_)
        eo += w.numel()
        co_ += w.shape[0]
    wtable = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    wq_all, gw_all = torch.empty(eo, device=dev), torch.empty(eo, device=dev)
    aux_all, gls_all = torch.empty(4, co_, device=dev), torch.empty(co_, device=dev)
    max_row = max(w.numel() // w.shape[0] for w in ws_)

    def weights_multi():
        off[0] += 1
        assert L.mhaq_fq_wlayer_fwd_multi(wtable.data_ptr(), len(ws_), co_, max_row, wq_all.data_ptr(),
                                          aux_all.data_ptr(), st) == 0
        assert L.mhaq_fq_wlayer_bwd_multi(wtable.data_ptr(), len(ws_), co_, max_row, aux_all.data_ptr(),
                                          gw_all.data_ptr(), gls_all.data_ptr(), wmethod, None, 1234, off[0], None,
                                          st) == 0
    t_w_multi = med(weights_multi, reps)

    # what QATTrainer runs (and the data-parallel trainer too): one model-wide forward launch, the backward in
    # groups of consecutive layers cut from the end of the model (multi.py: 3 launches on ResNet-18)
    from mhaq_amd.multi import backward_groups
    chan0 = [sum(x.shape[0] for x in ws_[:i]) for i in range(len(ws_))]
    elem0 = [sum(x.numel() for x in ws_[:i]) for i in range(len(ws_))]
    grouped = backward_groups([w.numel() for w in ws_], [wmethod] * len(ws_), 4 << 20)
    in_group = {i for a, b in grouped for i in range(a, b)}
    gplans = []
    for a, b in grouped:
        garr = (Desc * (b - a))()
        for k, i in enumerate(range(a, b)):
            garr[k] = Desc(ws_[i].data_ptr(), None, Gs_[i].data_ptr(), None, ws_[i].shape[0],
                           ws_[i].numel() // ws_[i].shape[0], elem0[i] - elem0[a], chan0[i] - chan0[a])
        gco = sum(ws_[i].shape[0] for i in range(a, b))
        gel = sum(ws_[i].numel() for i in range(a, b))
        gplans.append((torch.frombuffer(bytearray(bytes(garr)), dtype=torch.uint8).to(dev), b - a, gco,
                       max(ws_[i].numel() // ws_[i].shape[0] for i in range(a, b)), chan0[a],
                       torch.empty(gel, device=dev), torch.empty(gco, device=dev)))

    def weights_trainer_form():
        assert L.mhaq_fq_wlayer_fwd_multi(wtable.data_ptr(), len(ws_), co_, max_row, wq_all.data_ptr(),
                                          aux_all.data_ptr(), st) == 0
        for tab, n, gco, grow, c0, gwb, glb in gplans:
            off[0] += 1
            assert L.mhaq_fq_wlayer_bwd_group(tab.data_ptr(), n, gco, grow, aux_all.data_ptr() + 4 * c0, co_,
                                              gwb.data_ptr(), glb.data_ptr(), wmethod, None, 1234, off[0], None,
                                              st) == 0
        for i in reversed(range(len(ws_))):
            if i in in_group:
                continue
            w, co, row = ws_[i], ws_[i].shape[0], ws_[i].numel() // ws_[i].shape[0]
            off[0] += 1
            assert L.mhaq_fq_wlayer_bwd(w.data_ptr(), Gs_[i].data_ptr(), gw_[i].data_ptr(), glss[i].data_ptr(),
                                        aux_all[0, chan0[i]:].data_ptr(), aux_all[1, chan0[i]:].data_ptr(),
                                        aux_all[2, chan0[i]:].data_ptr(), None, co, row, wmethod, None, None, None,
                                        1234, off[0], None, st) == 0
    t_w_trainer = med(weights_trainer_form, reps)
    nw = eo

    # ---- the same through the product: NoisyAct modules, autograd ops, the gradient hub
    hub = ActGradHub(acts)

    def seq_autograd():
        for p in acts.parameters():
            p.grad = None
        hub.begin()
        outs = [a(x.detach().requires_grad_(True)) for a, x in zip(acts, xs)]
        hub.end()
        torch.autograd.backward(outs, gs)
    t_auto = med(seq_autograd, reps)

    # ... and the eager path with the launch queue pre-filled, as inside a GPU-bound training step (there the host
    # runs ~4x ahead of the device): a ~4 ms spin kernel goes first, the host enqueues the whole pass behind it, and
    # the events bracket only the pass
    def queued_once():
        torch.cuda.synchronize()
        torch.cuda._sleep(int(4e-3 * 2.4e9))
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        seq_autograd()
        e.record()
        torch.cuda.synchronize()
        return a.elapsed_time(e)
    t_queued = sorted(queued_once() for _ in range(7))[3]
    # ---- ... and replayed as one hipGraph (the product's capture mode: fresh sign streams per replay through the
    # device-resident offset word): the device-side rate of the product path, with the ~40-70 us of Python and
    # autograd per op -- which starve the GPU on the 6 M-element tensors of this isolated sweep, but not inside
    # a training step, where the host runs 4x ahead of the device -- off the clock
    t_graph = None
    try:
        base = torch.zeros(1, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), ops.rng.device_offset(base):
            seq_autograd()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for p in acts.parameters():
            p.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side), ops.rng.device_offset(base):
            drawn = ops.rng.drawn()
            seq_autograd()
            base.add_(ops.rng.drawn() - drawn)
        t_graph = med(graph.replay, reps)
        del graph
    except Exception as e:  # noqa: BLE001 -- measurement leg only
        log(f"activation set: graph leg failed: {e!r}")
    out = {"workload": f"all 16 NoisyAct tensors of ResNet-18 W4A4, per-GPU batch {batch} (SURVEY.md 8d config 3)",
           "elements": ntot, "bytes_per_pass": 20 * ntot, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "per_size": per_size,
           "set_capi_ms": round(t_capi, 4), "set_capi_GBps": round(20 * ntot / t_capi / 1e6, 1),
           "set_capi_frac": round(20 * ntot / t_capi / 1e6 / HBM_PEAK_GBPS, 4),
           "set_capi_per_quantizer_finalize_ms": round(t_own, 4),
           "weights": {"tensors": len(ws_), "elements": nw, "estimator": "STE", "per_layer_launches_ms": round(t_w_layers, 4),
                       "forward_multi_backward_grouped_ms": round(t_w_trainer, 4),
                       "backward_groups": [list(g) for g in grouped],
                       "multi_tensor_launches_ms": round(t_w_multi, 4)},
           "set_with_weights_capi_ms": round(t_capi + t_w_trainer, 4),
           "set_with_weights_capi_GBps": round(20 * (ntot + nw) / (t_capi + t_w_trainer) / 1e6, 1),
           "set_with_weights_per_layer_capi_GBps": round(20 * (ntot + nw) / (t_capi + t_w_layers) / 1e6, 1),
           "set_with_weights_multi_capi_ms": round(t_capi + t_w_multi, 4),
           "set_with_weights_multi_capi_GBps": round(20 * (ntot + nw) / (t_capi + t_w_multi) / 1e6, 1),
           "set_autograd_ms": round(t_auto, 4), "set_autograd_GBps": round(20 * ntot / t_auto / 1e6, 1),
           "set_autograd_frac": round(20 * ntot / t_auto / 1e6 / HBM_PEAK_GBPS, 4),
           "set_autograd_queued_ms": round(t_queued, 4),
           "set_autograd_queued_GBps": round(20 * ntot / t_queued / 1e6, 1),
           "set_autograd_queued_frac": round(20 * ntot / t_queued / 1e6 / HBM_PEAK_GBPS, 4),
           "set_graph_ms": None if t_graph is None else round(t_graph, 4),
           "set_graph_GBps": None if t_graph is None else round(20 * ntot / t_graph / 1e6, 1),
           "set_graph_frac": None if t_graph is None else round(20 * ntot / t_graph / 1e6 / HBM_PEAK_GBPS, 4)}
    del xs, gs, ys, gxs
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------ CPU baseline leg
def cpu_baseline(args):
    from mhaq_amd import nets
    from mhaq_amd.enums import QNMethod, QScheme
    from mhaq_amd.qat import QATConfig, QATTrainer
    from oracle.loss import LOSS_CLASSES
    from oracle.ref_layers import ORACLE_LAYERS  # the checker, timed as the reported CPU baseline
    # the box's CPU share, not the host's core count (oversubscribing a cgroup-limited box is ~30x slower)
    from tools.fq_sets import host_cores
    cores, cores_how = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    cfg = QATConfig(qscheme=QScheme.PER_CHANNEL, qnmethod=QNMethod[args.qnmethod],
                    distillation=not args.no_distillation)
    B = args.cpu_batch or args.batch            # the GPU line's own batch: both images/s figures are the same workload
    x = torch.randn(B, 3, args.image, args.image)
    y = torch.randint(0, 1000, (B,))
    tr = QATTrainer(nets.resnet18(1000), cfg, "cpu", calib_batches=[x], layers=ORACLE_LAYERS, loss_classes=LOSS_CLASSES,
                    minmax_fn=lambda t: torch.stack(list(t.aminmax())), distributed=False)
    log("cpu baseline: 2 warm-up steps")
    tr.train_step(x, y)
    tr.train_step(x, y)
    log("cpu baseline: timed steps")
    t0 = time.perf_counter()
    steps = 0
    while steps < args.cpu_steps or (time.perf_counter() - t0 < args.cpu_seconds and steps < 500):
        tr.train_step(x, y)
        steps += 1
        if steps % 4 == 0:
            log(f"cpu baseline: {steps} steps, {time.perf_counter() - t0:.1f} s")
    dt = time.perf_counter() - t0
    out = {"value": round(B * steps / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(),
           "kind": "port", "batch": B, "same_batch_as_gpu_line": B == args.batch,
           "ms_per_step": round(dt / steps * 1e3, 1),
           "sample": f"{steps} QAT steps ({dt:.1f} s) of the same ResNet-18 {args.qnmethod} config at batch {B} "
                     f"({args.image}x{args.image}; the GPU line's batch is {args.batch}) with the eager CPU oracle "
                     f"layers, after 2 warm-up steps, on {torch.get_num_threads()} threads = the box's CPU share "
                     f"({cores_how})"}
    # SURVEY.md 8(d) "CPU baseline timing" / BASELINE configs[0]: the eager fake-quant chain alone over the ResNet-20
    # batch-128 tensor set (18 + 18 quantizers, seeds 0-4, 2 warm-ups + 5 timed passes) on all host cores
    log("cpu baseline: the fake-quant chain over the ResNet-20 batch-128 tensor set")
    try:
        from tools.fq_sets import cpu_fake_quant_set
        out["fake_quant_set"] = cpu_fake_quant_set()
    except Exception as e:  # noqa: BLE001 -- a secondary leg
        out["fake_quant_set"] = {"error": repr(e)[:300]}
    return out


def gpu_eager_baseline(args, dev, hip_ms_per_step):
    """The same QAT step on the same MI355X with the ORACLE's eager layers (the reference's op sequence, run by torch's own
    kernels) instead of the HIP layers: what the path replaces, on one box, in one run.  A reported baseline like
    cpu_baseline (checker code timed, never shipped); N = 1, rank 0 only."""
    from mhaq_amd import nets, ops
    from mhaq_amd.enums import QNMethod, QScheme
    from mhaq_amd.qat import QATConfig, QATTrainer
    from oracle.ref_layers import ORACLE_LAYERS
    torch.manual_seed(1234)
    ops.manual_seed(1234)
    cfg = QATConfig(qscheme=QScheme.PER_CHANNEL, qnmethod=QNMethod[args.qnmethod], distillation=not args.no_distillation)
    net = nets.resnet18(1000)
    gen = torch.Generator(device=dev).manual_seed(100)
    x = torch.randn(args.batch, 3, args.image, args.image, device=dev, generator=gen)
    y = torch.randint(0, 1000, (args.batch,), device=dev, generator=gen)
    if not args.nchw:
        net = net.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    tr = QATTrainer(net, cfg, dev, calib_batches=[x[:min(args.batch, 64)]], layers=ORACLE_LAYERS,
                    minmax_fn=lambda t: torch.stack(list(t.aminmax())), distributed=False, capture_graph=False)
    for _ in range(3):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    steps = 8
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    out = {"value": round(args.batch / ms * 1e3, 2), "unit": "images/s", "ms_per_step": round(ms, 3), "batch": args.batch,
           "kind": "port", "device": torch.cuda.get_device_name(dev),
           "peak_allocated_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
           "hip_layers_speedup": round(ms / hip_ms_per_step, 3),
           "sample": f"{steps} steps after 3 warm-up steps of the same ResNet-18 {args.qnmethod} step, batch {args.batch}, with "
                     f"the eager oracle layers (oracle/ref_layers.py: the reference's op chain) on the same GPU, same MIOpen "
                     f"convolutions; the HIP-layer line above: {hip_ms_per_step:.2f} ms/step"}
    del tr, net, x, y
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------ main
def claim_stdout():
    """stdout carries exactly one line, the result.  Libraries write there too (RCCL prints a five-line version
    banner to stdout when its first communicator is created), so file descriptor 1 is pointed at stderr for the
    whole run and the result line goes to a private duplicate of the original stdout."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def main():
    args = parse()
    result_out = claim_stdout()
    start_heartbeat()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch `python bench.py --gpus N` (starts its "
                         f"own ranks) or torchrun --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the fake-quant path has no CPU fallback")
    if os.environ.get("MHAQ_BENCH_SHARE_GPU") != "1" and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible): "
                         f"--gpus {args.gpus} needs one device per rank")
    # Rehearsal hooks for a one-GPU box (tests/test_gpu_bench_two_ranks.py): RCCL refuses two ranks on one
    # device, so MHAQ_BENCH_BACKEND=gloo + MHAQ_BENCH_SHARE_GPU=1 run every rank on cuda:0 over gloo.  The
    # driver's runs set neither: one rank per GPU over RCCL.
    backend = os.environ.get("MHAQ_BENCH_BACKEND", "nccl")
    if os.environ.get("MHAQ_BENCH_SHARE_GPU") == "1":
        if backend == "nccl":
            raise SystemExit("MHAQ_BENCH_SHARE_GPU=1 needs MHAQ_BENCH_BACKEND=gloo (RCCL wants one device per rank)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force = os.environ.get("MHAQ_FORCE_COLLECTIVES") == "1"   # rehearse the N>1 code path on one GPU
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # "nccl" == RCCL on ROCm
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    n_gpus = world
    # what the collectives actually run on: RCCL ("nccl" on ROCm) with this many ranks, or the gloo rehearsal
    rccl_ranks = dist.get_world_size() if (dist.is_initialized() and dist.get_backend() == "nccl") else (
        1 if not dist.is_initialized() else 0)

    from mhaq_amd import nets, ops
    from mhaq_amd.enums import QNMethod, QScheme
    from mhaq_amd.qat import QATConfig, QATTrainer

    roof = extra = None
    if rank == 0:
        log("kernel roofline leg")
        roof, extra = kernel_roofline(dev, args.kernel_reps)
        log(f"roofline: {roof['achieved']} GB/s; extra {extra}")

    rset = None
    # the 16-tensor set leg (6.7 GB of buffers, a graph capture, ~5 s) is an N = 1 measurement: at N > 1 the other
    # ranks would only wait for rank 0 in DDP's constructor, and a capture next to a live RCCL communicator buys nothing
    if rank == 0 and world == 1 and not args.no_roofline_set:
        log("roofline over the whole ResNet-18 W4A4 activation set")
        try:
            rset = roofline_set(dev, args.batch if args.batch >= 250 else 250)
            log(f"activation set: C ABI {rset['set_capi_GBps']} GB/s, autograd {rset['set_autograd_GBps']} GB/s")
        except Exception as e:  # noqa: BLE001 -- a secondary leg must not take the headline metric down with it
            rset = {"error": repr(e)[:500]}
            log(f"activation set leg failed: {e!r}")
            torch.cuda.empty_cache()

    # every other BASELINE configuration's quantizer set through the product path and the raw C ABI (the headline step
    # is configs[3] / [2]; roofline_set above is their ResNet-18 set): configs[0] on the GPU, configs[1], configs[4]
    cfgsets = None
    if rank == 0 and world == 1 and not args.no_configs:
        from tools.fq_sets import measure_config
        cfgsets = {}
        for key in ("resnet20_b128", "resnet20_b1000_pt", "rfdn_ref", "rfdn_stress"):
            log(f"quantizer set of BASELINE config: {key}")
            try:
                cfgsets[key] = measure_config(key, dev, reps=8)
                log(f"  {key}: C ABI {cfgsets[key]['set_capi_GBps']} GB/s, product {cfgsets[key]['set_product_GBps']} GB/s")
            except Exception as e:  # noqa: BLE001 -- a secondary leg must not take the headline metric down with it
                cfgsets[key] = {"error": repr(e)[:500]}
                log(f"  {key} failed: {e!r}")
                torch.cuda.empty_cache()

    if args.roofline_only:
        print(json.dumps({"roofline": roof, **(extra or {}), "roofline_set": rset, "configs": cfgsets}),
              file=result_out, flush=True)
        return

    # pl.Trainer(benchmark=None) turns cudnn.benchmark on unless deterministic (the reference's trainer.py:80-100
    # passes neither): MIOpen then searches for the fastest convolution algorithm per shape on first use
    torch.backends.cudnn.benchmark = not args.no_cudnn_benchmark

    torch.manual_seed(1234)          # identical initial weights on every rank
    ops.manual_seed(1234)
    cfg = QATConfig(qscheme=QScheme.PER_CHANNEL, qnmethod=QNMethod[args.qnmethod],
                    distillation=not args.no_distillation, student_high_priority=args.student_high_priority,
                    multi_weight_forward=not args.no_multi_weight_forward,
                    weight_backward_group_elems=args.weight_group_elems)
    net = nets.resnet18(1000)
    gen = torch.Generator(device=dev).manual_seed(100 + rank)   # different synthetic data per rank
    x = torch.randn(args.batch, 3, args.image, args.image, device=dev, generator=gen)
    y = torch.randint(0, 1000, (args.batch,), device=dev, generator=gen)
    calib = torch.randn(min(args.batch, 64), 3, args.image, args.image, device=dev,
                        generator=torch.Generator(device=dev).manual_seed(7))
    if not args.nchw:
        # Physical layout only: logical shapes, arithmetic and results are unchanged.  MIOpen's fp32 kernels on
        # gfx950 are NHWC-native (the NCHW run spends 6 % of its GPU time in layout transposes around them); the
        # per-tensor fake-quant kernels are layout-agnostic streams and a channels_last weight still has one
        # contiguous row per output channel, so the HIP path takes the tensors as they are.
        net = net.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
        calib = calib.contiguous(memory_format=torch.channels_last)
    if rank == 0:
        log("building + calibrating the quantized model")
    trainer = QATTrainer(net, cfg, dev, calib_batches=[calib],
                         multi_tensor_weights=args.multi_tensor_weights and world == 1,
                         capture_graph=({"off": False, "on": True, "auto": "auto"}[args.capture_graph]
                                        if world == 1 else False))

    if args.no_teacher_overlap:
        trainer.teacher_stream = None
    for i in range(args.warmup):
        trainer.train_step(x, y)
        torch.cuda.synchronize()
        if rank == 0:
            log(f"warm-up step {i + 1}/{args.warmup} done")
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.train_step(x, y)
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    loss_val = float(loss)
    if rank == 0:
        log(f"timed {args.steps} steps: {dt / args.steps * 1e3:.2f} ms/step")

    # Data-parallel invariants after the timed steps, reported on the line (a real N-GPU run and the one-GPU rehearsals
    # alike): every rank holds the same parameters (replicated weights + all-reduced gradients), and every
    # rank draws its own random sign stream (the reference's ranks draw independent randint_like streams, gdnsq.py:54)
    dp_check = None
    if dist.is_initialized() and world > 1:
        with torch.no_grad():
            chk = torch.stack([p.detach().double().abs().sum() for p in trainer.net.parameters()]).sum().reshape(1)
        sums = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(sums, chk)
        seed_r = torch.tensor([ops.rng.next()[0] & 0x7FFFFFFFFFFFFFFF], device=dev, dtype=torch.int64)
        seeds = [torch.zeros_like(seed_r) for _ in range(world)]
        dist.all_gather(seeds, seed_r)
        dp_check = {"params_in_sync": bool(all(float(v) == float(sums[0]) for v in sums)),
                    "param_abs_sum": float(sums[0]),
                    "distinct_sign_streams": len({int(v) for v in seeds}), "ranks": world}

    # the share of a step spent in the path's only exchange, the packed [3, Co] AEWGS
    # statistics all-reduce of each per-channel weight layer (issued from inside backward, in stream order).
    exchange_ms = None
    bufs = []
    if dist.is_initialized() and world > 1 and args.qnmethod == "AEWGS":
        # A secondary figure: it must not take the headline metric down with it -- and a rank that fails here (say,
        # out of memory for a buffer) must not leave the others blocked in a collective until the process-group
        # timeout.  So: allocate first, let all ranks AGREE that everyone succeeded (one int all-reduce), and only then
        # enter the timed collectives; a failure after that point is a failure of the communicator itself.
        ok = 1
        try:
            wf = trainer.weight_forward
            if wf is not None:          # one packed [3, group_co] message per backward group, one per ungrouped layer
                bufs = [torch.zeros(3, g.co, device=dev) for g in wf.groups]
                bufs += [torch.zeros(3, wf.co[i], device=dev) for i in range(wf.nlayers) if wf.group_of[i] is None]
            else:
                bufs = [torch.zeros(3, m.weight.shape[0], device=dev) for m in trainer.net.modules()
                        if hasattr(m, "log_wght_s") and getattr(m, "log_wght_s").numel() > 1]
        except Exception as e:  # noqa: BLE001
            ok = 0
            log(f"AEWGS exchange measurement: rank {rank} could not set up: {e!r}")
        flag = torch.tensor([ok], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            for _ in range(3):
                for b_ in bufs:
                    ops._allreduce_avg_(b_)
            torch.cuda.synchronize()
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                for b_ in bufs:
                    ops._allreduce_avg_(b_)
            e1.record()
            torch.cuda.synchronize()
            exchange_ms = e0.elapsed_time(e1) / 10
        elif rank == 0:
            log("AEWGS exchange measurement skipped: a rank could not set it up")

    n_weight_groups = len(trainer.weight_forward.groups) if trainer.weight_forward is not None else 0
    gpu_eager = None
    if rank == 0 and n_gpus == 1 and not args.no_gpu_eager_baseline and not args.no_cpu_baseline:
        log("gpu eager baseline: the same step with the oracle's eager layers on this GPU")
        try:
            del trainer
            torch.cuda.empty_cache()
            gpu_eager = gpu_eager_baseline(args, dev, dt / args.steps * 1e3)
            log(f"gpu eager baseline: {gpu_eager['ms_per_step']} ms/step")
        except Exception as e:  # noqa: BLE001 -- a secondary leg must not take the headline metric down with it
            gpu_eager = {"error": repr(e)[:500]}
            torch.cuda.empty_cache()
    cpu = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    if rank == 0:
        out = {
            "metric": "qat_images_per_sec",
            "value": round(n_gpus * args.batch * args.steps / dt, 2),
            "unit": "images/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ResNet-18 ImageNet-1k QAT step, {args.qnmethod} weights per-channel + STE "
                                   f"activations, {'Sym-KL distillation from FP teacher' if not args.no_distillation else 'CE'}"
                                   f", RAdam, synthetic {args.image}x{args.image}, {'NCHW' if args.nchw else 'channels_last'} memory format",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * n_gpus,
                       "parallelism": f"dp{n_gpus}", "final_loss": round(loss_val, 5),
                       "sync_batchnorm": bool(cfg.sync_batchnorm and n_gpus > 1),
                       "weight_backward_groups": n_weight_groups},
            "roofline": roof, "cpu_baseline": cpu, "gpu_eager_baseline": gpu_eager,
            "roofline_set": rset,
            "configs": cfgsets,
            "rccl_ranks": rccl_ranks,
            "collective_backend": dist.get_backend() if dist.is_initialized() else None,
        }
        out.update(extra or {})
        if dp_check is not None:
            out["data_parallel_check"] = dp_check
        if exchange_ms is not None:
            out["aewgs_allreduce_ms_per_step"] = round(exchange_ms, 4)
            out["aewgs_allreduces_per_step"] = len(bufs)
            out["aewgs_allreduce_share"] = round(exchange_ms / (dt / args.steps * 1e3), 5)
        print(json.dumps(out), file=result_out, flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
