"""The quantizer sets of the BASELINE configs as measurable units: every NoisyAct tensor and every
weight tensor of a config, fused fake-quant forward + backward, 20 B/elem algorithmic --

  * through the raw C ABI (include/mhaq_fq.h via ctypes: the device-side rate, ~5 us of host per launch),
  * through the PRODUCT path (NoisyAct / NoisyConv2d modules, the compiled autograd nodes, the activation hub, the
    model-wide weight forward and grouped weight backward), eager from an idle stream,
  * the product path replayed as a hipGraph (activations),
  * and, on the host, the eager CPU oracle on the same tensors (`cpu_fake_quant_set`; bench.py's cpu_baseline leg --
    the only function here that imports oracle/).

Used by bench.py (the `configs` block of the JSON line) and tools/fq_sweep.py (one JSON line per config).
"""
from __future__ import annotations

import ctypes
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

CONFIGS = {
    # key: (BASELINE.json configs[] index, description, model, batch, weight scheme, weight estimator)
    "resnet20_b128": (0, "ResNet-20 CIFAR-10 STE W4A4, batch 128 (the CPU-reference configuration, on the GPU)",
                      "resnet20", 128, "per_channel", "STE"),
    "resnet20_b1000_pt": (1, "ResNet-20 CIFAR-100 STE W4A4, batch 1000, per-tensor weight quantizers",
                          "resnet20", 1000, "per_tensor", "STE"),
    "resnet18_b250": (2, "ResNet-18 ImageNet-1k STE W4A4, batch 250, per-channel weights", "resnet18", 250,
                      "per_channel", "STE"),
    "rfdn_ref": (4, "RFDN LSQ W2A2, batch 24, reference training shape [24,50,24,24]", "rfdn", 24, "per_channel", "LSQ"),
    "rfdn_stress": (4, "RFDN LSQ W2A2, batch 24, stress shape [24,50,180,320]", "rfdn_stress", 24, "per_channel",
                    "LSQ"),
}
METHOD_ID = {"STE": 0, "EWGS": 1, "AEWGS": 2, "LSQ": 3}


def act_shapes(model, B):
    if model == "resnet20":
        return [(B, 16, 32, 32)] * 7 + [(B, 32, 16, 16)] * 6 + [(B, 64, 8, 8)] * 5
    if model == "resnet18":
        return [(B, 64, 56, 56)] * 5 + [(B, 128, 28, 28)] * 4 + [(B, 256, 14, 14)] * 4 + [(B, 512, 7, 7)] * 3
    if model == "rfdn":       # reference training shape h = w = 24 (sr/datamodule.py:63: HR crop 96 => LR 24)
        per = [(B, 50, 24, 24)] * 4 + [(B, 12, 24, 24)] + [(B, 12, 2, 2)] * 3
        return per * 4 + [(B, 50, 24, 24)]
    if model == "rfdn_stress":
        per = [(B, 50, 180, 320)] * 4 + [(B, 12, 180, 320)] + [(B, 12, 15, 26)] * 3
        return per * 4 + [(B, 50, 180, 320)]
    raise ValueError(model)


def weight_shapes(model):
    if model == "resnet20":
        return [(16, 16, 3, 3)] * 6 + [(32, 16, 3, 3)] + [(32, 32, 3, 3)] * 5 + [(64, 32, 3, 3)] + [(64, 64, 3, 3)] * 5
    if model == "resnet18":
        return [(64, 64, 3, 3)] * 4 + [(128, 64, 3, 3)] + [(128, 128, 3, 3)] * 3 + [(256, 128, 3, 3)] + \
               [(256, 256, 3, 3)] * 3 + [(512, 256, 3, 3)] + [(512, 512, 3, 3)] * 3
    if model == "vggfc":       # a per-channel model with ONE long-row layer: VGG-16's convolutions + a [64, 25088] Linear
        return [(64, 3, 3, 3), (64, 64, 3, 3), (128, 64, 3, 3), (128, 128, 3, 3), (256, 128, 3, 3)] + \
               [(256, 256, 3, 3)] * 2 + [(512, 256, 3, 3)] + [(512, 512, 3, 3)] * 5 + [(64, 25088, 1, 1)]
    per = [(50, 50, 3, 3)] * 3 + [(25, 50, 3, 3)] + [(12, 12, 3, 3)] * 4
    return per * 4 + [(50, 50, 3, 3)]


def _timeit(fn, reps, rounds=3):
    """median of `rounds` event-timed rounds of `reps` calls (3 warm-up calls first: clocks ramp over the first ms)"""
    for _ in range(3):
        fn()
    out = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / reps)
    return sorted(out)[len(out) // 2]


def _graph_timeit(fn, reps):
    """The same sequence captured ONCE into a hipGraph and timed per replay: the device-side rate of a launch sequence
    whose eager form is bound by the host's ~4-6 us per call (python + ctypes here, python + dispatcher for torch's own
    kernels) -- both sides of a comparison then pay the same per-node replay cost.  None if the capture fails."""
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            fn()
        t = _timeit(g.replay, reps)
        del g
        return t
    except Exception as e:  # noqa: BLE001 -- a measurement leg
        print(f"[fq_sets] graph leg failed: {e!r}", flush=True)
        return None


class _Desc(ctypes.Structure):          # mhaq_wlayer_desc
    _fields_ = [("w", ctypes.c_void_p), ("log_s", ctypes.c_void_p), ("G", ctypes.c_void_p), ("g_lwq", ctypes.c_void_p),
                ("co", ctypes.c_int64), ("row", ctypes.c_int64), ("elem_offset", ctypes.c_int64),
                ("chan_offset", ctypes.c_int64)]


def measure_config(key, dev, reps=10, graph=True):
    """One BASELINE configuration's quantizer set on the GPU.  Every quantizer has its own parameters (post-calibration
    state, minmaxobserver.py:56-61,82) and its own tensors."""
    import mhaq_amd as M
    from mhaq_amd import _lib, ops
    from mhaq_amd.act_hub import ActGradHub
    from mhaq_amd.multi import MultiTensorWeightQuant, backward_groups
    idx, desc, model, B, wscheme, wmethod = CONFIGS[key]
    L = _lib.lib()
    st = ops._stream            # the CURRENT stream's handle at call time (the replayed legs capture on a side stream)
    shapes = act_shapes(model, B)
    gen = torch.Generator(device=dev).manual_seed(idx)
    acts = torch.nn.ModuleList([M.NoisyAct() for _ in shapes]).to(dev).train()
    xs, gs = [], []
    for a, shp in zip(acts, shapes):
        x = torch.randn(shp, device=dev, generator=gen) * 2
        g = torch.randn(shp, device=dev, generator=gen)
        mn, mx = ops.minmax(x).tolist()
        with torch.no_grad():
            a.log_act_s.fill_(math.log2((mx - mn) / 1023))
            a.log_act_q.fill_(math.log2((mx - mn) / 1023) + 10)
            a.act_b.fill_(mn)
        xs.append(x)
        gs.append(g)
    n_act = sum(x.numel() for x in xs)

    # ---------------------------------------------------------------- activations, raw C ABI
    ys = [torch.empty_like(x) for x in xs]
    gxs = [torch.empty_like(x) for x in xs]
    params = [torch.empty(5, device=dev) for _ in xs]
    wss = [torch.empty(L.mhaq_fq_act_bwd_workspace_bytes(x.numel()), dtype=torch.uint8, device=dev) for x in xs]
    nparts = ctypes.c_int32(0)
    off = [0]

    def a_fwd(i):
        a = acts[i]
        assert L.mhaq_fq_act_fwd(xs[i].data_ptr(), ys[i].data_ptr(), xs[i].numel(), a.log_act_s.data_ptr(),
                                 a.log_act_q.data_ptr(), a.act_b.data_ptr(), params[i].data_ptr(), None, None, None, 0,
                                 st()) == 0

    def a_bwd(i):
        off[0] += 1
        assert L.mhaq_fq_act_bwd_partials(xs[i].data_ptr(), gs[i].data_ptr(), gxs[i].data_ptr(), xs[i].numel(),
                                          params[i].data_ptr(), 0, None, 1234, off[0], None, wss[i].data_ptr(),
                                          wss[i].numel(), ctypes.byref(nparts), st()) == 0
        return nparts.value
    for i in range(len(xs)):
        a_fwd(i)
    nps = [a_bwd(i) for i in range(len(xs))]
    table = torch.tensor([[w.data_ptr(), k] for w, k in zip(wss, nps)], dtype=torch.int64).to(dev)
    slab = torch.empty(len(xs), 3, device=dev)

    def acts_capi():
        for i in range(len(xs)):
            a_fwd(i)
        for i in reversed(range(len(xs))):
            a_bwd(i)
        assert L.mhaq_fq_act_bwd_finalize_multi(table.data_ptr(), len(xs), slab.data_ptr(), st()) == 0
    t_a_capi = _timeit(acts_capi, reps)

    def acts_bare():          # the same sequence as bare streams: torch's 1R1W / 2R1W elementwise kernels on the same tensors
        for i in range(len(xs)):
            torch.mul(xs[i], 2.0, out=ys[i])
        for i in reversed(range(len(xs))):
            torch.add(xs[i], gs[i], out=gxs[i])
    t_a_bare = _timeit(acts_bare, reps)
    t_a_capi_g = _graph_timeit(acts_capi, reps) if graph else None
    t_a_bare_g = _graph_timeit(acts_bare, reps) if graph else None

    # ---------------------------------------------------------------- activations, product path
    hub = ActGradHub(acts)

    def acts_product():       # all forwards, then ONE backward over every quantizer, like a training step
        for p in acts.parameters():
            p.grad = None
        hub.begin()
        outs = [a(x.detach().requires_grad_(True)) for a, x in zip(acts, xs)]
        hub.end()
        torch.autograd.backward(outs, gs)
    t_a_prod = _timeit(acts_product, reps)
    t_a_graph = None
    if graph:
        try:
            base = torch.zeros(1, dtype=torch.int64, device=dev)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), ops.rng.device_offset(base):
                acts_product()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_, stream=side), ops.rng.device_offset(base):
                drawn = ops.rng.drawn()
                acts_product()
                base.add_(ops.rng.drawn() - drawn)
            t_a_graph = _timeit(g_.replay, reps)
            del g_
            hub.release_captured()
        except Exception as e:  # noqa: BLE001 -- a measurement leg
            t_a_graph = None
            print(f"[fq_sets] {key}: graph leg failed: {e!r}", flush=True)

    # ---------------------------------------------------------------- weights
    wsh = weight_shapes("rfdn" if model.startswith("rfdn") else model)
    per_tensor = wscheme == "per_tensor"
    convs = torch.nn.ModuleList([
        M.NoisyConv2d(shp[1], shp[0], 3, bias=False, qscheme=M.QScheme.PER_TENSOR if per_tensor else M.QScheme.PER_CHANNEL,
                      qnmethod=M.QNMethod[wmethod]) for shp in wsh]).to(dev)
    Gs = []
    with torch.no_grad():
        for c, shp in zip(convs, wsh):
            c.weight.copy_(torch.randn(shp, device=dev, generator=gen) * math.sqrt(2.0 / (shp[1] * 9)))
            mn, mx = ops.row_minmax(c.weight)
            if per_tensor:
                mn, mx = mn.min(), mx.max()
            c.log_wght_s.copy_(torch.clamp(torch.log2((mx - mn) / 1023), min=-12.0).reshape(c.log_wght_s.shape))
            Gs.append(torch.randn(shp, device=dev, generator=gen))
    n_w = sum(G.numel() for G in Gs)
    mid = METHOD_ID[wmethod]
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
    arr = (_Desc * len(wsh))()
    for i, cv in enumerate(convs):
        arr[i] = _Desc(cv.weight.data_ptr(), cv.log_wght_s.data_ptr(), None, None, co[i], row[i], eo[i], cho[i])
    ftable = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    wq_all = torch.empty(tot_e, device=dev)
    aux_all = torch.empty(4, tot_c, device=dev)
    groups = backward_groups([a_ * b_ for a_, b_ in zip(co, row)], [mid] * len(wsh), 4 << 20)
    in_group = {i for a_, b_ in groups for i in range(a_, b_)}
    gplans = []
    for a_, b_ in groups:
        garr = (_Desc * (b_ - a_))()
        for k, i in enumerate(range(a_, b_)):
            garr[k] = _Desc(convs[i].weight.data_ptr(), None, Gs[i].data_ptr(), None, co[i], row[i], eo[i] - eo[a_],
                            cho[i] - cho[a_])
        gco = sum(co[i] for i in range(a_, b_))
        gel = sum(co[i] * row[i] for i in range(a_, b_))
        gplans.append((torch.frombuffer(bytearray(bytes(garr)), dtype=torch.uint8).to(dev), b_ - a_, gco,
                       max(row[i] for i in range(a_, b_)), cho[a_], torch.empty(gel, device=dev),
                       torch.empty(gco, device=dev)))
    gw_single = [torch.empty_like(G) for G in Gs]
    gl_single = [torch.empty(c, device=dev) for c in co]

    def weights_capi():      # the trainer's form: one model-wide forward launch, the backward in groups of layers
        assert L.mhaq_fq_wlayer_fwd_multi(ftable.data_ptr(), len(wsh), tot_c, max_row, wq_all.data_ptr(),
                                          aux_all.data_ptr(), st()) == 0
        for tab, n, gco, grow, c0, gwb, glb in gplans:
            off[0] += 1
            assert L.mhaq_fq_wlayer_bwd_group(tab.data_ptr(), n, gco, grow, aux_all.data_ptr() + 4 * c0, tot_c,
                                              gwb.data_ptr(), glb.data_ptr(), mid, None, 1234, off[0], None, st()) == 0
        for i in reversed(range(len(wsh))):
            if i in in_group:
                continue
            off[0] += 1
            assert L.mhaq_fq_wlayer_bwd(convs[i].weight.data_ptr(), Gs[i].data_ptr(), gw_single[i].data_ptr(),
                                        gl_single[i].data_ptr(), aux_all[0, cho[i]:].data_ptr(),
                                        aux_all[1, cho[i]:].data_ptr(), aux_all[2, cho[i]:].data_ptr(), None, co[i],
                                        row[i], mid, None, None, None, 1234, off[0], None, st()) == 0
    t_w_capi = _timeit(weights_capi, reps)
    t_w_capi_g = _graph_timeit(weights_capi, reps) if graph else None

    plan = MultiTensorWeightQuant(convs, joint_backward=False, backward_group_elems=4 << 20)
    ones = [torch.ones(c, device=dev) for c in co]

    def weights_product():
        plan.run()
        outs, grads = [], []
        for cv, G, o in zip(convs, Gs, ones):
            wq, _, _ = cv._quantized_weight()
            outs += [wq, cv.regulariser_input()]
            grads += [G, o]
        torch.autograd.backward(outs, grads)
        for cv in convs:
            cv.weight.grad = None
    t_w_prod = _timeit(weights_product, reps)
    # ... and replayed as a hipGraph: how the trainer runs them on the host-bound configurations (QATTrainer's captured
    # step) -- the eager figure above is the host's time for 18-33 layer modules, not the device's.  On its OWN module
    # instances that only ever run on the capture stream, like the trainer's: parameters whose AccumulateGrad nodes were
    # created on the default stream by the eager leg above must not meet a capture on another stream.
    t_w_graph = None
    if graph:
        try:
            import copy
            convs_g = copy.deepcopy(convs)
            plan_g = MultiTensorWeightQuant(convs_g, joint_backward=False, backward_group_elems=4 << 20)

            def weights_replayed():
                plan_g.run()
                outs, grads = [], []
                for cv, G, o in zip(convs_g, Gs, ones):
                    wq, _, _ = cv._quantized_weight()
                    outs += [wq, cv.regulariser_input()]
                    grads += [G, o]
                torch.autograd.backward(outs, grads)
                for cv in convs_g:
                    cv.weight.grad = None
                    cv.log_wght_s.grad = None
            base_w = torch.zeros(1, dtype=torch.int64, device=dev)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), ops.rng.device_offset(base_w):
                for _ in range(3):
                    weights_replayed()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            gw_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gw_, stream=side), ops.rng.device_offset(base_w):
                drawn = ops.rng.drawn()
                weights_replayed()
                base_w.add_(ops.rng.drawn() - drawn)
            t_w_graph = _timeit(gw_.replay, reps)
            del gw_
            plan_g.release_captured()
        except Exception as e:  # noqa: BLE001 -- a measurement leg
            t_w_graph = None
            print(f"[fq_sets] {key}: weight graph leg failed: {e!r}", flush=True)
    t_set_graph = None if (t_a_graph is None or t_w_graph is None) else t_a_graph + t_w_graph

    def gbps(n, ms):
        return None if ms is None else round(20.0 * n / ms / 1e6, 1)
    out = {"baseline_config": idx, "workload": desc, "act_tensors": len(shapes), "act_elements": n_act,
           "weight_tensors": len(wsh), "weight_elements": n_w, "weight_scheme": wscheme, "weight_estimator": wmethod,
           "weight_backward_groups": len(groups), "bytes_per_pass": 20 * (n_act + n_w),
           "act_capi_ms": round(t_a_capi, 4), "act_capi_GBps": gbps(n_act, t_a_capi),
           # live yardstick: the same 2 x N launches as torch's own bare 1R1W / 2R1W streams (host-bound, hence no
           # yardstick, where the set is small: ~6 us of host per torch call)
           "act_torch_streams_ms": round(t_a_bare, 4), "act_torch_streams_GBps": gbps(n_act, t_a_bare),
           "act_capi_vs_torch_streams": round(t_a_bare / t_a_capi, 3),
           # the two sequences above replayed as hipGraphs: the device-side comparison (eager, a set of 0.3-2 M-element
           # tensors measures ctypes against torch's dispatcher, not the kernels)
           "act_capi_graph_ms": None if t_a_capi_g is None else round(t_a_capi_g, 4),
           "act_capi_graph_GBps": gbps(n_act, t_a_capi_g),
           "act_torch_streams_graph_ms": None if t_a_bare_g is None else round(t_a_bare_g, 4),
           "act_capi_vs_torch_streams_graph": None if (t_a_capi_g is None or t_a_bare_g is None) else round(
               t_a_bare_g / t_a_capi_g, 3),
           "weight_capi_graph_ms": None if t_w_capi_g is None else round(t_w_capi_g, 4),
           "set_capi_graph_ms": None if (t_a_capi_g is None or t_w_capi_g is None) else round(t_a_capi_g + t_w_capi_g, 4),
           "set_capi_graph_frac_of_peak": None if (t_a_capi_g is None or t_w_capi_g is None) else round(
               20.0 * (n_act + n_w) / (t_a_capi_g + t_w_capi_g) / 1e6 / 8000.0, 4),
           "act_product_ms": round(t_a_prod, 4), "act_product_GBps": gbps(n_act, t_a_prod),
           "act_product_graph_ms": None if t_a_graph is None else round(t_a_graph, 4),
           "act_product_graph_GBps": gbps(n_act, t_a_graph),
           "weight_capi_ms": round(t_w_capi, 4), "weight_product_ms": round(t_w_prod, 4),
           "weight_product_graph_ms": None if t_w_graph is None else round(t_w_graph, 4),
           "set_product_graph_ms": None if t_set_graph is None else round(t_set_graph, 4),
           "set_product_graph_GBps": gbps(n_act + n_w, t_set_graph),
           "set_product_graph_frac_of_peak": None if t_set_graph is None else round(
               20.0 * (n_act + n_w) / t_set_graph / 1e6 / 8000.0, 4),
           "set_capi_ms": round(t_a_capi + t_w_capi, 4), "set_capi_GBps": gbps(n_act + n_w, t_a_capi + t_w_capi),
           "set_product_ms": round(t_a_prod + t_w_prod, 4), "set_product_GBps": gbps(n_act + n_w, t_a_prod + t_w_prod),
           "set_capi_frac_of_peak": round(20.0 * (n_act + n_w) / (t_a_capi + t_w_capi) / 1e6 / 8000.0, 4),
           "set_product_frac_of_peak": round(20.0 * (n_act + n_w) / (t_a_prod + t_w_prod) / 1e6 / 8000.0, 4),
           "cache_note": ("tensors of 0.3-65 MB: individually Infinity-Cache-resident (256 MB), the set as a whole is not"
                          if n_act * 16 > 3e8 else "the whole set fits the 256 MB Infinity Cache: not an HBM rate")}
    del xs, gs, ys, gxs, wss
    torch.cuda.empty_cache()
    return out


def host_cores():
    """(threads to use, how that number was found): the CPU share of THIS box -- the scheduler affinity mask, cut down to
    the cgroup's CPU quota when there is one (a GPU box leases 16 CPUs per GPU of a much larger host; running 100+
    threads against a 16-CPU quota is ~30x slower than 16 threads), MHAQ_CPU_THREADS overrides.  Without any quota
    information the count is capped at 16 per GPU-box convention and says so."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    if os.environ.get("MHAQ_CPU_THREADS"):
        return max(1, min(aff, int(os.environ["MHAQ_CPU_THREADS"]))), f"MHAQ_CPU_THREADS (affinity mask: {aff})"
    quota = None
    try:                                        # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                    # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(aff, int(math.ceil(quota))))
        return n, f"cgroup CPU quota {quota:.1f} (affinity mask: {aff})"
    if aff > 16:
        return 16, f"affinity mask {aff}, no cgroup quota visible: capped at the 16-CPU share of a one-GPU box"
    return aff, f"affinity mask {aff}"


def cpu_fake_quant_set(seeds=(0, 1, 2, 3, 4), warmups=2, batch=128, threads=None):
    """BASELINE configs[0] / SURVEY.md 8(d) 'CPU baseline timing': the eager PyTorch restatement of the reference's
    fake-quant chain (oracle/fq_eager.py, held to the reference's golden vectors) over the ResNet-20 batch-128 tensor set
    -- 18 activation + 18 per-channel weight quantizers, forward + backward -- on ALL host cores of the box:
    2 warm-ups, then one timed pass per seed (>= 5), time.perf_counter.  The pattern of the reference's only benchmark
    (tests/quant_implementatoin_perf.py:1-42: time an eager chain over a fixed tensor).  Checker code: only bench.py's
    cpu_baseline leg and tests call this."""
    from oracle import fq_eager as O
    cores, how = host_cores()
    threads = cores if threads is None else threads
    torch.set_num_threads(threads)
    a_shapes, w_shapes = act_shapes("resnet20", batch), weight_shapes("resnet20")

    def make(seed):
        gen = torch.Generator().manual_seed(seed)
        acts = []
        for shp in a_shapes:
            x = torch.randn(shp, generator=gen) * 2
            g = torch.randn(shp, generator=gen)
            mn, mx = float(x.min()), float(x.max())
            ls = math.log2((mx - mn) / 1023)
            acts.append((x, g, torch.tensor([ls]), torch.tensor([ls + 10]), torch.tensor([mn])))
        ws = []
        for shp in w_shapes:
            w = torch.randn(shp, generator=gen) * math.sqrt(2.0 / (shp[1] * 9))
            G = torch.randn(shp, generator=gen)
            span = w.amax((1, 2, 3)) - w.amin((1, 2, 3))
            ls = torch.clamp(torch.log2(span / 1023), min=-12.0).reshape(-1, 1, 1, 1)
            ws.append((w, G, ls))
        return acts, ws

    def one_pass(acts, ws):
        t0 = time.perf_counter()
        for x, g, ls, lq, b in acts:
            p = [t.clone().requires_grad_(True) for t in (ls, lq, b)]
            xr = x.clone().requires_grad_(True)
            y, _ = O.act_fake_quant(xr, *p, method="STE")
            y.backward(g)
        t1 = time.perf_counter()
        for w, G, ls in ws:
            wr, lr = w.clone().requires_grad_(True), ls.clone().requires_grad_(True)
            O.weight_fake_quant(wr, lr, True, "STE")[0].backward(G)
        return t1 - t0, time.perf_counter() - t1

    data = make(seeds[0])
    for _ in range(warmups):
        one_pass(*data)
    ta, tw = [], []
    for s in seeds:
        if s != seeds[0]:
            data = make(s)
        a, w = one_pass(*data)
        ta.append(a)
        tw.append(w)
    n_act = sum(math.prod(s) for s in a_shapes)
    n_w = sum(math.prod(s) for s in w_shapes)
    med_a, med_w = sorted(ta)[len(ta) // 2], sorted(tw)[len(tw) // 2]
    return {"workload": "ResNet-20 CIFAR W4A4 quantizer set, batch 128 (BASELINE configs[0]): 18 NoisyAct + 18 per-channel "
                        "weight quantizers, eager fake-quant forward + backward (oracle/fq_eager.py on torch CPU)",
            "kind": "port", "cores": threads, "cores_source": how, "seeds": list(seeds), "warmups": warmups,
            "timed_passes": len(ta), "act_elements": n_act, "weight_elements": n_w,
            "act_ms": round(med_a * 1e3, 2), "weight_ms": round(med_w * 1e3, 2),
            "act_GBps": round(20.0 * n_act / med_a / 1e9, 3), "weight_GBps": round(20.0 * n_w / med_w / 1e9, 4),
            "set_ms": round((med_a + med_w) * 1e3, 2),
            "set_GBps": round(20.0 * (n_act + n_w) / (med_a + med_w) / 1e9, 3), "unit": "GB/s (20 B/elem algorithmic)",
            "reference_in_container": "the real reference, 8 threads, same set: acts 403 ms, weights 17.8 ms (BASELINE.md)"}


if __name__ == "__main__":      # python3 tools/fq_sets.py [config ...]: one JSON object per BASELINE configuration's quantizer set
    import json
    for key in (sys.argv[1:] or list(CONFIGS)):
        print(json.dumps({key: measure_config(key, torch.device("cuda:0"), reps=8)}), flush=True)
