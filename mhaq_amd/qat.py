"""Minimal data-parallel QAT loop around the fake-quant path: what the reference delegates to
PyTorch-Lightning, reduced to what BASELINE's images/sec metric needs (SURVEY.md section 2 rows
13/14/18 are out of scope as code; their step semantics are restated here):

  training step     GDNSQQuant.distillation_noisy_training_step / noisy_train_decorator
                    (/root/reference/src/quantization/gdnsq/gdnsq_quant.py:194-233, 315-351)
  calibration       min/max observers (gdnsq/calib/minmaxobserver.py:39-88, training/trainer.py:187-223)
  temperature / LR  TemperatureScale.on_train_batch_end (src/callbacks/temperature_adjust.py:36-54)
  parallelism       DDPStrategy(find_unused_parameters=True) + SyncBatchNorm (training/trainer.py:83-97)
                    -> torch DDP over RCCL, one process per GPU, bucketed all-reduce overlapped with
                    backward (the never-used log_b_s is frozen instead of searched for every step);
                    AEWGS statistics ride ONE packed all-reduce per layer (ops.py).
"""
from __future__ import annotations

import copy
import time
from dataclasses import dataclass, field

import torch
import torch.distributed as dist
from torch import nn

from . import ops
from ._lib import MhaqFqError
from .enums import QNMethod, QScheme
from .loss import FusedPotentialLoss, FusedPotentialLossNoPred, SymmetricalKL
from .wrap import get_model_values, quantize_model


@dataclass
class QATConfig:
    """Mirrors the YAML keys the step loop reads (config/gdnsq_config_resnet18_imagenet_aewgs_w1a1.yaml)."""
    qscheme: QScheme = QScheme.PER_CHANNEL
    qnmethod: QNMethod = QNMethod.AEWGS
    act_bit: int = 1
    weight_bit: int = 1
    calib_act_bit: int = 10
    calib_weight_bit: int = 10
    excluded_layers: tuple = ("conv1", "fc")
    quantize_bias: bool = False
    distillation: bool = True
    learning_rate: float = 3e-4
    warmup: int = 100
    scale_lr: float = 1.0
    scale_t: float = 2.0
    sync_batchnorm: bool = True
    overlap_teacher: bool = True     # frozen teacher forward on a second HIP stream (CUDA devices only)
    joint_act_finalize: bool = True  # one finalize launch per backward for all NoisyAct quantizers (act_hub.py)
    student_high_priority: bool = False   # run the step on a priority -1 HIP stream, the teacher stays at 0
    multi_weight_forward: bool = True     # all per-channel weight FORWARDS in one launch per step (multi.py)
    # ... and their BACKWARD in groups of consecutive layers of at least this many weights, cut from the end of the
    # model (multi.py): one launch -- and, for AEWGS under data parallelism, one packed statistics all-reduce -- per
    # group instead of per layer.  0 = every layer keeps its own backward launch.
    weight_backward_group_elems: int = 4 << 20
    criterion: nn.Module = field(default_factory=nn.CrossEntropyLoss)


# ----------------------------------------------------------------------------- calibration
@torch.no_grad()
def calibrate_weights(model: nn.Module, wbits: int, max_bits: int = 24, row_minmax_fn=None) -> None:
    """apply_quantile_weights_s (gdnsq/calib/minmaxobserver.py:69-88): raise log_wght_s so that the channel
    range fits `wbits` bits.  The per-channel min / max come from ONE read-only HIP sweep per layer
    (mhaq_fq_row_minmax; whole-tensor layers: mhaq_fq_minmax); the [Co]-sized log2 / max arithmetic runs on the host
    in fp32 (one-off, and bit-identical to the reference's CPU result).
    Like the reference, `wbits = max_bits` sticks once a frozen layer has been met (minmaxobserver.py:78-79).
    PER_TENSOR layers use the global range (the reference's function raises on them: its reshape of the [Co] range to
    log_wght_s' shape [1] fails)."""
    row_minmax_fn = row_minmax_fn or ops.row_minmax      # the CPU checker (bench cpu_baseline / tests) passes its own
    for m in model.modules():
        if hasattr(m, "log_wght_s") and hasattr(m, "weight"):
            mn, mx = row_minmax_fn(m.weight)
            mn, mx = mn.cpu(), mx.cpu()
            if m.log_wght_s.numel() == 1 and mn.numel() != 1:
                mn, mx = mn.min(), mx.max()
            if not m.log_wght_s.requires_grad:
                wbits = max_bits
            floor = torch.log2((mx - mn) / (2 ** wbits - 1)).reshape(m.log_wght_s.shape)
            m.log_wght_s.copy_(torch.max(m.log_wght_s.detach().cpu(), floor))


def _cpu_row_minmax(w):
    dims = tuple(range(1, w.dim()))
    return w.detach().amin(dims), w.detach().amax(dims)


@torch.no_grad()
def calibrate_activations(model: nn.Module, batches, abits: int, max_bits: int = 24, minmax_fn=None) -> None:
    """MinMaxObserver hooks on every NoisyAct over `batches` (an eval-mode pass, trainer.py:205-213), then
    apply_mean_stats_activations (minmaxobserver.py:39-66).  min/max of each quantizer input come from the fused
    HIP sweep (ops.minmax: one read of the tensor instead of torch.min + torch.max); the scalar arithmetic runs on
    the host in fp32 like the reference's `torch.tensor([...])` round trip.
    Like the reference, `abits = max_bits` sticks once a frozen quantizer has been met (minmaxobserver.py:52-53)."""
    minmax_fn = minmax_fn or ops.minmax   # the CPU checker (bench cpu_baseline / tests) passes its own
    acts = [m for m in model.modules() if hasattr(m, "log_act_s") and hasattr(m, "act_b")]
    seen = {id(a): [] for a in acts}
    hooks = [a.register_forward_pre_hook(lambda mod, inp: seen[id(mod)].append(minmax_fn(inp[0]))) for a in acts]
    was_training = model.training
    model.eval()
    for x in batches:
        model(x)
    for h in hooks:
        h.remove()
    model.train(was_training)
    for a in acts:
        mm = torch.stack(seen[id(a)]).cpu()
        mn, mx = mm[:, 0].min(), mm[:, 1].max()
        if not a.log_act_q.requires_grad and not a.log_act_s.requires_grad:
            abits = max_bits
        if mx - mn > 0:
            log_s = torch.log2((mx - mn) / (2 ** abits - 1))
            a.act_b.fill_(mn)
            a.log_act_s.fill_(log_s)
            a.log_act_q.fill_(log_s + abits)
        else:  # pruned layer: zero-width range
            a.log_act_q.zero_(), a.log_act_s.zero_(), a.act_b.fill_(mn)
            a.log_act_q.requires_grad_(False), a.log_act_s.requires_grad_(False), a.act_b.requires_grad_(False)


class TemperatureSchedule:
    """TemperatureScale callback: warm-up LR ramp, then t += lr*scale_t per batch."""

    def __init__(self, base_lr, warmup=100, scale_lr=1.0, scale_t=2.0, scale_anneal=0.9995):
        self.lr, self.warmup, self.scale_lr, self.scale_t, self.scale_anneal = base_lr, warmup, scale_lr, scale_t, scale_anneal
        self.total_batch, self.t, self.lr_t, self.converged = 0, 0.0, 1.0, False

    @staticmethod
    def _set_lr(optimizer, new_lr):
        for g in optimizer.param_groups:
            if torch.is_tensor(g["lr"]):
                g["lr"].fill_(new_lr)      # capturable optimizers read the rate from the device (hipGraph replay)
            else:
                g["lr"] = new_lr

    def start(self, optimizer):
        """on_train_start (temperature_adjust.py:28-33): change_lr(..., 0) -- the first optimizer step runs at
        rate 0 and the ramp lr * k / warmup starts from there."""
        self._set_lr(optimizer, 0.0)

    def step(self, loss_mod, optimizer):
        self.total_batch += 1
        past = self.total_batch > self.warmup
        if past:
            self.t += self.lr * self.scale_t
            self.lr_t *= self.scale_lr if not self.converged else self.scale_anneal
        loss_mod.t = self.t
        new_lr = self.lr * self.lr_t if past else self.lr * self.total_batch / self.warmup
        self._set_lr(optimizer, new_lr)
        return new_lr


class _QATModule(nn.Module):
    """forward(x) -> (logits, las, laq, lws, lwq): keeps the regulariser inputs inside the DDP-wrapped
    forward so every parameter they touch is seen by the reducer (noisy_step, gdnsq_quant.py:315-317)."""

    def __init__(self, net, qscheme, act_hub=None, weight_forward=None):
        super().__init__()
        self.model = net
        self.qscheme = qscheme
        self.act_hub = act_hub
        self.weight_forward = weight_forward      # MultiTensorWeightQuant(joint_backward=False) or None
        self._quantized = None                    # the wrapped layers in module order (named_modules() walks ~10 us
                                                  # per module per step otherwise)

    def forward(self, x):
        if self.weight_forward is not None and self.training:
            self.weight_forward.run()             # every per-channel weight quantized by one launch
        if self.act_hub is not None and self.training:
            self.act_hub.begin()
        if self._quantized is None:
            self._quantized = [m for m in self.model.modules() if hasattr(m, "log_wght_s") or hasattr(m, "log_act_s")]
        try:
            return (self.model(x), *get_model_values(self.model, self.qscheme, modules=self._quantized))
        finally:
            if self.act_hub is not None:
                self.act_hub.end()


class QATTrainer:
    def __init__(self, net: nn.Module, cfg: QATConfig, device, calib_batches=None, layers=None,
                 distributed=None, minmax_fn=None, optimizer_factory=None,
                 multi_tensor_weights=False, capture_graph=None, loss_classes=None):
        """loss_classes = (PotentialLoss, PotentialLossNoPred) module classes; None = the HIP-backed ones
        (mhaq_amd/loss.py).  A CPU trainer is checker territory (tests, bench.py's cpu_baseline): it must bring its
        layers AND its loss (oracle/ref_layers.py, oracle/loss.py) -- the product has no CPU arithmetic."""
        self.cfg, self.device = cfg, torch.device(device)
        self.distributed = ops._dist_active() if distributed is None else distributed
        net = net.to(self.device)
        self.teacher = copy.deepcopy(net).eval().requires_grad_(False) if cfg.distillation else None
        quantize_model(net, cfg.qscheme, cfg.qnmethod, cfg.excluded_layers, cfg.quantize_bias, cfg.act_bit,
                       layers=layers)
        net.to(self.device)
        if calib_batches is not None:
            calibrate_weights(net, cfg.calib_weight_bit,
                              row_minmax_fn=_cpu_row_minmax if self.device.type != "cuda" else None)
            calibrate_activations(net, calib_batches, cfg.calib_act_bit, minmax_fn=minmax_fn)
        if self.distributed and cfg.sync_batchnorm and self.device.type == "cuda":
            net = nn.SyncBatchNorm.convert_sync_batchnorm(net)
        self.net = net
        self.teacher_stream = None
        if cfg.distillation and cfg.overlap_teacher and self.device.type == "cuda":
            self.teacher_stream = torch.cuda.Stream(device=self.device)
        # measured option (docs/NOTEBOOK.md section 6, "Teacher stream"): the student's fwd -> bwd chain is the critical path of a
        # distillation step; on a high-priority stream its kernels win the arbitration against the teacher's
        self._hp_stream = None
        if cfg.student_high_priority and self.device.type == "cuda":
            self._hp_stream = torch.cuda.Stream(device=self.device, priority=-1)
        self.multi = None
        if multi_tensor_weights and self.distributed:
            # the joint backward uses rank-local AEWGS statistics (no [3, total_co] all-reduce) and delays every
            # weight gradient to the end of backward, which defeats DDP's overlap: single-GPU option only
            raise ValueError("multi_tensor_weights is a single-GPU option (the data-parallel trainer keeps the "
                             "per-layer ops: AEWGS statistics all-reduce + gradient overlap)")
        if multi_tensor_weights:      # one launch for all weight quantizers (single-GPU option, multi.py)
            from .multi import MultiTensorWeightQuant
            self.multi = MultiTensorWeightQuant(net)
        self.act_hub = None
        if cfg.joint_act_finalize and self.device.type == "cuda" and (layers is None):
            from .act_hub import ActGradHub
            hub = ActGradHub(net)
            self.act_hub = hub if len(hub) > 1 else None
        self.weight_forward = None
        if cfg.multi_weight_forward and self.multi is None and self.device.type == "cuda" and layers is None:
            from .multi import MultiTensorWeightQuant
            try:
                wf = MultiTensorWeightQuant(net, joint_backward=False,
                                            backward_group_elems=cfg.weight_backward_group_elems)
                self.weight_forward = wf if wf.nlayers > 1 else None
            except ValueError:       # no layer the model-wide launch could serve
                pass
        self.module = _QATModule(net, cfg.qscheme, self.act_hub, self.weight_forward)
        # Data parallelism comes in two forms.  (a) torch DDP: bucketed all-reduce overlapped with backward -- right for
        # GPU-bound steps, but its reducer hooks are host code, so such a step cannot be replayed as a hipGraph.
        # (b) `_flat_sync`: the step runs locally with NO collective between its first launch and its last gradient,
        # then ONE all-reduce of the flattened gradients follows -- which leaves the whole forward + backward
        # capturable.  (b) is what the host-bound configurations want under data parallelism (RFDN at its 24x24
        # training shape: 11.3 ms of host per eager step against 5.4 ms replayed; 1.7 MB of gradients, an
        # all-reduce of microseconds) and needs a step without collectives of its own: no SyncBatchNorm, no AEWGS
        # statistics exchange.  Under capture_graph="auto" the trainer settles in form (b) and moves to DDP if
        # the step turns out GPU-bound.
        self._flat_sync = False
        want_capture = capture_graph
        if want_capture is None:
            want_capture = "auto" if (layers is None and optimizer_factory is None and not multi_tensor_weights) else False
        if self.distributed and want_capture and self.device.type == "cuda" and self.multi is None:
            # buffers: torch DDP re-broadcasts them from rank 0 every forward (plain BatchNorm's running statistics);
            # the flat form has no such exchange, so a model with buffers keeps DDP and its semantics
            step_collectives = any(isinstance(m, nn.SyncBatchNorm) for m in net.modules()) or any(
                ops._method_value(m.Q.qnmethod) == QNMethod.AEWGS.value for m in net.modules() if hasattr(m, "Q")
            ) or next(iter(net.buffers()), None) is not None
            self._flat_sync = not step_collectives
        if self.distributed:
            # The reference needs find_unused_parameters=True only because NoisyConv2d registers log_b_s,
            # which never receives a gradient (gdnsq_conv2d.py:57-59, trainer.py:92-95).
            # Freezing exactly those parameters lets the reducer skip its per-step graph traversal.
            # (log_b_s is read only by the constructor, here and in the reference: Q_b.scale is overwritten with
            # s.ravel() every forward, gdnsq_conv2d.py:86-88 -- so it is unused with quant_bias too.)
            for p in [m.log_b_s for m in net.modules() if hasattr(m, "log_b_s")]:
                p.requires_grad_(False)
            if self._flat_sync:
                with torch.no_grad():            # what DDP's constructor does: every rank starts from rank 0's state
                    for t in list(net.parameters()) + list(net.buffers()):
                        dist.broadcast(t, 0)
            else:
                self._wrap_ddp()
        # the hinge arithmetic is one HIP launch per direction (loss.py); a CPU trainer brings the checker's modules
        if loss_classes is None:
            if self.device.type != "cuda":
                raise MhaqFqError("QATTrainer on a CPU device needs loss_classes (and layers): the fake-quant path and "
                                  "its PotentialLoss run only as HIP kernels (no CPU fallback by design)")
            loss_classes = (FusedPotentialLoss, FusedPotentialLossNoPred)
        if cfg.distillation:
            self.loss = loss_classes[0](SymmetricalKL(), p=1, a=cfg.act_bit, w=cfg.weight_bit)
        else:
            self.loss = loss_classes[1](cfg.criterion, p=1, a=cfg.act_bit, w=cfg.weight_bit)
        # hipGraph option (single GPU): after three eager steps the device work of a step up to the gradients --
        # teacher and student forward, loss, backward, ~500 launches -- is captured once and replayed; the optimizer
        # then steps eagerly on the static gradient tensors (its ordinary foreach form: torch's graph-capturable
        # RAdam costs +5 ms per ResNet-18 step, measured in tools/graph_probe.py, which more than ate the gain).  It
        # pays where the host is the limit (ResNet-20 at batch 128: 11 ms/step of Python).  Everything a replay must
        # see fresh lives on the device: the loss state {loss_sum, cnt, t} and the offset of the random sign
        # streams (a uint64 word the backward kernels add to their frozen host offset, advanced by one step's worth
        # of streams at the end of every replay: replay k draws exactly the streams eager step k would have drawn,
        # at 0 extra bytes per element).  capture_graph="auto" decides from the settling steps: capture only if
        # the host needs more than 80 % of the step's wall time to enqueue it.
        if capture_graph is None:            # default: automatic for the stock HIP-layer trainer, off otherwise
            stock = layers is None and optimizer_factory is None and not multi_tensor_weights
            capture_graph = "auto" if stock else False
        self.capture_graph = capture_graph if capture_graph == "auto" else bool(capture_graph)
        self._graph = self._static = self._static_loss = None
        self._rng_base = None
        self._rng_host0 = self._rng_stride = 0
        self._preflight, self._preflight_syncs = False, []
        self._flat_cache = None
        self._eager_steps = 0
        self._host_share = []
        self._static_grads, self._grads_detached = [], False
        if self.capture_graph:
            if (self.distributed and not self._flat_sync) or self.device.type != "cuda" or self.multi is not None:
                if self.capture_graph == "auto":
                    self.capture_graph = False
                else:
                    raise ValueError("capture_graph needs a step without host code or collectives inside it: a "
                                     "single GPU, or data parallelism without SyncBatchNorm and without the AEWGS "
                                     "statistics exchange (then the gradients take one flat all-reduce after the "
                                     "replay); not the multi-tensor joint backward")
        if self.capture_graph:
            self._rng_base = torch.zeros(1, dtype=torch.int64, device=self.device)
            # the eager settling steps and the capture share one side stream: autograd keeps the AccumulateGrad
            # nodes of earlier iterations alive (with the stream they first ran on), and a node that belongs to
            # the default stream cannot take part in a capture
            self._gstream = torch.cuda.Stream(device=self.device)
        # RAdam as in every shipped config (vision_cls_module.py:54-55); a factory may override it
        self.optimizer = (optimizer_factory or torch.optim.RAdam)(self.net.parameters(), cfg.learning_rate)
        self.schedule = TemperatureSchedule(cfg.learning_rate, cfg.warmup, cfg.scale_lr, cfg.scale_t)
        self.schedule.start(self.optimizer)

    def _wrap_ddp(self):
        ids = [self.device.index] if self.device.type == "cuda" else None
        self.module = nn.parallel.DistributedDataParallel(self.module, device_ids=ids, find_unused_parameters=False,
                                                          gradient_as_bucket_view=True)
        self._flat_sync = False

    def _agree(self, *flags):
        """ONE all-reduce (MAX) over 0/1 verdicts: the value every rank acts on.  Single process: the flags themselves."""
        if not self.distributed:
            return tuple(bool(f) for f in flags)
        v = torch.tensor([1.0 if f else 0.0 for f in flags], dtype=torch.float32, device=self.device)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        return tuple(x > 0.5 for x in v.tolist())

    def _step_is_gpu_bound(self) -> bool:
        """This rank's verdict from its settling steps: the host needed less than 80 % of a step's wall time."""
        return min(self._host_share[1:]) < 0.8

    @staticmethod
    def _flat_alias(g):
        """A 1-D alias of a dense gradient in its physical order (a channels_last weight gradient has no flat view in
        logical order; an elementwise all-reduce does not care, every rank holds the same layout)."""
        if not ops._is_dense(g):
            raise MhaqFqError("flat gradient all-reduce: a gradient is not dense")
        return g.as_strided((g.numel(),), (1,), g.storage_offset())

    def _sync_grads(self, static=False):
        """Form (b) of data parallelism: ONE all-reduce (AVG) over all gradients of the step, flattened.  `static`: the
        gradient tensors are the captured graph's own (the same every replay), so the aliases are built once."""
        cache = self._flat_cache if static else None
        if cache is None:
            grads = [p.grad for p in self.net.parameters() if p.grad is not None]
            views = [self._flat_alias(g) for g in grads]
            flat = torch.empty(sum(v.numel() for v in views), dtype=torch.float32, device=self.device)
            cache = (views, flat, list(flat.split([v.numel() for v in views])))
            if static:
                self._flat_cache = cache
        views, flat, parts = cache
        torch.cat(views, out=flat)
        ops._allreduce_avg_(flat)
        torch._foreach_copy_(views, parts)

    def train_step(self, x, y):
        if not self.capture_graph:
            if self._hp_stream is not None:
                cur = torch.cuda.current_stream()
                self._hp_stream.wait_stream(cur)
                with torch.cuda.stream(self._hp_stream):
                    loss = self._step(x, y)
                cur.wait_stream(self._hp_stream)
            else:
                loss = self._step(x, y)
            self.schedule.step(self.loss, self.optimizer)
            return loss.detach()
        if self._graph is None and self._eager_steps < 3:
            # MIOpen's algorithm search, the optimizer's lazy state and the allocator settle here
            self._eager_steps += 1
            auto = self.capture_graph == "auto"
            if auto:
                torch.cuda.synchronize(self.device)
                t0 = time.perf_counter()
            cur = torch.cuda.current_stream()
            self._gstream.wait_stream(cur)
            self._preflight = self._eager_steps == 3       # the last settling step doubles as the capturability check
            with torch.cuda.stream(self._gstream), ops.rng.device_offset(self._rng_base):
                loss = self._step(x, y).detach()
            cur.wait_stream(self._gstream)
            if auto:
                t1 = time.perf_counter()
                torch.cuda.synchronize(self.device)
                self._host_share.append((t1 - t0) / max(time.perf_counter() - t0, 1e-9))
            if self._eager_steps == 3:
                # The mode decisions.  Under data parallelism they are COLLECTIVE: each rank contributes its local
                # verdict, ONE all-reduce (MAX) at this fixed step index makes every rank act on the same value -- the
                # reference picks one strategy before any rank starts (training/trainer.py:92-97).  Ranks deciding on
                # their own wall clocks could land on either side of the threshold: one would build torch DDP (constructor
                # broadcasts, bucketed all-reduces) while its peer captures and issues one flat all-reduce -> a hang.
                local_syncs = self._preflight_syncs
                self._preflight_syncs = []
                syncs, gpu_bound = self._agree(bool(local_syncs), auto and self._step_is_gpu_bound())
                if syncs or gpu_bound:
                    if syncs:
                        # a host synchronisation inside forward / loss / backward (a user layer calling .item(), .cpu(),
                        # ...): a capture of this step would fail, and a failed capture is hard to recover from
                        import warnings
                        where = local_syncs[0][:160] if local_syncs else "reported by another rank"
                        warnings.warn("QATTrainer: the training step synchronises with the host (" + where + "): it "
                                      "cannot be captured in a hipGraph; continuing with the eager loop", RuntimeWarning)
                    # GPU-bound (or not capturable): the eager loop.  It stays on the settling stream: the AccumulateGrad
                    # nodes remember it, and a step on another stream would pay a cross-stream event pair per parameter
                    # (+2 ms per ResNet-18 step, measured)
                    self.capture_graph = False
                    if self._hp_stream is None:
                        self._hp_stream = self._gstream
                    if gpu_bound and not syncs and self._flat_sync:
                        self._wrap_ddp()     # a GPU-bound step wants DDP's overlap of the all-reduce with backward
        elif self._static is not None and (x.shape != self._static[0].shape or y.shape != self._static[1].shape):
            # a batch of another shape (the last one of an epoch): this step runs eagerly, the graph stays
            # Sign streams: the captured launches hold host offsets c+1 .. c+K and the device word says how many steps
            # have run (R*K after R replays).  This step stands in for replay R: it draws from the SAME host counter
            # the capture started from (so it lands on the offsets replay R would have used), and then moves the
            # device word on like a replay does -- no later replay meets its streams again.
            cur = torch.cuda.current_stream()
            self._gstream.wait_stream(cur)
            host_after = ops.rng.drawn()
            ops.rng.set_drawn(self._rng_host0)
            with torch.cuda.stream(self._gstream), ops.rng.device_offset(self._rng_base):
                loss = self._step(x, y).detach()
                self._rng_base.add_(self._rng_stride)
            ops.rng.set_drawn(max(host_after, ops.rng.drawn()))
            cur.wait_stream(self._gstream)
            self._grads_detached = True      # p.grad now points at this step's tensors, not at the graph's
        else:
            if self._graph is None and not self._capture(x, y):
                return self.train_step(x, y)         # capture failed: the trainer has switched to the eager loop
            if self._grads_detached:
                for p, g in self._static_grads:
                    p.grad = g
                self._grads_detached = False
            self._static[0].copy_(x)
            self._static[1].copy_(y)
            self._graph.replay()
            if self._flat_sync:
                self._sync_grads(static=True)        # the step's one collective, after the replay
            self.optimizer.step()                    # eager, on the graph's static gradient tensors
            loss = self._static_loss.detach().clone()
        self.schedule.step(self.loss, self.optimizer)
        return loss

    def _capture(self, x, y) -> bool:
        """Capture the device work of one step (teacher + student forward, loss, backward).  A step that cannot be
        captured (a host sync or another non-capturable call in a user model, a descriptor pool run dry, ...) must
        not end the training run: the graph is discarded, the reason is reported once and the trainer carries on
        eagerly on the settling stream, exactly as capture_graph=False would have."""
        self._static = (x.clone(), y.clone())
        self.optimizer.zero_grad(set_to_none=True)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        drawn = ops.rng.drawn()
        # Under data parallelism another thread of this process talks to the GPU while we capture: the process group's
        # watchdog polls its work events.  In the default ("global") error mode such a call from ANY thread invalidates
        # a capture in progress; "thread_local" confines the check to the capturing thread.
        mode = "thread_local" if self.distributed else "global"
        err = None
        try:
            with torch.cuda.graph(graph, stream=self._gstream, capture_error_mode=mode), \
                    ops.rng.device_offset(self._rng_base):
                self._static_loss = self._forward_backward(*self._static)
                # the captured launches hold host offsets drawn+1 .. drawn+K; every replay ends by moving the
                # device word K further, so replay k runs at the offsets eager step k would have used
                self._rng_stride = ops.rng.drawn() - drawn
                self._rng_base.add_(self._rng_stride)
        except Exception as e:      # noqa: BLE001 -- whatever made the capture fail, training goes on eagerly
            err = e
        # every rank reaches this point in the same step; a capture that failed on ONE rank takes all of them to the
        # eager loop (same collective either way -- the flat all-reduce -- but one mode per job, like the reference)
        (failed,) = self._agree(err is not None)
        if failed:
            import warnings
            why = f"{type(err).__name__}: {err}" if err is not None else "it failed on another rank"
            warnings.warn(f"QATTrainer: hipGraph capture of the training step failed ({why}); "
                          "continuing with the eager loop", RuntimeWarning)
            del graph
            torch.cuda.synchronize(self.device)
            ops.rng.set_drawn(drawn)
            self.capture_graph = False
            self._graph = self._static = self._static_loss = None
            if self._hp_stream is None:
                self._hp_stream = self._gstream       # stay on the stream the AccumulateGrad nodes remember
            self.release_captured()
            self.optimizer.zero_grad(set_to_none=True)
            return False
        self._graph = graph
        self._rng_host0 = drawn
        # the captured backward assigned its output tensors to p.grad (set_to_none before it): every
        # replay rewrites exactly these, so they must be the ones the optimizer reads
        self._static_grads = [(p, p.grad) for p in self.net.parameters()]
        self._grads_detached = False
        return True

    def release_captured(self) -> None:
        """Let go of the device tables / workspaces that were only kept because a captured graph had their addresses
        baked in (act_hub.py, multi.py).  Called when the graph is dropped."""
        if self.act_hub is not None:
            self.act_hub.release_captured()
        if self.weight_forward is not None:
            self.weight_forward.release_captured()

    def drop_graph(self) -> None:
        """Forget the captured step (e.g. before the model's shapes change for good); the next step settles and
        captures again if capture_graph is still on."""
        if self._graph is not None:
            torch.cuda.synchronize(self.device)
        self._graph = self._static = self._static_loss = None
        self._static_grads, self._grads_detached = [], False
        self._flat_cache = None
        self._eager_steps = 0
        self.release_captured()

    def _step(self, x, y):
        loss = self._forward_backward(x, y)
        if self._flat_sync:
            self._sync_grads()
        self.optimizer.step()
        return loss

    def _forward_backward(self, x, y):
        """teacher + student forward, loss, backward: the body one replay of the captured graph repeats."""
        if self._preflight:
            # capturability check: run this (eager) step with torch's sync debug mode on "warn" and note every host
            # synchronisation it reports -- the step itself runs to completion unchanged
            import warnings
            self._preflight = False
            prev = torch.cuda.get_sync_debug_mode()
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                torch.cuda.set_sync_debug_mode("warn")
                try:
                    loss = self._forward_backward(x, y)
                finally:
                    torch.cuda.set_sync_debug_mode(prev)
            # c10's text is "called a synchronizing HIP operation" ("... CUDA operation" upstream); the one-time notice
            # that the debug mode is a prototype feature is not a synchronisation
            is_sync = lambda w: "called a synchronizing" in str(w.message)      # noqa: E731
            self._preflight_syncs = [str(w.message) for w in caught if is_sync(w)]
            for w in caught:
                if not is_sync(w) and "debug mode is a prototype" not in str(w.message):
                    warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
            return loss
        if not self.module.training:     # Module.train() walks every submodule: 0.6 ms per ResNet-20 step if unconditional
            self.module.train()
        if not self.loss.training:
            self.loss.train()
        if self.multi is not None:
            self.multi.run()
        side = self.teacher_stream if self.cfg.distillation else None
        fp = None
        if side is not None:
            # the frozen FP teacher has no dependence on the student: its forward runs on a second HIP stream
            # so that its kernels fill the gaps of the student's forward (45.7 -> 43.7 ms/step on ResNet-18)
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side), torch.no_grad():
                fp = self.teacher(x)
        out = self.module(x)
        if self.cfg.distillation:
            if side is None:
                with torch.no_grad():
                    fp = self.teacher(x)
            else:
                main.wait_stream(side)
                if not torch.cuda.is_current_stream_capturing():   # a graph's private pool needs no such note
                    fp.record_stream(main)
            loss = self.loss(out, fp)
        else:
            loss = self.loss((self.cfg.criterion(out[0], y), *out[1:]))
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        return loss

    @torch.no_grad()
    def validate_step(self, x, y):
        """What the reference logs per validation batch (noisy_validation_step + noisy_val_decorator,
        gdnsq_quant.py:234-301, 385-420): the plain criterion on the quantized model's prediction in eval mode,
        the six bit-width statistics and the converged flag.  The eval forward runs the fused kernels with
        their integrity flags (gdnsq.py:211-217), checked here with ONE host sync for the whole model."""
        from . import stats
        from .gdnsq import check_model_integrity
        self.module.eval()
        try:
            out = self.net(x)
            check_model_integrity(self.net)
        finally:
            self.module.train()
        with stats.memoised():          # one HIP sweep per weight layer for the six statistics + is_converged
            return self._validation_record(out, y, stats)

    def _validation_record(self, out, y, stats):
        return {
            "val_loss": self.cfg.criterion(out, y),
            "top1": (out.argmax(1) == y).float().mean(),
            "mean_weights_bit_width": stats.get_weights_bit_width_mean(self.net),
            "actual_weights_bit_width": stats.get_true_weights_width(self.net, max=False),
            "actual_weights_max_bit_width": stats.get_true_weights_width(self.net),
            "mean_activations_bit_width": stats.get_activations_bit_width_mean(self.net),
            "actual_activations_bit_width": stats.get_true_activations_width(self.net, max=False),
            "actual_activations_max_bit_width": stats.get_true_activations_width(self.net),
            "converged": stats.is_converged(self.net, self.loss),
        }
