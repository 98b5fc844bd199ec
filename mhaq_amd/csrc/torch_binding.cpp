// Compiled autograd binding of the fused fake-quant layer ops: torch::autograd::Function nodes in C++ over the SAME
// C ABI (include/mhaq_fq.h -> libmhaq_fq.so, resolved with dlopen / dlsym from the path the Python side hands over).
//
// Why: the reference's ops are Python torch.autograd.Functions invoked once per layer
//     /root/reference/src/quantization/gdnsq/gdnsq.py:13-23        (QNoise / QN* Functions)
//     /root/reference/src/quantization/gdnsq/layers/gdnsq_act.py:39-55      (NoisyAct.forward)
//     /root/reference/src/quantization/gdnsq/layers/gdnsq_conv2d.py:71-100  (NoisyConv2d.forward)
// and a Python Function over ctypes costs ~24 us per forward and ~60 us per backward op of host time (Python frames,
// ctypes argument conversion, the GIL hand-over to the autograd thread) -- the whole run time of the small
// configurations (ResNet-20 at batch 128, RFDN at 24x24), whose kernels take 2-10 us.  Here the forward is one
// pybind11 call and the backward never takes the GIL: the autograd engine calls straight into apply(), which
// allocates the outputs through the torch caching allocator and launches on the current HIP stream.
//
// This file holds NO arithmetic: every number is produced by the kernels behind the C ABI; results are bit-identical
// to the ctypes path (same entry points, same arguments).  Host-only C++ (no device code); torch is plumbing
// (device memory, streams, the autograd graph).
#include <torch/extension.h>
#include <torch/version.h>
#include <torch/csrc/autograd/custom_function.h>

#include <c10/hip/HIPCachingAllocator.h>
#include <c10/hip/HIPGraphsC10Utils.h>
#include <c10/hip/HIPStream.h>
#include <c10/core/DeviceGuard.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mhaq_fq.h"

namespace py = pybind11;
using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

namespace {

// ------------------------------------------------------------------------------------------------ the C ABI
#define MHAQ_API(X)                                                                                                   \
  X(mhaq_fq_abi_version) X(mhaq_fq_error_string) X(mhaq_fq_act_fwd) X(mhaq_fq_act_bwd_workspace_bytes)                \
  X(mhaq_fq_act_bwd) X(mhaq_fq_act_bwd_partials) X(mhaq_fq_act_bwd_finalize_multi) X(mhaq_fq_wlayer_fwd)              \
  X(mhaq_fq_wlayer_bwd) X(mhaq_fq_pc_aewgs_stats) X(mhaq_fq_wlayer_fwd_multi) X(mhaq_fq_wlayer_bwd_group)             \
  X(mhaq_fq_wlayer_aewgs_stats_group) X(mhaq_fq_wlayer_pt_fwd) X(mhaq_fq_wlayer_pt_bwd)                               \
  X(mhaq_fq_potential_loss_fwd) X(mhaq_fq_potential_loss_bwd) X(mhaq_fq_wlayer_ptl_workspace_bytes)                   \
  X(mhaq_fq_wlayer_ptl_fwd) X(mhaq_fq_wlayer_ptl_bwd) X(mhaq_fq_pt_aewgs_colstats_workspace_bytes)                     \
  X(mhaq_fq_pt_aewgs_colstats)

struct Api {
#define X(n) decltype(&::n) n = nullptr;
  MHAQ_API(X)
#undef X
  void* handle = nullptr;
  std::string path;
};
Api A;

struct MhaqError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

void bind_library(const std::string& path) {
  if (A.handle && A.path == path) return;
  void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) throw MhaqError("cannot load " + path + ": " + dlerror() + " (there is no CPU fallback)");
  Api a;
#define X(n)                                                                        \
  a.n = reinterpret_cast<decltype(a.n)>(dlsym(h, #n));                              \
  if (!a.n) throw MhaqError(std::string(#n) + " is not exported by " + path);
  MHAQ_API(X)
#undef X
  if (a.mhaq_fq_abi_version() != MHAQ_FQ_ABI_VERSION) throw MhaqError("libmhaq_fq.so ABI version mismatch");
  a.handle = h;
  a.path = path;
  A = a;
}

inline void need_lib() {
  if (!A.handle) throw MhaqError("the C-ABI library is not bound (mhaq_amd._ext.bind)");
}

inline void check(int rc, const char* what) {
  if (rc != 0)
    throw MhaqError(std::string(what) + " failed: " + A.mhaq_fq_error_string(rc) + " (code " + std::to_string(rc) + ")");
}

inline void* cur_stream(const Tensor& t) {
  return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}

inline bool capturing() {
  return c10::hip::currentStreamCaptureStatusMayInitCtx() != c10::hip::CaptureStatus::None;
}

// The C ABI launches on the stream it is handed and expects the calling thread's CURRENT device to be that stream's (and the
// pointers') device -- include/mhaq_fq.h, "Devices".  torch ops work on a tensor's device whatever the current one is (the
// reference is plain torch ops: a model moved to cuda:1 in a process whose current device is cuda:0 just works), so every
// forward entry point below switches to its input's device for its own duration.  (Backward nodes run on the autograd
// engine's thread of their device, which has made it current.)  ~50 ns when the device already is the current one.
#define MHAQ_ON_DEVICE_OF(t) const c10::OptionalDeviceGuard mhaq_device_guard_(at::device_of(t))

inline const float* fptr(const Tensor& t) { return static_cast<const float*>(t.const_data_ptr()); }
inline float* fptr_mut(const Tensor& t) { return static_cast<float*>(t.mutable_data_ptr()); }
inline const float* fptr_or_null(const Tensor& t) { return t.defined() ? fptr(t) : nullptr; }

// the upstream gradient in the memory order of x (the kernels walk both as flat streams)
inline Tensor like_layout(const Tensor& g, const Tensor& x) {
  if (g.sizes() == x.sizes() && g.strides() == x.strides()) return g;
  Tensor out = at::empty_like(x);
  out.copy_(g);
  return out;
}

// ------------------------------------------------------------------------------------------------ host timers
// Where does the host time of a node go?  Cheap enough to stay compiled in (steady_clock::now() is ~25 ns); read and
// reset with host_timers() (tools/host_profile.py).  Relaxed atomics: the backward nodes run on the autograd thread.
enum { T_ACT_FWD, T_ACT_FWD_LAUNCH, T_ACT_BWD, T_ACT_BWD_SAVED, T_ACT_BWD_ALLOC, T_ACT_BWD_HUB, T_ACT_BWD_LAUNCH, T_HUB_BWD,
       T_N };
const char* kTimerNames[T_N] = {"act_fwd", "act_fwd_launch", "act_bwd", "act_bwd_saved", "act_bwd_alloc", "act_bwd_hub",
                                "act_bwd_launch", "hub_bwd"};
std::atomic<int64_t> g_tns[T_N];
std::atomic<int64_t> g_tcnt[T_N];
inline int64_t now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline void tick(int which, int64_t t0, int64_t t1) {
  g_tns[which].fetch_add(t1 - t0, std::memory_order_relaxed);
  g_tcnt[which].fetch_add(1, std::memory_order_relaxed);
}

// ------------------------------------------------------------------------------------------------ sign streams
// (seed, offset) source of the in-kernel Philox sign stream (gdnsq.py:54 randint_like; SURVEY.md section 8e: ranks
// draw different streams, every backward call gets a fresh offset).  One owner for the Python ops (ctypes) and the
// nodes below: mhaq_amd.ops.rng is a facade over this object.
struct Rng {
  std::mutex mu;
  bool seeded = false;
  uint64_t seed = 0, count = 0;
  Tensor base;   // nullable one-element int64 device tensor the kernels add to their host offset (hipGraph replays)
};
Rng& R = *new Rng();     // never destroyed (it may hold a device tensor): see the registries below

constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;

struct Draw {
  uint64_t seed = 0, offset = 0;
  const uint64_t* offset_dev = nullptr;
};

// what ops._signs does for a backward call: none for explicit signs / LSQ, else the next offset of the rank's stream
Draw draw_signs(bool has_r_sign, int64_t method, int64_t rank, const Tensor& like) {
  Draw d;
  if (has_r_sign || method == MHAQ_FQ_LSQ) return d;
  std::lock_guard<std::mutex> g(R.mu);
  if (!R.seeded) throw MhaqError("sign stream used before it was seeded (mhaq_amd.ops.manual_seed)");
  d.seed = R.seed ^ ((uint64_t)rank * kGolden);
  d.offset = ++R.count;
  if (R.base.defined() && R.base.device() == like.device())
    d.offset_dev = static_cast<const uint64_t*>(R.base.const_data_ptr());
  return d;
}

// ------------------------------------------------------------------------------------------------ activation hub
// One finalize launch for every NoisyAct quantizer of a backward pass (mhaq_amd/act_hub.py describes the scheme).
// Lifetime rules for device memory a captured hipGraph may have baked into its launches: a descriptor table or
// workspace that was handed out WHILE A CAPTURE WAS ACTIVE is held until release_captured(); everything else is
// ordinary caching-allocator memory -- eager tables live in a small LRU, an outgrown eager workspace is simply
// dropped (the allocator orders its reuse behind the kernels that read it).
struct Hub {
  struct Pending { int64_t slot, nparts; Tensor ws; };
  struct Table { Tensor dev, host; bool captured = false; uint64_t stamp = 0; };
  std::mutex mu;
  int64_t n = 0;
  std::vector<Tensor> ws;
  std::vector<char> ws_captured;
  std::vector<Pending> pending;
  std::map<std::vector<int64_t>, Table> tables;
  std::vector<Tensor> retired;
  std::vector<std::vector<int64_t>> shapes;   // parameter shapes of this step's begin()
  std::unordered_map<int, Tensor> placeholders;
  Tensor last_table;
  uint64_t clock = 0;
  static constexpr size_t kEagerTables = 4;

  Tensor workspace(int64_t slot, int64_t nbytes, const Tensor& like) {
    Tensor& w = ws.at(slot);
    const bool cap = capturing();
    if (!w.defined() || w.numel() < nbytes || w.device() != like.device()) {
      if (w.defined() && ws_captured[slot]) retired.push_back(w);
      w = at::empty({nbytes}, like.options().dtype(at::kByte));
      ws_captured[slot] = 0;
    }
    if (cap) ws_captured[slot] = 1;
    return w;
  }

  Tensor placeholder(const Tensor& like) {
    Tensor& p = placeholders[like.device().index()];
    if (!p.defined()) p = at::zeros({1}, like.options());
    return p;
  }

  void release_captured() {
    std::lock_guard<std::mutex> g(mu);
    retired.clear();
    std::fill(ws_captured.begin(), ws_captured.end(), 0);
    for (auto it = tables.begin(); it != tables.end();) it = it->second.captured ? tables.erase(it) : std::next(it);
  }
};

// The registries are never destroyed (heap-allocated on first use): a static map would free device tensors and HIP events
// from a static destructor, after the HIP runtime and the interpreter are gone.
std::mutex g_hub_mu;
std::unordered_map<int64_t, std::shared_ptr<Hub>>& g_hubs = *new std::unordered_map<int64_t, std::shared_ptr<Hub>>();
int64_t g_next_hub = 1;

std::shared_ptr<Hub> hub_get(int64_t id) {
  std::lock_guard<std::mutex> g(g_hub_mu);
  auto it = g_hubs.find(id);
  if (it == g_hubs.end()) throw MhaqError("ActGradHub " + std::to_string(id) + " no longer exists");
  return it->second;
}

int64_t hub_create(int64_t n) {
  auto h = std::make_shared<Hub>();
  h->n = n;
  h->ws.resize(n);
  h->ws_captured.assign(n, 0);
  std::lock_guard<std::mutex> g(g_hub_mu);
  g_hubs[g_next_hub] = h;
  return g_next_hub++;
}

void hub_destroy(int64_t id) {
  std::lock_guard<std::mutex> g(g_hub_mu);
  g_hubs.erase(id);
}

// pinned staging + async copy on the current stream; the caching host allocator keeps `host` until the copy has run
std::pair<Tensor, Tensor> upload(const void* bytes, size_t nbytes, const Tensor& like) {
  Tensor host = at::empty({(int64_t)nbytes}, at::TensorOptions().dtype(at::kByte).pinned_memory(true));
  std::memcpy(host.mutable_data_ptr(), bytes, nbytes);
  Tensor dev = host.to(like.device(), /*non_blocking=*/true);
  return {dev, host};
}

class HubFn : public torch::autograd::Function<HubFn> {
 public:
  static variable_list forward(AutogradContext* ctx, int64_t hub_id, at::TensorList params) {
    ctx->saved_data["hub"] = hub_id;
    ctx->set_materialize_grads(false);
    auto hub = hub_get(hub_id);
    std::lock_guard<std::mutex> g(hub->mu);
    hub->pending.clear();
    hub->shapes.clear();
    variable_list out;
    out.reserve(params.size());
    for (const auto& p : params) {
      hub->shapes.emplace_back(p.sizes().vec());
      out.push_back(p.view_as(p));          // aliases: the kernels read the parameters in place
    }
    return out;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    struct Scope { int64_t t0 = now_ns(); ~Scope() { tick(T_HUB_BWD, t0, now_ns()); } } scope;
    auto hub = hub_get(ctx->saved_data["hub"].toInt());
    std::lock_guard<std::mutex> g(hub->mu);
    variable_list out(1 + grads.size());
    std::vector<Hub::Pending> pending;
    pending.swap(hub->pending);
    if (pending.empty()) return out;
    const Tensor& like = pending[0].ws;
    std::vector<int64_t> key;
    key.reserve(3 * pending.size());
    for (auto& p : pending) {
      key.push_back(p.slot);
      key.push_back(p.nparts);
      key.push_back((int64_t)(uintptr_t)p.ws.const_data_ptr());
    }
    const bool cap = capturing();
    auto it = hub->tables.find(key);
    if (it == hub->tables.end()) {
      std::vector<mhaq_act_finalize_desc> descs(pending.size());
      for (size_t j = 0; j < pending.size(); ++j)
        descs[j] = mhaq_act_finalize_desc{static_cast<const float*>(pending[j].ws.const_data_ptr()), pending[j].nparts};
      auto up = upload(descs.data(), descs.size() * sizeof(descs[0]), like);
      if (!cap) {   // eager tables: a small LRU (a table the stream still reads stays valid: the allocator orders reuse)
        size_t eager = 0;
        for (auto& kv : hub->tables) eager += !kv.second.captured;
        while (eager >= Hub::kEagerTables) {
          auto victim = hub->tables.end();
          for (auto jt = hub->tables.begin(); jt != hub->tables.end(); ++jt)
            if (!jt->second.captured && (victim == hub->tables.end() || jt->second.stamp < victim->second.stamp)) victim = jt;
          hub->tables.erase(victim);
          --eager;
        }
      }
      it = hub->tables.emplace(key, Hub::Table{up.first, up.second, cap, 0}).first;
    }
    it->second.captured = it->second.captured || cap;
    it->second.stamp = ++hub->clock;
    hub->last_table = it->second.dev;
    Tensor slab = at::empty({(int64_t)pending.size(), 3}, like.options().dtype(at::kFloat));
    check(A.mhaq_fq_act_bwd_finalize_multi(static_cast<const mhaq_act_finalize_desc*>(it->second.dev.const_data_ptr()),
                                           (int)pending.size(), fptr_mut(slab), cur_stream(slab)),
          "mhaq_fq_act_bwd_finalize_multi");
    // one unbind makes all 3n one-element views (0.15 us each; select + narrow + view per gradient cost 1.2 us)
    const std::vector<Tensor> pieces = slab.view({(int64_t)pending.size() * 3, 1}).unbind(0);
    for (size_t j = 0; j < pending.size(); ++j) {
      const int64_t s = pending[j].slot;
      for (int c = 0; c < 3; ++c) {
        const size_t k = (size_t)(3 * s + c);
        if (k < grads.size() && grads[k].defined()) {    // autograd asked for it (requires_grad + reached)
          const auto& shp = hub->shapes.at(k);
          const Tensor& piece = pieces[3 * j + c];
          out[1 + k] = (shp.size() == 1 && shp[0] == 1) ? piece : piece.view(shp);
        }
      }
    }
    return out;
  }
};

// (the parameters go in as an at::TensorList: Function<T>::apply only unpacks that list type into autograd inputs)
variable_list hub_begin(int64_t hub_id, const variable_list& params) { need_lib(); return HubFn::apply(hub_id, at::TensorList(params)); }      // (no launch: aliases only)

// ------------------------------------------------------------------------------------------------ NoisyAct layer op
// NoisyAct.forward from its learnable parameters (gdnsq_act.py:39-55): returns (y, params[5] = {s, zp, lo, hi, qr}).
class ActLayerFn : public torch::autograd::Function<ActLayerFn> {
 public:
  static variable_list forward(AutogradContext* ctx, const Tensor& x, const Tensor& log_s, const Tensor& log_q,
                               const Tensor& b, int64_t method, const std::optional<Tensor>& r_sign, int64_t hub_id,
                               int64_t slot, int64_t rank) {
    Tensor y = at::empty_like(x);
    Tensor params = at::empty({5}, x.options());
    const int64_t tl0 = now_ns();
    check(A.mhaq_fq_act_fwd(fptr(x), fptr_mut(y), x.numel(), fptr(log_s), fptr(log_q), fptr(b), fptr_mut(params),
                            nullptr, nullptr, nullptr, 0, cur_stream(x)),
          "mhaq_fq_act_fwd");
    tick(T_ACT_FWD_LAUNCH, tl0, now_ns());
    if (r_sign.has_value() && r_sign->defined()) ctx->save_for_backward({x, params, *r_sign});
    else ctx->save_for_backward({x, params});
    ctx->saved_data["i"] = std::vector<int64_t>{method, hub_id, slot, rank};       // one map entry instead of four
    if (hub_id <= 0)            // the hub path hands out placeholders: no shapes needed
      ctx->saved_data["shapes"] = std::vector<std::vector<int64_t>>{log_s.sizes().vec(), log_q.sizes().vec(), b.sizes().vec()};
    ctx->mark_non_differentiable({params});
    return {y, params};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const int64_t t0 = now_ns();
    auto saved = ctx->get_saved_variables();
    const Tensor& x = saved[0];
    const Tensor& params = saved[1];
    const bool has_r = saved.size() > 2;
    const std::vector<int64_t> ic = ctx->saved_data["i"].toIntVector();
    const int64_t method = ic[0], hub_id = ic[1], slot_i = ic[2], rank_i = ic[3];
    const int64_t t1 = now_ns();
    Tensor g = like_layout(grads[0], x);
    Tensor gx = at::empty_like(x);
    const int64_t t2 = now_ns();
    tick(T_ACT_BWD_SAVED, t0, t1);
    tick(T_ACT_BWD_ALLOC, t1, t2);
    const int64_t n = x.numel();
    const size_t nb = A.mhaq_fq_act_bwd_workspace_bytes(n);
    const Draw d = draw_signs(has_r, method, rank_i, x);
    const int8_t* r = has_r ? static_cast<const int8_t*>(saved[2].const_data_ptr()) : nullptr;
    variable_list out(9);
    const bool nx = ctx->needs_input_grad(0), ns = ctx->needs_input_grad(1), nq = ctx->needs_input_grad(2),
               nbias = ctx->needs_input_grad(3);
    if (hub_id > 0) {
      auto hub = hub_get(hub_id);
      std::lock_guard<std::mutex> lk(hub->mu);
      const int64_t slot = slot_i;
      Tensor ws = hub->workspace(slot, (int64_t)nb, x);
      int32_t nparts = 0;
      const int64_t t3 = now_ns();
      check(A.mhaq_fq_act_bwd_partials(fptr(x), fptr(g), fptr_mut(gx), n, fptr(params), (int)method, r, d.seed, d.offset,
                                       d.offset_dev, ws.mutable_data_ptr(), nb, &nparts, cur_stream(x)),
            "mhaq_fq_act_bwd_partials");
      const int64_t t4 = now_ns();
      hub->pending.push_back(Hub::Pending{slot, nparts, ws});
      Tensor ph = hub->placeholder(x);
      if (nx) out[0] = gx;
      if (ns) out[1] = ph;
      if (nq) out[2] = ph;
      if (nbias) out[3] = ph;
      tick(T_ACT_BWD_HUB, t2, t3);
      tick(T_ACT_BWD_LAUNCH, t3, t4);
      tick(T_ACT_BWD, t0, now_ns());
      return out;
    }
    Tensor gr = at::empty({3}, x.options());
    Tensor ws = at::empty({(int64_t)nb}, x.options().dtype(at::kByte));
    check(A.mhaq_fq_act_bwd(fptr(x), fptr(g), fptr_mut(gx), n, fptr(params), (int)method, r, d.seed, d.offset,
                            d.offset_dev, fptr_mut(gr), ws.mutable_data_ptr(), nb, cur_stream(x)),
          "mhaq_fq_act_bwd");
    const auto shapes = ctx->saved_data["shapes"].to<std::vector<std::vector<int64_t>>>();
    if (nx) out[0] = gx;
    if (ns) out[1] = gr.narrow(0, 0, 1).view(shapes[0]);
    if (nq) out[2] = gr.narrow(0, 1, 1).view(shapes[1]);
    if (nbias) out[3] = gr.narrow(0, 2, 1).view(shapes[2]);
    return out;
  }
};

// (y, params, s = params[0:1], hi = params[3:4]): the two views are what NoisyAct publishes on its Quantizer
std::tuple<Tensor, Tensor, Tensor, Tensor> act_layer(const Tensor& x_in, const Tensor& log_s, const Tensor& log_q,
                                                     const Tensor& b, int64_t method,
                                                     const std::optional<Tensor>& r_sign, int64_t hub_id, int64_t slot,
                                                     int64_t rank) {
  need_lib();
  MHAQ_ON_DEVICE_OF(x_in);
  TORCH_CHECK(x_in.is_cuda() && x_in.scalar_type() == at::kFloat, "act_layer: x must be a float32 device tensor");
  TORCH_CHECK(log_s.numel() == 1 && log_q.numel() == 1 && b.numel() == 1 && log_s.is_cuda() && log_q.is_cuda() &&
                  b.is_cuda() && log_s.scalar_type() == at::kFloat && log_q.scalar_type() == at::kFloat &&
                  b.scalar_type() == at::kFloat,
              "act_layer: log_act_s / log_act_q / act_b must be one-element float32 device tensors");
  const int64_t t0 = now_ns();
  const Tensor x = x_in.is_non_overlapping_and_dense() ? x_in : x_in.contiguous();
  auto out = ActLayerFn::apply(x, log_s, log_q, b, method, r_sign, hub_id, slot, rank);
  const Tensor& params = out[1];
  std::tuple<Tensor, Tensor, Tensor, Tensor> res{out[0], params, params.narrow(0, 0, 1), params.narrow(0, 3, 1)};
  tick(T_ACT_FWD, t0, now_ns());
  return res;
}

// ------------------------------------------------------------------------------------------------ weight layer ops
struct Pre {   // this step's forward of the layer out of the model-wide launch (multi.py, forward-only mode)
  bool has = false;
  Tensor wq, s, zp, mx, lwq;
};

// set from Python: the packed AEWGS statistics exchange (takes the GIL).  A raw, never-released reference: static
// destructors run after the interpreter is gone.
PyObject* g_allreduce_avg = nullptr;
std::atomic<bool> g_dist_active{false};

void allreduce_avg(const Tensor& t) {
  py::gil_scoped_acquire gil;
  if (!g_allreduce_avg) throw MhaqError("AEWGS statistics exchange requested but no all-reduce is installed");
  py::reinterpret_borrow<py::object>(g_allreduce_avg)(t);
}

// Per-channel NoisyConv2d weight path from log_wght_s plus the layer's regulariser input lwq = log2(max - min + s)
// (gdnsq_conv2d.py:71-98, model_helper.py:24-44): returns (wq, zp, s, lwq).
class WeightLayerFn : public torch::autograd::Function<WeightLayerFn> {
 public:
  static variable_list forward(AutogradContext* ctx, const Tensor& w, const Tensor& log_s, int64_t method,
                               const std::optional<Tensor>& r_sign, bool zp_grad, const Pre& pre, int64_t rank,
                               bool distributed) {
    const int64_t co = w.size(0), row = co ? w.numel() / co : 0;
    Tensor wq, s, zp, mx, lwq;
    if (pre.has) {
      wq = pre.wq; s = pre.s; zp = pre.zp; mx = pre.mx; lwq = pre.lwq;
    } else {
      wq = at::empty_like(w);
      Tensor aux = at::empty({4, co}, w.options());
      s = aux.select(0, 0); zp = aux.select(0, 1); mx = aux.select(0, 2); lwq = aux.select(0, 3);
      check(A.mhaq_fq_wlayer_fwd(fptr(w), fptr_mut(wq), fptr(log_s), co, row, fptr_mut(s), fptr_mut(zp), fptr_mut(mx),
                                 fptr_mut(lwq), cur_stream(w)),
            "mhaq_fq_wlayer_fwd");
    }
    if (r_sign.has_value() && r_sign->defined()) ctx->save_for_backward({w, s, zp, mx, *r_sign});
    else ctx->save_for_backward({w, s, zp, mx});
    ctx->saved_data["method"] = method;
    ctx->saved_data["rank"] = rank;
    ctx->saved_data["dist"] = distributed;
    ctx->saved_data["ls_shape"] = log_s.sizes().vec();
    ctx->set_materialize_grads(false);
    // zp_grad = the quantized-bias mode (gdnsq_conv2d.py:86-94): the bias quantizer reuses this layer's s and zp and
    // sends gradient into both, so both stay differentiable outputs
    if (!zp_grad) ctx->mark_non_differentiable({s, zp});
    return {wq, zp, s, lwq};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto saved = ctx->get_saved_variables();
    const Tensor &w = saved[0], &s = saved[1], &zp = saved[2], &mx = saved[3];
    const bool has_r = saved.size() > 4;
    const int64_t method = ctx->saved_data["method"].toInt();
    Tensor G = grads[0].defined() ? like_layout(grads[0], w) : at::zeros_like(w);
    Tensor gzp_extra = grads[1].defined() ? grads[1].contiguous() : Tensor();
    Tensor g_lwq = grads[3].defined() ? grads[3].contiguous() : Tensor();
    const int64_t co = w.size(0), row = co ? w.numel() / co : 0;
    Tensor stats;
    if (method == MHAQ_FQ_AEWGS && ctx->saved_data["dist"].toBool()) {
      stats = at::empty({3, co}, w.options());
      check(A.mhaq_fq_pc_aewgs_stats(fptr(w), fptr(G), fptr(s), fptr(zp), co, row, fptr_mut(stats), cur_stream(w)),
            "mhaq_fq_pc_aewgs_stats");
      allreduce_avg(stats);          // gdnsq.py:126-129, one packed message
    }
    Tensor gw = at::empty_like(w);
    Tensor gls = at::empty({co}, w.options());
    const Draw d = draw_signs(has_r, method, ctx->saved_data["rank"].toInt(), w);
    check(A.mhaq_fq_wlayer_bwd(fptr(w), fptr(G), fptr_mut(gw), fptr_mut(gls), fptr(s), fptr(zp), fptr(mx),
                               fptr_or_null(g_lwq), co, row, (int)method, fptr_or_null(stats), fptr_or_null(gzp_extra),
                               has_r ? static_cast<const int8_t*>(saved[4].const_data_ptr()) : nullptr, d.seed, d.offset,
                               d.offset_dev, cur_stream(w)),
          "mhaq_fq_wlayer_bwd");
    if (grads[2].defined())     // gradient reaching s from the quantized bias: exp2 backward, grad * s * ln2
      gls = gls + at::mul(at::mul(grads[2].reshape({co}), s), 0.69314718055994531);
    variable_list out(8);
    out[0] = gw;
    out[1] = gls.view(ctx->saved_data["ls_shape"].toIntVector());
    return out;
  }
};

// returns (wq, zp [co,1,..], s [co,1,..], lwq [co])
std::tuple<Tensor, Tensor, Tensor, Tensor> weight_layer(const Tensor& w_in, const Tensor& log_s_in, int64_t method,
                                                        const std::optional<Tensor>& r_sign, bool zp_grad,
                                                        const std::optional<std::vector<Tensor>>& pre, int64_t rank) {
  need_lib();
  MHAQ_ON_DEVICE_OF(w_in);
  TORCH_CHECK(w_in.is_cuda() && w_in.scalar_type() == at::kFloat && log_s_in.is_cuda() &&
                  log_s_in.scalar_type() == at::kFloat,
              "weight_layer: weight and log_wght_s must be float32 device tensors");
  TORCH_CHECK(w_in.dim() >= 1 && log_s_in.numel() == w_in.size(0), "per-channel log scale must have ", w_in.size(0),
              " elements, got ", log_s_in.sizes());
  const Tensor w = w_in.is_non_overlapping_and_dense() ? w_in : w_in.contiguous();
  const Tensor log_s = log_s_in.is_contiguous() ? log_s_in : log_s_in.contiguous();
  Pre p;
  if (pre.has_value()) {
    TORCH_CHECK(pre->size() >= 5, "weight_layer: pre = (wq, s, zp, mx, lwq[, s and zp in the published shape])");
    p.has = true;
    p.wq = (*pre)[0]; p.s = (*pre)[1]; p.zp = (*pre)[2]; p.mx = (*pre)[3]; p.lwq = (*pre)[4];
  }
  auto out = WeightLayerFn::apply(w, log_s, method, r_sign, zp_grad, p, rank, g_dist_active.load());
  std::vector<int64_t> shp(w.dim(), 1);
  shp[0] = w.size(0);
  return {out[0], out[1].view(shp), out[2].view(shp), out[3]};
}

// PER_TENSOR weight layer small enough for one workgroup (every CIFAR ResNet-20 / RFDN layer): one launch per
// direction from log_wght_s, regulariser input included.  Returns (wq, aux[4] = {s, zp, max, lwq}, lwq[1]).
class WeightLayerPTFn : public torch::autograd::Function<WeightLayerPTFn> {
 public:
  static variable_list forward(AutogradContext* ctx, const Tensor& w, const Tensor& log_s, int64_t method,
                               const std::optional<Tensor>& r_sign, int64_t rank) {
    Tensor wq = at::empty_like(w);
    Tensor aux = at::empty({4}, w.options());
    check(A.mhaq_fq_wlayer_pt_fwd(fptr(w), fptr_mut(wq), fptr(log_s), w.numel(), fptr_mut(aux), cur_stream(w)),
          "mhaq_fq_wlayer_pt_fwd");
    Tensor lwq = aux.narrow(0, 3, 1).clone();
    if (r_sign.has_value() && r_sign->defined()) ctx->save_for_backward({w, aux, *r_sign});
    else ctx->save_for_backward({w, aux});
    ctx->saved_data["method"] = method;
    ctx->saved_data["rank"] = rank;
    ctx->saved_data["ls_shape"] = log_s.sizes().vec();
    ctx->set_materialize_grads(false);
    ctx->mark_non_differentiable({aux});
    return {wq, aux, lwq};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto saved = ctx->get_saved_variables();
    const Tensor &w = saved[0], &aux = saved[1];
    const bool has_r = saved.size() > 2;
    const int64_t method = ctx->saved_data["method"].toInt();
    Tensor G = grads[0].defined() ? grads[0].contiguous() : at::zeros_like(w);
    Tensor g_lwq = grads[2].defined() ? grads[2].contiguous() : Tensor();
    Tensor gw = at::empty_like(w);
    Tensor gls = at::empty({1}, w.options());
    const Draw d = draw_signs(has_r, method, ctx->saved_data["rank"].toInt(), w);
    check(A.mhaq_fq_wlayer_pt_bwd(fptr(w), fptr(G), fptr_mut(gw), fptr_mut(gls), fptr(aux), fptr_or_null(g_lwq), w.numel(),
                                  (int)method, has_r ? static_cast<const int8_t*>(saved[2].const_data_ptr()) : nullptr,
                                  d.seed, d.offset, d.offset_dev, cur_stream(w)),
          "mhaq_fq_wlayer_pt_bwd");
    variable_list out(5);
    out[0] = gw;
    out[1] = gls.view(ctx->saved_data["ls_shape"].toIntVector());
    return out;
  }
};

// returns (wq, zp 0-dim, s [1], lwq [1])
std::tuple<Tensor, Tensor, Tensor, Tensor> weight_layer_pt(const Tensor& w_in, const Tensor& log_s, int64_t method,
                                                           const std::optional<Tensor>& r_sign, int64_t rank) {
  need_lib();
  MHAQ_ON_DEVICE_OF(w_in);
  TORCH_CHECK(w_in.is_cuda() && w_in.scalar_type() == at::kFloat && log_s.is_cuda() &&
                  log_s.scalar_type() == at::kFloat && log_s.numel() == 1,
              "weight_layer_pt: weight and a one-element log_wght_s must be float32 device tensors");
  const Tensor w = w_in.contiguous();
  auto out = WeightLayerPTFn::apply(w, log_s, method, r_sign, rank);
  const Tensor& aux = out[1];
  return {out[0], aux.select(0, 1), aux.narrow(0, 0, 1), out[2]};
}

// PER_TENSOR weight layer of any size (and every PER_TENSOR AEWGS layer): streaming launches, regulariser input and its
// amin / amax backward included (mhaq_fq_wlayer_ptl_fwd / _bwd).  Returns (wq, aux[7], lwq[1]).
class WeightLayerPTLFn : public torch::autograd::Function<WeightLayerPTLFn> {
 public:
  static variable_list forward(AutogradContext* ctx, const Tensor& w, const Tensor& log_s, int64_t method,
                               const std::optional<Tensor>& r_sign, int64_t rank, bool distributed) {
    Tensor wq = at::empty_like(w);
    Tensor aux = at::empty({7}, w.options());
    const size_t nb = A.mhaq_fq_wlayer_ptl_workspace_bytes(w.numel());
    Tensor ws = at::empty({(int64_t)nb}, w.options().dtype(at::kByte));
    check(A.mhaq_fq_wlayer_ptl_fwd(fptr(w), fptr_mut(wq), fptr(log_s), w.numel(), fptr_mut(aux), ws.mutable_data_ptr(), nb,
                                   cur_stream(w)),
          "mhaq_fq_wlayer_ptl_fwd");
    Tensor lwq = aux.narrow(0, 3, 1).clone();
    if (r_sign.has_value() && r_sign->defined()) ctx->save_for_backward({w, aux, *r_sign});
    else ctx->save_for_backward({w, aux});
    ctx->saved_data["method"] = method;
    ctx->saved_data["rank"] = rank;
    ctx->saved_data["dist"] = distributed;
    ctx->saved_data["ls_shape"] = log_s.sizes().vec();
    ctx->set_materialize_grads(false);
    ctx->mark_non_differentiable({aux});
    return {wq, aux, lwq};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto saved = ctx->get_saved_variables();
    const Tensor &w = saved[0], &aux = saved[1];
    const bool has_r = saved.size() > 2;
    const int64_t method = ctx->saved_data["method"].toInt();
    Tensor G = grads[0].defined() ? like_layout(grads[0], w) : at::zeros_like(w);
    Tensor g_lwq = grads[2].defined() ? grads[2].contiguous() : Tensor();
    const int64_t n = w.numel();
    void* stream = cur_stream(w);
    Tensor stats;
    int64_t period = 0;
    if (method == MHAQ_FQ_AEWGS) {
      // statistics of a [1]-shaped scale: means over dim 0 per position (gdnsq.py:150-152), then the all-reduce of
      // gdnsq.py:126-129 as ONE packed [3, period] message
      const int64_t co = w.size(0);
      period = n / co;
      stats = at::empty({3, period}, w.options());
      const size_t sb = A.mhaq_fq_pt_aewgs_colstats_workspace_bytes(co, period);
      Tensor sws = sb ? at::empty({(int64_t)sb}, w.options().dtype(at::kByte)) : Tensor();
      check(A.mhaq_fq_pt_aewgs_colstats(fptr(w), fptr(G), co, period, fptr(aux), fptr(aux) + 1, nullptr, nullptr,
                                        fptr_mut(stats), sb ? sws.mutable_data_ptr() : nullptr, sb, stream),
            "mhaq_fq_pt_aewgs_colstats");
      if (ctx->saved_data["dist"].toBool()) allreduce_avg(stats);
    }
    Tensor gw = at::empty_like(w);
    Tensor gls = at::empty({1}, w.options());
    const size_t nb = A.mhaq_fq_wlayer_ptl_workspace_bytes(n);
    Tensor ws = at::empty({(int64_t)nb}, w.options().dtype(at::kByte));
    const Draw d = draw_signs(has_r, method, ctx->saved_data["rank"].toInt(), w);
    check(A.mhaq_fq_wlayer_ptl_bwd(fptr(w), fptr(G), fptr_mut(gw), fptr_mut(gls), fptr(aux), fptr_or_null(g_lwq), n,
                                   (int)method, fptr_or_null(stats), period,
                                   has_r ? static_cast<const int8_t*>(saved[2].const_data_ptr()) : nullptr, d.seed,
                                   d.offset, d.offset_dev, ws.mutable_data_ptr(), nb, stream),
          "mhaq_fq_wlayer_ptl_bwd");
    variable_list out(6);
    out[0] = gw;
    out[1] = gls.view(ctx->saved_data["ls_shape"].toIntVector());
    return out;
  }
};

// returns (wq, zp 0-dim, s [1], lwq [1])
std::tuple<Tensor, Tensor, Tensor, Tensor> weight_layer_ptl(const Tensor& w_in, const Tensor& log_s, int64_t method,
                                                            const std::optional<Tensor>& r_sign, int64_t rank) {
  need_lib();
  MHAQ_ON_DEVICE_OF(w_in);
  TORCH_CHECK(w_in.is_cuda() && w_in.scalar_type() == at::kFloat && log_s.is_cuda() &&
                  log_s.scalar_type() == at::kFloat && log_s.numel() == 1 && w_in.numel() > 0 && w_in.dim() >= 1,
              "weight_layer_ptl: a non-empty weight and a one-element log_wght_s must be float32 device tensors");
  // any dense layout with dim 0 outermost (contiguous, channels_last): the quantizer is elementwise and the AEWGS
  // statistics are per physical position within a dim-0 slice
  const Tensor w = w_in.is_non_overlapping_and_dense() ? w_in : w_in.contiguous();
  auto out = WeightLayerPTLFn::apply(w, log_s, method, r_sign, rank, g_dist_active.load());
  const Tensor& aux = out[1];
  return {out[0], aux.select(0, 1), aux.narrow(0, 0, 1), out[2]};
}

// ------------------------------------------------------------------------------------------------ weight plan / groups
// The model-wide weight forward and the grouped backward of mhaq_amd/multi.py (MultiTensorWeightQuant with
// joint_backward=False, _WeightGroup, _TablePool), host side in C++.
struct TablePool {
  // Device descriptor tables for launches whose pointers (the dL/dWq tensors autograd hands over) are only known at
  // backward time.  Device tables and pinned staging buffers are allocated up front: inside a hipGraph capture
  // nothing may be allocated, and a captured upload re-reads its staging buffer at every replay -- so an entry
  // filled during a capture is never reused (until release_captured()).
  std::vector<Tensor> dev, host;
  std::vector<std::vector<int64_t>> keys;
  std::vector<char> used, held;
  std::vector<hipEvent_t> events;
  std::vector<uint64_t> stamp;
  uint64_t clock = 0;

  TablePool(size_t nbytes, const Tensor& like, int size = 8) {
    for (int i = 0; i < size; ++i) {
      dev.push_back(at::empty({(int64_t)nbytes}, like.options().dtype(at::kByte)));
      host.push_back(at::empty({(int64_t)nbytes}, at::TensorOptions().dtype(at::kByte).pinned_memory(true)));
    }
    keys.resize(size);
    used.assign(size, 0);
    held.assign(size, 0);
    events.assign(size, nullptr);
    stamp.assign(size, 0);
  }
  ~TablePool() {
    for (auto e : events)
      if (e) (void)hipEventDestroy(e);
  }

  template <class Fill>
  const Tensor& get(const std::vector<int64_t>& key, Fill fill, void* stream) {
    ++clock;
    const bool cap = capturing();
    const int n = (int)keys.size();
    for (int i = 0; i < n; ++i)
      if (used[i] && keys[i] == key && (held[i] || !cap)) { stamp[i] = clock; return dev[i]; }
    int pick = -1;
    for (int i = 0; i < n && pick < 0; ++i)
      if (!used[i]) pick = i;
    if (pick < 0) {
      // (waiting on an eager upload's event is not a capturable call: a capture only takes unused entries)
      if (cap) throw MhaqError("weight-group descriptor tables: no unused entry left for a captured launch");
      for (int i = 0; i < n; ++i)
        if (!held[i] && (pick < 0 || stamp[i] < stamp[pick])) pick = i;
      if (pick < 0) throw MhaqError("weight-group descriptor tables exhausted by captured graphs");
      if (events[pick]) C10_HIP_CHECK(hipEventSynchronize(events[pick]));   // the staging buffer's previous upload must have run
    }
    fill(host[pick].mutable_data_ptr());
    C10_HIP_CHECK(hipMemcpyAsync(dev[pick].mutable_data_ptr(), host[pick].const_data_ptr(), (size_t)dev[pick].numel(),
                                 hipMemcpyHostToDevice, (hipStream_t)stream));
    keys[pick] = key;
    used[pick] = 1;
    held[pick] = cap;
    stamp[pick] = clock;
    if (!cap) {
      if (!events[pick]) C10_HIP_CHECK(hipEventCreateWithFlags(&events[pick], hipEventDisableTiming));
      C10_HIP_CHECK(hipEventRecord(events[pick], (hipStream_t)stream));
    }
    return dev[pick];
  }

  void release_captured() {
    for (size_t i = 0; i < keys.size(); ++i)
      if (held[i]) { held[i] = 0; used[i] = 0; keys[i].clear(); }
  }
};

struct Plan;
struct Group {
  Plan* plan = nullptr;
  int64_t first = 0, n = 0, chan0 = 0, elem0 = 0, co = 0, elems = 0, max_row = 0, method = 0;
  std::unique_ptr<TablePool> pool;
};

struct Plan {
  std::mutex mu;
  int64_t nlayers = 0, total_elems = 0, total_co = 0, max_row = 0;
  std::vector<int64_t> co, row, elem_off, chan_off, methods, group_of;
  std::vector<Group> groups;
  // forward table cache: pointers of (weights, log scales) -> device table; captured entries are held, eager ones LRU
  struct FwdTable { Tensor dev, host; bool captured = false; uint64_t stamp = 0; };
  std::map<std::vector<int64_t>, FwdTable> fwd_tables;
  uint64_t clock = 0;
  Tensor cur_wq, cur_aux;   // this step's model-wide forward
  static constexpr size_t kEagerTables = 2;
};

std::mutex g_plan_mu;
std::unordered_map<int64_t, std::shared_ptr<Plan>>& g_plans = *new std::unordered_map<int64_t, std::shared_ptr<Plan>>();
int64_t g_next_plan = 1;

std::shared_ptr<Plan> plan_get(int64_t id) {
  std::lock_guard<std::mutex> g(g_plan_mu);
  auto it = g_plans.find(id);
  if (it == g_plans.end()) throw MhaqError("weight plan " + std::to_string(id) + " no longer exists");
  return it->second;
}

int64_t plan_create(std::vector<int64_t> co, std::vector<int64_t> row, std::vector<int64_t> methods,
                    std::vector<std::pair<int64_t, int64_t>> groups) {
  auto p = std::make_shared<Plan>();
  p->nlayers = (int64_t)co.size();
  TORCH_CHECK(row.size() == co.size() && methods.size() == co.size(), "plan_create: co / row / methods disagree");
  p->co = co; p->row = row; p->methods = methods;
  int64_t e = 0, c = 0;
  for (size_t i = 0; i < co.size(); ++i) {
    p->elem_off.push_back(e);
    p->chan_off.push_back(c);
    e += co[i] * row[i];
    c += co[i];
    p->max_row = std::max(p->max_row, row[i]);
  }
  p->total_elems = e;
  p->total_co = c;
  p->group_of.assign(co.size(), -1);
  for (auto& fl : groups) {
    TORCH_CHECK(0 <= fl.first && fl.first < fl.second && fl.second <= p->nlayers, "plan_create: bad group range");
    Group g;
    g.plan = p.get();
    g.first = fl.first;
    g.n = fl.second - fl.first;
    g.chan0 = p->chan_off[fl.first];
    g.elem0 = p->elem_off[fl.first];
    g.method = methods[fl.first];
    for (int64_t j = fl.first; j < fl.second; ++j) {
      g.co += co[j];
      g.elems += co[j] * row[j];
      g.max_row = std::max(g.max_row, row[j]);
      p->group_of[j] = (int64_t)p->groups.size();
    }
    p->groups.push_back(std::move(g));
  }
  std::lock_guard<std::mutex> gl(g_plan_mu);
  g_plans[g_next_plan] = p;
  return g_next_plan++;
}

void plan_destroy(int64_t id) {
  std::lock_guard<std::mutex> g(g_plan_mu);
  g_plans.erase(id);
}

// One launch quantizes every layer of the plan; returns (wq_all, aux_all, per layer: wq view with the weight's own
// strides, s, zp, mx, lwq slices of aux_all).
std::tuple<Tensor, Tensor, std::vector<std::vector<Tensor>>> plan_forward(int64_t plan_id, const std::vector<Tensor>& ws,
                                                                           const std::vector<Tensor>& lss) {
  need_lib();
  auto p = plan_get(plan_id);
  std::lock_guard<std::mutex> g(p->mu);
  const int64_t n = p->nlayers;
  TORCH_CHECK((int64_t)ws.size() == n && (int64_t)lss.size() == n, "plan_forward: expected ", n, " weights and log scales");
  c10::OptionalDeviceGuard device_guard;
  if (n > 0) device_guard.reset_device(ws[0].device());      // see MHAQ_ON_DEVICE_OF
  std::vector<int64_t> key;
  key.reserve(2 * n);
  for (int64_t i = 0; i < n; ++i) {
    const Tensor& w = ws[i];
    TORCH_CHECK(w.is_cuda() && w.scalar_type() == at::kFloat && w.is_non_overlapping_and_dense() &&
                    w.numel() == p->co[i] * p->row[i],
                "plan_forward: weight ", i, " must be a dense float32 device tensor of the planned size");
    TORCH_CHECK(lss[i].is_cuda() && lss[i].scalar_type() == at::kFloat && lss[i].is_contiguous() &&
                    lss[i].numel() == p->co[i],
                "plan_forward: log scale ", i, " must be a contiguous float32 device tensor of the planned size");
    key.push_back((int64_t)(uintptr_t)w.const_data_ptr());
  }
  for (int64_t i = 0; i < n; ++i) key.push_back((int64_t)(uintptr_t)lss[i].const_data_ptr());
  const Tensor& like = ws[0];
  const bool cap = capturing();
  auto it = p->fwd_tables.find(key);
  if (it == p->fwd_tables.end()) {
    std::vector<mhaq_wlayer_desc> descs(n);
    for (int64_t i = 0; i < n; ++i)
      descs[i] = mhaq_wlayer_desc{fptr(ws[i]), fptr(lss[i]), nullptr, nullptr, p->co[i], p->row[i], p->elem_off[i],
                                  p->chan_off[i]};
    auto up = upload(descs.data(), descs.size() * sizeof(descs[0]), like);
    if (!cap) {
      size_t eager = 0;
      for (auto& kv : p->fwd_tables) eager += !kv.second.captured;
      while (eager >= Plan::kEagerTables) {
        auto victim = p->fwd_tables.end();
        for (auto jt = p->fwd_tables.begin(); jt != p->fwd_tables.end(); ++jt)
          if (!jt->second.captured && (victim == p->fwd_tables.end() || jt->second.stamp < victim->second.stamp)) victim = jt;
        p->fwd_tables.erase(victim);
        --eager;
      }
    }
    it = p->fwd_tables.emplace(key, Plan::FwdTable{up.first, up.second, cap, 0}).first;
  }
  it->second.captured = it->second.captured || cap;
  it->second.stamp = ++p->clock;
  Tensor wq_all = at::empty({p->total_elems}, like.options());
  Tensor aux_all = at::empty({4, p->total_co}, like.options());
  check(A.mhaq_fq_wlayer_fwd_multi(static_cast<const mhaq_wlayer_desc*>(it->second.dev.const_data_ptr()), (int)n,
                                   p->total_co, p->max_row, fptr_mut(wq_all), fptr_mut(aux_all), cur_stream(like)),
        "mhaq_fq_wlayer_fwd_multi");
  p->cur_wq = wq_all;
  p->cur_aux = aux_all;
  std::vector<std::vector<Tensor>> per(n);
  for (int64_t i = 0; i < n; ++i) {
    // the slab holds each layer in the physical order of its weight (a channels_last weight is [Co][kh][kw][Ci] in
    // memory): give the slice the weight's own strides
    Tensor wq = at::as_strided(wq_all, ws[i].sizes(), ws[i].strides(), p->elem_off[i]);
    Tensor s_i = aux_all.select(0, 0).narrow(0, p->chan_off[i], p->co[i]);
    Tensor zp_i = aux_all.select(0, 1).narrow(0, p->chan_off[i], p->co[i]);
    std::vector<int64_t> shp(ws[i].dim(), 1);
    shp[0] = p->co[i];
    // ... and s / zp once more in the [co, 1, ..] shape the layer publishes on its Quantizer (saves two Python view calls)
    per[i] = {wq, s_i, zp_i, aux_all.select(0, 2).narrow(0, p->chan_off[i], p->co[i]),
              aux_all.select(0, 3).narrow(0, p->chan_off[i], p->co[i]), s_i.view(shp), zp_i.view(shp)};
  }
  return {wq_all, aux_all, per};
}

// The autograd node of one backward group: outputs are the group's slices of this step's model-wide forward (no launch
// here); its backward is ONE launch (plus, for AEWGS under data parallelism, one statistics launch and ONE packed
// all-reduce).
class WeightGroupFn : public torch::autograd::Function<WeightGroupFn> {
 public:
  static variable_list forward(AutogradContext* ctx, int64_t plan_id, int64_t gi, at::TensorList tensors, int64_t rank,
                               bool distributed) {
    auto p = plan_get(plan_id);
    std::lock_guard<std::mutex> lk(p->mu);
    const Group& grp = p->groups.at(gi);
    const int64_t n = grp.n;
    TORCH_CHECK((int64_t)tensors.size() == 2 * n, "weight group: expected ", 2 * n, " tensors");
    TORCH_CHECK(p->cur_wq.defined(), "weight group: the model-wide forward of this step has not run");
    ctx->saved_data["plan"] = plan_id;
    ctx->saved_data["group"] = gi;
    ctx->saved_data["rank"] = rank;
    ctx->saved_data["dist"] = distributed;
    ctx->saved_data["aux"] = p->cur_aux;
    std::vector<std::vector<int64_t>> ls_shapes;
    variable_list ws(tensors.begin(), tensors.begin() + n);
    for (int64_t k = 0; k < n; ++k) ls_shapes.push_back(tensors[n + k].sizes().vec());
    ctx->saved_data["ls_shapes"] = ls_shapes;
    ctx->save_for_backward(ws);
    ctx->set_materialize_grads(false);
    variable_list outs;
    outs.reserve(2 * n);
    for (int64_t k = 0; k < n; ++k) {
      const int64_t i = grp.first + k;
      outs.push_back(at::as_strided(p->cur_wq, ws[k].sizes(), ws[k].strides(), p->elem_off[i]));
    }
    for (int64_t k = 0; k < n; ++k) {
      const int64_t i = grp.first + k;
      outs.push_back(p->cur_aux.select(0, 3).narrow(0, p->chan_off[i], p->co[i]));
    }
    return outs;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto p = plan_get(ctx->saved_data["plan"].toInt());
    std::unique_lock<std::mutex> lk(p->mu);
    Group& grp = p->groups.at(ctx->saved_data["group"].toInt());
    const int64_t n = grp.n;
    auto ws = ctx->get_saved_variables();
    const Tensor aux_all = ctx->saved_data["aux"].toTensor();
    const Tensor& like = aux_all;
    void* stream = cur_stream(like);
    std::vector<Tensor> Gs(n), gl(n);
    for (int64_t k = 0; k < n; ++k) {
      Gs[k] = grads[k].defined() ? like_layout(grads[k], ws[k]) : at::zeros_like(ws[k]);
      if (grads[n + k].defined()) gl[k] = grads[n + k].contiguous();
    }
    if (!grp.pool) grp.pool = std::make_unique<TablePool>(sizeof(mhaq_wlayer_desc) * n, like);
    std::vector<int64_t> key;
    key.reserve(3 * n);
    for (int64_t k = 0; k < n; ++k) key.push_back((int64_t)(uintptr_t)ws[k].const_data_ptr());
    for (int64_t k = 0; k < n; ++k) key.push_back((int64_t)(uintptr_t)Gs[k].const_data_ptr());
    for (int64_t k = 0; k < n; ++k) key.push_back(gl[k].defined() ? (int64_t)(uintptr_t)gl[k].const_data_ptr() : 0);
    // held by value: the plan's lock is dropped around the statistics exchange below, and the pool may move on meanwhile
    const Tensor table = grp.pool->get(key, [&](void* dst) {
      auto* d = static_cast<mhaq_wlayer_desc*>(dst);
      for (int64_t k = 0; k < n; ++k) {
        const int64_t i = grp.first + k;
        d[k] = mhaq_wlayer_desc{fptr(ws[k]), nullptr, fptr(Gs[k]), fptr_or_null(gl[k]), p->co[i], p->row[i],
                                p->elem_off[i] - grp.elem0, p->chan_off[i] - grp.chan0};
      }
    }, stream);
    const auto* descs = static_cast<const mhaq_wlayer_desc*>(table.const_data_ptr());
    const float* aux = fptr(aux_all) + grp.chan0;       // the group's first channel in row 0 of [4][total_co]
    Tensor stats;
    if (grp.method == MHAQ_FQ_AEWGS && ctx->saved_data["dist"].toBool()) {
      stats = at::empty({3, grp.co}, like.options());
      check(A.mhaq_fq_wlayer_aewgs_stats_group(descs, (int)n, grp.co, aux, p->total_co, fptr_mut(stats), stream),
            "mhaq_fq_wlayer_aewgs_stats_group");
      // The exchange calls into Python (it takes the GIL on this autograd thread).  The plan_* entry points take this
      // plan's mutex WITH the GIL held: holding the mutex across the callback would invert that order and deadlock
      // against a Python thread that reads the plan's state during a distributed AEWGS backward.
      lk.unlock();
      allreduce_avg(stats);          // gdnsq.py:126-129, one message for the whole group
      lk.lock();
    }
    Tensor gw = at::empty({grp.elems}, like.options());
    Tensor gls = at::empty({grp.co}, like.options());
    const Draw d = draw_signs(false, grp.method, ctx->saved_data["rank"].toInt(), like);
    check(A.mhaq_fq_wlayer_bwd_group(descs, (int)n, grp.co, grp.max_row, aux, p->total_co, fptr_mut(gw), fptr_mut(gls),
                                     (int)grp.method, fptr_or_null(stats), d.seed, d.offset, d.offset_dev, stream),
          "mhaq_fq_wlayer_bwd_group");
    const auto ls_shapes = ctx->saved_data["ls_shapes"].to<std::vector<std::vector<int64_t>>>();
    variable_list out(2 + 2 * n + 2);
    for (int64_t k = 0; k < n; ++k) {
      const int64_t i = grp.first + k;
      out[2 + k] = at::as_strided(gw, ws[k].sizes(), ws[k].strides(), p->elem_off[i] - grp.elem0);
      out[2 + n + k] = gls.narrow(0, p->chan_off[i] - grp.chan0, p->co[i]).view(ls_shapes[k]);
    }
    return out;
  }
};

// (wq_k, lwq_k) for every layer k of group `gi`, as outputs of the group's autograd node
variable_list plan_group_apply(int64_t plan_id, int64_t gi, variable_list ws, variable_list lss, int64_t rank) {
  need_lib();
  c10::OptionalDeviceGuard device_guard;
  if (!ws.empty()) device_guard.reset_device(ws[0].device());      // see MHAQ_ON_DEVICE_OF
  variable_list all;
  all.reserve(ws.size() + lss.size());
  for (auto& t : ws) all.push_back(t);
  for (auto& t : lss) all.push_back(t);
  return WeightGroupFn::apply(plan_id, gi, at::TensorList(all), rank, g_dist_active.load());
}

// ------------------------------------------------------------------------------------------------ PotentialLoss
// gdnsq_loss.py:47-71 / 129-153 in one launch per direction.  Returns (ploss, stats[12]).
class PotentialLossFn : public torch::autograd::Function<PotentialLossFn> {
 public:
  static variable_list forward(AutogradContext* ctx, const Tensor& base_in, const Tensor& las_in, const Tensor& laq_in,
                               const Tensor& lws_in, const Tensor& lwq_in, const Tensor& state, double a_bits,
                               double w_bits, double p, bool lossless, bool update_state) {
    ctx->saved_data["shapes"] = std::vector<std::vector<int64_t>>{base_in.sizes().vec(), las_in.sizes().vec(),
                                                                  laq_in.sizes().vec(), lws_in.sizes().vec(),
                                                                  lwq_in.sizes().vec()};
    Tensor base = base_in.reshape({1}).contiguous();
    Tensor las = las_in.reshape({-1}).contiguous(), laq = laq_in.reshape({-1}).contiguous();
    Tensor lws = lws_in.reshape({-1}).contiguous(), lwq = lwq_in.reshape({-1}).contiguous();
    TORCH_CHECK(las.numel() == laq.numel() && lws.numel() == lwq.numel(),
                "potential_loss: scale and range vectors must pair up");
    Tensor out = at::empty({12}, base.options());
    check(A.mhaq_fq_potential_loss_fwd(fptr(base), fptr(las), fptr(laq), las.numel(), fptr(lws), fptr(lwq), lws.numel(),
                                       (float)a_bits, (float)w_bits, (float)p, lossless ? 1 : 0, fptr_mut(state),
                                       update_state ? 1 : 0, fptr_mut(out), cur_stream(base)),
          "mhaq_fq_potential_loss_fwd");
    ctx->save_for_backward({out, las, laq, lws, lwq});
    ctx->saved_data["a"] = a_bits;
    ctx->saved_data["w"] = w_bits;
    ctx->saved_data["p"] = p;
    ctx->mark_non_differentiable({out});
    return {out.select(0, 0).clone(), out};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto saved = ctx->get_saved_variables();
    const Tensor &out = saved[0], &las = saved[1], &laq = saved[2], &lws = saved[3], &lwq = saved[4];
    Tensor g = grads[0].reshape({1}).contiguous();
    const int64_t na = las.numel(), nw = lws.numel();
    Tensor slab = at::empty({1 + 2 * na + 2 * nw}, out.options());
    Tensor g_base = slab.narrow(0, 0, 1), g_las = slab.narrow(0, 1, na), g_laq = slab.narrow(0, 1 + na, na),
           g_lws = slab.narrow(0, 1 + 2 * na, nw), g_lwq = slab.narrow(0, 1 + 2 * na + nw, nw);
    check(A.mhaq_fq_potential_loss_bwd(fptr(g), fptr(out), fptr(las), fptr(laq), na, fptr(lws), fptr(lwq), nw,
                                       (float)ctx->saved_data["a"].toDouble(), (float)ctx->saved_data["w"].toDouble(),
                                       (float)ctx->saved_data["p"].toDouble(), fptr_mut(g_base), fptr_mut(g_las),
                                       fptr_mut(g_laq), fptr_mut(g_lws), fptr_mut(g_lwq), cur_stream(out)),
          "mhaq_fq_potential_loss_bwd");
    const auto shapes = ctx->saved_data["shapes"].to<std::vector<std::vector<int64_t>>>();
    variable_list res(11);
    res[0] = g_base.reshape(shapes[0]);
    res[1] = g_las.reshape(shapes[1]);
    res[2] = g_laq.reshape(shapes[2]);
    res[3] = g_lws.reshape(shapes[3]);
    res[4] = g_lwq.reshape(shapes[4]);
    return res;
  }
};

std::pair<Tensor, Tensor> potential_loss(const Tensor& base, const Tensor& las, const Tensor& laq, const Tensor& lws,
                                         const Tensor& lwq, const Tensor& state, double a_bits, double w_bits, double p,
                                         bool lossless, bool update_state) {
  need_lib();
  MHAQ_ON_DEVICE_OF(base);
  for (const Tensor* t : {&base, &las, &laq, &lws, &lwq})
    TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kFloat, "potential_loss: inputs must be float32 device tensors");
  TORCH_CHECK(state.is_cuda() && state.scalar_type() == at::kFloat && state.numel() == 3 && state.is_contiguous(),
              "potential_loss: state must be a contiguous float32 device tensor {loss_sum, cnt, t}");
  auto out = PotentialLossFn::apply(base, las, laq, lws, lwq, state, a_bits, w_bits, p, lossless, update_state);
  return {out[0], out[1]};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "compiled autograd binding of the mhaq_amd fused layer ops over the C ABI of include/mhaq_fq.h";
  // the torch this extension was COMPILED against (torch/version.h): mhaq_amd/_ext.py compares it with the running torch
  // before anything is bound -- the only check a stampless prebuilt extension can still be given
  m.attr("TORCH_VERSION") = TORCH_VERSION;
  static PyObject* err_class = nullptr;   // mhaq_amd._lib.MhaqFqError, installed by bind(); never released (see above)
  py::register_exception_translator([](std::exception_ptr p) {
    try {
      if (p) std::rethrow_exception(p);
    } catch (const MhaqError& e) {
      PyErr_SetString(err_class ? err_class : PyExc_RuntimeError, e.what());
    }
  });
  m.def("bind", [](const std::string& path, py::object error_class) {
    if (!err_class) err_class = error_class.inc_ref().ptr();
    bind_library(path);
  }, "dlopen the C-ABI library and resolve the entry points the nodes call");
  m.def("host_timers", [](bool reset) {
    py::dict d;
    for (int i = 0; i < T_N; ++i) {
      const int64_t n = g_tcnt[i].load(), ns = g_tns[i].load();
      d[kTimerNames[i]] = py::make_tuple(n, n ? (double)ns / (double)n * 1e-3 : 0.0);     // (calls, mean us)
      if (reset) { g_tcnt[i].store(0); g_tns[i].store(0); }
    }
    return d;
  }, py::arg("reset") = true);
  m.def("bound_library", []() { return A.path; });

  // sign streams
  m.def("rng_manual_seed", [](uint64_t seed) { std::lock_guard<std::mutex> g(R.mu); R.seed = seed; R.seeded = true; R.count = 0; });
  m.def("rng_seed", []() -> py::object { std::lock_guard<std::mutex> g(R.mu); return R.seeded ? py::object(py::int_(R.seed)) : py::none(); });
  m.def("rng_next", [](int64_t rank) {
    std::lock_guard<std::mutex> g(R.mu);
    if (!R.seeded) throw MhaqError("sign stream used before it was seeded");
    return std::make_pair(R.seed ^ ((uint64_t)rank * kGolden), ++R.count);
  });
  m.def("rng_drawn", []() { std::lock_guard<std::mutex> g(R.mu); return R.count; });
  m.def("rng_set_drawn", [](uint64_t n) { std::lock_guard<std::mutex> g(R.mu); R.count = n; });
  m.def("rng_set_base", [](const std::optional<Tensor>& base) {
    std::lock_guard<std::mutex> g(R.mu);
    R.base = (base.has_value() && base->defined()) ? *base : Tensor();
  });
  m.def("rng_base", []() -> std::optional<Tensor> {
    std::lock_guard<std::mutex> g(R.mu);
    return R.base.defined() ? std::optional<Tensor>(R.base) : std::nullopt;
  });

  // data-parallel hooks
  m.def("set_allreduce_avg", [](py::object fn) {
    g_allreduce_avg = fn.is_none() ? nullptr : fn.inc_ref().ptr();
  });
  m.def("set_dist_active", [](bool v) { g_dist_active.store(v); });

  // activation hub
  m.def("hub_create", &hub_create);
  m.def("hub_destroy", &hub_destroy);
  m.def("hub_begin", &hub_begin);
  m.def("hub_clear_pending", [](int64_t id) { auto h = hub_get(id); std::lock_guard<std::mutex> g(h->mu); h->pending.clear(); });
  m.def("hub_release_captured", [](int64_t id) { hub_get(id)->release_captured(); });
  m.def("hub_state", [](int64_t id) {
    auto h = hub_get(id);
    std::lock_guard<std::mutex> g(h->mu);
    py::dict d;
    size_t captured = 0;
    for (auto& kv : h->tables) captured += kv.second.captured;
    d["tables"] = h->tables.size();
    d["captured_tables"] = captured;
    d["retired"] = h->retired.size();
    d["pending"] = h->pending.size();
    d["has_table"] = h->last_table.defined();
    int64_t bytes = 0;
    for (auto& w : h->ws) if (w.defined()) bytes += w.numel();
    for (auto& w : h->retired) bytes += w.numel();
    d["workspace_bytes"] = bytes;
    return d;
  });
  m.def("act_layer", &act_layer, py::arg("x"), py::arg("log_act_s"), py::arg("log_act_q"), py::arg("act_b"),
        py::arg("method"), py::arg("r_sign") = py::none(), py::arg("hub") = 0, py::arg("slot") = 0, py::arg("rank") = 0);

  // weight layers
  m.def("weight_layer", &weight_layer, py::arg("w"), py::arg("log_wght_s"), py::arg("method"),
        py::arg("r_sign") = py::none(), py::arg("zp_grad") = false, py::arg("pre") = py::none(), py::arg("rank") = 0);
  m.def("weight_layer_pt", &weight_layer_pt, py::arg("w"), py::arg("log_wght_s"), py::arg("method"),
        py::arg("r_sign") = py::none(), py::arg("rank") = 0);
  m.def("weight_layer_ptl", &weight_layer_ptl, py::arg("w"), py::arg("log_wght_s"), py::arg("method"),
        py::arg("r_sign") = py::none(), py::arg("rank") = 0);
  m.def("plan_create", &plan_create);
  m.def("plan_destroy", &plan_destroy);
  m.def("plan_forward", &plan_forward);
  m.def("plan_group_apply", &plan_group_apply);
  m.def("plan_release_captured", [](int64_t id) {
    auto p = plan_get(id);
    std::lock_guard<std::mutex> g(p->mu);
    for (auto& grp : p->groups) if (grp.pool) grp.pool->release_captured();
    for (auto it = p->fwd_tables.begin(); it != p->fwd_tables.end();) it = it->second.captured ? p->fwd_tables.erase(it) : std::next(it);
  });
  m.def("plan_state", [](int64_t id) {
    auto p = plan_get(id);
    std::lock_guard<std::mutex> g(p->mu);
    py::dict d;
    d["fwd_tables"] = p->fwd_tables.size();
    py::list pools;
    for (auto& grp : p->groups) {
      py::dict e;
      size_t used = 0, held = 0, size = 0;
      if (grp.pool) {
        size = grp.pool->keys.size();
        for (size_t i = 0; i < size; ++i) { used += grp.pool->used[i]; held += grp.pool->held[i]; }
      }
      e["size"] = size; e["used"] = used; e["held"] = held;
      pools.append(e);
    }
    d["pools"] = pools;
    return d;
  });

  m.def("potential_loss", &potential_loss);
}
