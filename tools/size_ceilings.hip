// Per-size ceilings for the streaming kernels: at every activation size of the BASELINE configs,
// the bare 1R1W stream (what pt_fwd_kernel moves) and the bare 2R1W stream (what pt_bwd_kernel moves) next to the
// library's own launches, timed the same way on the same box, with the non-temporal and the default load policy, and
// with the second input either cold (written long ago: rotated buffer sets far beyond the 256 MB Infinity Cache) or
// FRESH (written by the kernel right before, as dgrad writes `g` in a training step).
//   hipcc -O3 --offload-arch=gfx950 tools/size_ceilings.hip -o tools/size_ceilings \
//         -Lmhaq_amd/csrc -lmhaq_fq -Wl,-rpath,'$ORIGIN/../mhaq_amd/csrc'
//   ./tools/size_ceilings [lib]          ("lib": library launches only -- for A/B runs over tools/variants/*)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../include/mhaq_fq.h"

typedef float vf4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT> __device__ inline vf4 xld(const vf4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }

template <int U, bool NTL>
__global__ __launch_bounds__(256) void copy_k(const vf4* __restrict__ a, vf4* __restrict__ out, int64_t nvec) {
  const int64_t i = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
  vf4 va[U];
#pragma unroll
  for (int u = 0; u < U; ++u) if (i + u * 256 < nvec) va[u] = xld<NTL>(&a[i + u * 256]);
#pragma unroll
  for (int u = 0; u < U; ++u) if (i + u * 256 < nvec) __builtin_nontemporal_store(va[u], &out[i + u * 256]);
}

// NTA / NTB: load policy of the first (x: saved by the forward, cold) and the second (g: dgrad's output) stream
template <int U, bool NTA, bool NTB>
__global__ __launch_bounds__(256) void triad_k(const vf4* __restrict__ a, const vf4* __restrict__ b, vf4* __restrict__ out, int64_t nvec) {
  const int64_t i = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
  vf4 va[U], vb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) if (i + u * 256 < nvec) { va[u] = xld<NTA>(&a[i + u * 256]); vb[u] = xld<NTB>(&b[i + u * 256]); }
#pragma unroll
  for (int u = 0; u < U; ++u) if (i + u * 256 < nvec) __builtin_nontemporal_store(va[u] + vb[u], &out[i + u * 256]);
}

// the producer of the FRESH legs: writes `g` with the default (cache-allocating) store policy, like a convolution's dgrad
__global__ __launch_bounds__(256) void produce_k(vf4* __restrict__ out, int64_t nvec, float v) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < nvec) { vf4 o = {v, -v, 0.5f * v, 0.25f * v}; out[i] = o; }
}

// back-to-back launches between two events: the throughput a sequence of such launches sees (kernel boundaries included)
template <class F>
static float b2b(int reps, F f) {
  for (int i = 0; i < 5; ++i) f(i);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int r = 0; r < 7; ++r) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f(r * reps + i);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ts.push_back(ms / reps);
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  std::sort(ts.begin(), ts.end());
  return ts[3] * 1e3f;   // us
}

// [producer(i); consumer(i)] pairs between two events minus the producers alone: what the consumer costs behind a
// kernel that has just written its input
template <class P, class F>
static float fresh(int reps, P p, F f) {
  const float both = b2b(reps, [&](int i) { p(i); f(i); });
  const float prod = b2b(reps, [&](int i) { p(i); });
  return both - prod;
}

int main(int argc, char** argv) {
  const bool lib_only = argc > 1 && !strcmp(argv[1], "lib");
  // ResNet-18 b250: 50.2 / 25.1 / 12.5 / 6.3 M; ResNet-20 b1000 (configs[1]): 16.4 / 8.2 / 4.1 M; RFDN reference shape: 0.69 M
  std::vector<int64_t> sizes = {50176000, 25088000, 16384000, 12544000, 8192000, 6272000, 4096000, 691200};
  // MHAQ_EXTRA_SIZES="10985472,4718592,1548288": more element counts (multiples of 4), e.g. the four launch sizes of ResNet-18's
  // model-wide weight launches: what bare streams of exactly those sizes cost
  if (const char* extra = getenv("MHAQ_EXTRA_SIZES")) {
    sizes.clear();
    for (const char* p = extra; *p;) {
      char* end;
      const long long v = strtoll(p, &end, 10);
      if (end == p) break;
      if (v >= 1024 && v % 4 == 0) sizes.push_back(v);
      p = (*end == ',') ? end + 1 : end;
    }
  }
  float hp[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  float *ls, *lq, *bb, *params;
  CK(hipMalloc(&ls, 4)); CK(hipMalloc(&lq, 4)); CK(hipMalloc(&bb, 4)); CK(hipMalloc(&params, 20));
  hp[0] = -2.0764f; CK(hipMemcpy(ls, hp, 4, hipMemcpyHostToDevice));     // s = 0.2371
  hp[0] = 1.9236f; CK(hipMemcpy(lq, hp, 4, hipMemcpyHostToDevice));      // 16 levels
  hp[0] = -1.9f; CK(hipMemcpy(bb, hp, 4, hipMemcpyHostToDevice));
  printf("# us per launch (GB/s algorithmic: 8 B/elem 1R1W, 12 B/elem 2R1W); b2b = back-to-back launches over rotated cold buffers,\n");
  printf("# fresh = behind a kernel that has just written the second input (g) with cache-allocating stores\n");
  // MHAQ_SIZES="691200,4096000,6272000": only these sizes (a rocprofv3 pass over the small tensors)
  const char* only = getenv("MHAQ_SIZES");
  for (int64_t n : sizes) {
    if (only) {
      char key[32];
      snprintf(key, sizeof key, "%lld", (long long)n);
      if (!strstr(only, key)) continue;
    }
    const int NB = (int)std::min<int64_t>(64, std::max<int64_t>(3, (int64_t)(1.8e9 / (12.0 * n)) + 1));
    std::vector<float*> x(NB), g(NB), y(NB);
    std::vector<float> h(n);
    for (int i = 0; i < NB; ++i) {
      CK(hipMalloc(&x[i], n * 4)); CK(hipMalloc(&g[i], n * 4)); CK(hipMalloc(&y[i], n * 4));
      if (i < 2) {
        for (int64_t j = 0; j < n; ++j) h[j] = (float)((j * 2654435761u + i * 977) % 100003) / 25000.f - 2.f;
        CK(hipMemcpy(x[i], h.data(), n * 4, hipMemcpyHostToDevice));
        for (int64_t j = 0; j < n; ++j) h[j] = (float)((j * 40503u + i * 31) % 65521) / 32760.f - 1.f;
        CK(hipMemcpy(g[i], h.data(), n * 4, hipMemcpyHostToDevice));
      } else {
        CK(hipMemcpy(x[i], x[i & 1], n * 4, hipMemcpyDeviceToDevice));
        CK(hipMemcpy(g[i], g[i & 1], n * 4, hipMemcpyDeviceToDevice));
      }
    }
    const size_t wsb = mhaq_fq_act_bwd_workspace_bytes(n);
    void* ws; CK(hipMalloc(&ws, wsb));
    const int64_t nvec = n / 4, full = (nvec + 255) / 256;
    const int reps = n > 20000000 ? 20 : 40;
    printf("n = %9lld (%6.1f MB/tensor, %2d buffer sets)\n", (long long)n, n * 4 / 1e6, NB);
    auto line = [&](const char* name, double bytes, float cold, float fr) {
      if (fr >= 0) printf("  %-44s b2b %7.2f us %7.0f GB/s | fresh %7.2f us %7.0f GB/s\n", name, cold, bytes / cold / 1e3, fr, bytes / fr / 1e3);
      else printf("  %-44s b2b %7.2f us %7.0f GB/s\n", name, cold, bytes / cold / 1e3);
    };
    auto prod = [&](int i) { hipLaunchKernelGGL(produce_k, dim3((unsigned)full), dim3(256), 0, 0, (vf4*)g[i % NB], nvec, 1.f + (i & 7)); };
    auto prodx = [&](int i) { hipLaunchKernelGGL(produce_k, dim3((unsigned)full), dim3(256), 0, 0, (vf4*)x[i % NB], nvec, 1.f + (i & 7)); };
    if (!lib_only) {
#define COPY(U, NTL) [&](int i) { hipLaunchKernelGGL((copy_k<U, NTL>), dim3((unsigned)((full + U - 1) / U)), dim3(256), 0, 0, (const vf4*)x[i % NB], (vf4*)y[i % NB], nvec); }
#define TRIAD(U, NA, NB_) [&](int i) { hipLaunchKernelGGL((triad_k<U, NA, NB_>), dim3((unsigned)((full + U - 1) / U)), dim3(256), 0, 0, (const vf4*)x[i % NB], (const vf4*)g[i % NB], (vf4*)y[i % NB], nvec); }
      line("copy 1R1W U1 nt loads", 8.0 * n, b2b(reps, COPY(1, true)), fresh(reps, prodx, COPY(1, true)));
      line("copy 1R1W U1 default loads", 8.0 * n, b2b(reps, COPY(1, false)), fresh(reps, prodx, COPY(1, false)));
      line("triad 2R1W U2 nt x, nt g", 12.0 * n, b2b(reps, TRIAD(2, true, true)), fresh(reps, prod, TRIAD(2, true, true)));
      line("triad 2R1W U2 nt x, default g", 12.0 * n, b2b(reps, TRIAD(2, true, false)), fresh(reps, prod, TRIAD(2, true, false)));
      line("triad 2R1W U2 default x, default g", 12.0 * n, b2b(reps, TRIAD(2, false, false)), fresh(reps, prod, TRIAD(2, false, false)));
    }
    auto fwd = [&](int i) { mhaq_fq_act_fwd(x[i % NB], y[i % NB], n, ls, lq, bb, params, nullptr, nullptr, nullptr, 0, nullptr); };
    fwd(0);
    int32_t np = 0;
    auto bwd_ste = [&](int i) { mhaq_fq_act_bwd_partials(x[i % NB], g[i % NB], y[i % NB], n, params, MHAQ_FQ_STE, nullptr, 99, (uint64_t)i + 1, nullptr, ws, wsb, &np, nullptr); };
    auto bwd_lsq = [&](int i) { mhaq_fq_act_bwd_partials(x[i % NB], g[i % NB], y[i % NB], n, params, MHAQ_FQ_LSQ, nullptr, 99, (uint64_t)i + 1, nullptr, ws, wsb, &np, nullptr); };
    line("mhaq_fq_act_fwd", 8.0 * n, b2b(reps, fwd), fresh(reps, prodx, fwd));
    line("mhaq_fq_act_bwd_partials STE", 12.0 * n, b2b(reps, bwd_ste), fresh(reps, prod, bwd_ste));
    line("mhaq_fq_act_bwd_partials LSQ", 12.0 * n, b2b(reps, bwd_lsq), fresh(reps, prod, bwd_lsq));
    fflush(stdout);
    for (int i = 0; i < NB; ++i) { CK(hipFree(x[i])); CK(hipFree(g[i])); CK(hipFree(y[i])); }
    CK(hipFree(ws));
  }
  return 0;
}
