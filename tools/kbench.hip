// Standalone HIP micro-benchmark used while tuning the streaming kernels: times copy-kernel
// variants (the achievable HBM ceiling on this box) and the library's pt_fwd / pt_bwd entry
// points with hipEvents, rotating buffers to defeat the 256 MiB Infinity Cache.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/kbench.hip -o tools/kbench \
//         -Lmhaq_amd/csrc -lmhaq_fq -Wl,-rpath,'$ORIGIN/../mhaq_amd/csrc'
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#include "../include/mhaq_fq.h"

typedef float vf4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copy_stride(const vf4* __restrict__ in, vf4* __restrict__ out, int64_t nvec) {
  int64_t i = (int64_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256 * UNROLL;
  for (; i < nvec; i += stride) {
    vf4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (i + u * 256 < nvec) v[u] = NT ? __builtin_nontemporal_load(&in[i + u * 256]) : in[i + u * 256];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (i + u * 256 < nvec) { if (NT) __builtin_nontemporal_store(v[u], &out[i + u * 256]); else out[i + u * 256] = v[u]; }
  }
}


#include "../mhaq_amd/csrc/fq_common.hpp"
using mhaq::quant_core; using mhaq::dequant; using mhaq::QCore;

// ---- experiment: forward math on two work mappings ---------------------------------
template <bool NT> __device__ inline vf4 xld(const vf4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ inline void xst(vf4* p, vf4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <int MATH>
__device__ inline vf4 fmath(vf4 a, float s, float zp, float lo, float hi) {
  if (MATH == 0) return a;
  vf4 o;
  o.x = dequant(quant_core(a.x, s, zp, lo, hi).q, s, zp);
  o.y = dequant(quant_core(a.y, s, zp, lo, hi).q, s, zp);
  o.z = dequant(quant_core(a.z, s, zp, lo, hi).q, s, zp);
  o.w = dequant(quant_core(a.w, s, zp, lo, hi).q, s, zp);
  return o;
}

// block-contiguous: each block iteration covers 256*U consecutive float4
template <int U, bool NT, int MATH>
__global__ __launch_bounds__(256) void fwd_blockmap(const vf4* __restrict__ in, vf4* __restrict__ out, int64_t nvec, const float* p) {
  const float s = p[0], zp = p[1], lo = p[2], hi = p[3];
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  for (int64_t i = (int64_t)blockIdx.x * 256 * U + threadIdx.x; i < nvec; i += stride) {
    vf4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < nvec) v[u] = xld<NT>(&in[i + u * 256]);
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < nvec) xst<NT>(&out[i + u * 256], fmath<MATH>(v[u], s, zp, lo, hi));
  }
}


// 2-read 1-write ceiling with the single-pass block-contiguous mapping (what pt_bwd moves)
template <int U, bool NT>
__global__ __launch_bounds__(256) void triad_blockmap(const vf4* __restrict__ a, const vf4* __restrict__ b, vf4* __restrict__ out, int64_t nvec) {
  const int64_t i = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
  vf4 va[U], vb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) if (i + u * 256 < nvec) { va[u] = xld<NT>(&a[i + u * 256]); vb[u] = xld<NT>(&b[i + u * 256]); }
#pragma unroll
  for (int u = 0; u < U; ++u) if (i + u * 256 < nvec) xst<NT>(&out[i + u * 256], va[u] + vb[u]);
}

// the same triad with an XCD-chunked work mapping: workgroups are dealt round-robin to the 8 XCDs, so
// block b -> chunk (b % 8) makes every XCD stream its own contiguous eighth of the tensor (24 DRAM streams
// instead of 3).  Measured: no faster than the plain mapping -- a pure stream has no L2 reuse to localise.
template <int U, bool NT>
__global__ __launch_bounds__(256) void triad_xcdmap(const vf4* __restrict__ a, const vf4* __restrict__ b, vf4* __restrict__ out, int64_t nvec) {
  const int64_t per = ((int64_t)gridDim.x + 7) / 8;
  const int64_t blk = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int64_t i = blk * 256 * U + threadIdx.x;
  vf4 va[U], vb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) if (i + u * 256 < nvec) { va[u] = xld<NT>(&a[i + u * 256]); vb[u] = xld<NT>(&b[i + u * 256]); }
#pragma unroll
  for (int u = 0; u < U; ++u) if (i + u * 256 < nvec) xst<NT>(&out[i + u * 256], va[u] + vb[u]);
}

// read-only (sum) and write-only kernels to see each direction's ceiling
__global__ __launch_bounds__(256) void read_only(const float4* __restrict__ in, float* __restrict__ out, int64_t nvec) {
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    float4 v = in[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void write_only(float4* __restrict__ out, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256)
    out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

struct Timer {
  std::vector<hipEvent_t> a, b;
  Timer(int n) : a(n), b(n) { for (int i = 0; i < n; ++i) { CK(hipEventCreate(&a[i])); CK(hipEventCreate(&b[i])); } }
};

template <class F>
static float bench(const char* name, double bytes, int reps, F&& f) {
  for (int i = 0; i < 3; ++i) f(i);
  CK(hipDeviceSynchronize());
  Timer t(reps);
  for (int i = 0; i < reps; ++i) { CK(hipEventRecord(t.a[i], 0)); f(i); CK(hipEventRecord(t.b[i], 0)); }
  CK(hipDeviceSynchronize());
  std::vector<float> ms(reps);
  for (int i = 0; i < reps; ++i) CK(hipEventElapsedTime(&ms[i], t.a[i], t.b[i]));
  std::sort(ms.begin(), ms.end());
  float med = ms[reps / 2];
  printf("%-34s %8.4f ms (min %7.4f)  %8.1f GB/s\n", name, med, ms[0], bytes / med / 1e6);
  return med;
}

int main(int argc, char** argv) {
  int64_t n = argc > 1 ? atoll(argv[1]) : 250LL * 64 * 56 * 56;
  int reps = argc > 2 ? atoi(argv[2]) : 20;
  const int NB = 3;
  float *x[NB], *g[NB], *y[NB];
  for (int i = 0; i < NB; ++i) {
    CK(hipMalloc(&x[i], n * 4)); CK(hipMalloc(&g[i], n * 4)); CK(hipMalloc(&y[i], n * 4));
    std::vector<float> h(n);
    for (int64_t j = 0; j < n; ++j) h[j] = (float)((j * 2654435761u + i * 977) % 100003) / 25000.f - 2.f;
    CK(hipMemcpy(x[i], h.data(), n * 4, hipMemcpyHostToDevice));
    for (int64_t j = 0; j < n; ++j) h[j] = (float)((j * 40503u + i * 31) % 65521) / 32760.f - 1.f;
    CK(hipMemcpy(g[i], h.data(), n * 4, hipMemcpyHostToDevice));
  }
  float hp[4] = {0.2371f, -1.9f, -1.9f, -1.9f + 16 * 0.2371f - 0.2371f};
  float* p; CK(hipMalloc(&p, 16)); CK(hipMemcpy(p, hp, 16, hipMemcpyHostToDevice));
  float* grads; CK(hipMalloc(&grads, 32));
  size_t wsb = mhaq_fq_pt_bwd_workspace_bytes(n);
  void* ws; CK(hipMalloc(&ws, wsb));
  const int64_t nvec = n / 4;
  printf("n = %lld elements (%.1f MB per tensor)\n", (long long)n, n * 4 / 1e6);

  const bool lib_only = argc > 3;
  if (!lib_only) {
  for (int grid : {1024, 2048, 4096, 8192, 16384}) {
    char nm[64];
    snprintf(nm, 64, "copy u1 grid %d", grid);
    bench(nm, 8.0 * n, reps, [&](int i) { hipLaunchKernelGGL((copy_stride<1, false>), dim3(grid), dim3(256), 0, 0, (const vf4*)x[i % NB], (vf4*)y[i % NB], nvec); });
  }
  for (int grid : {512, 1024, 2048, 4096}) {
    char nm[64];
    snprintf(nm, 64, "copy u4 grid %d", grid);
    bench(nm, 8.0 * n, reps, [&](int i) { hipLaunchKernelGGL((copy_stride<4, false>), dim3(grid), dim3(256), 0, 0, (const vf4*)x[i % NB], (vf4*)y[i % NB], nvec); });
    snprintf(nm, 64, "copy u4 nt grid %d", grid);
    bench(nm, 8.0 * n, reps, [&](int i) { hipLaunchKernelGGL((copy_stride<4, true>), dim3(grid), dim3(256), 0, 0, (const vf4*)x[i % NB], (vf4*)y[i % NB], nvec); });
  }
  {
    int64_t full = (nvec + 255) / 256;
    bench("copy u1 one-vec-per-thread", 8.0 * n, reps, [&](int i) { hipLaunchKernelGGL((copy_stride<1, false>), dim3((unsigned)full), dim3(256), 0, 0, (const vf4*)x[i % NB], (vf4*)y[i % NB], nvec); });
    bench("copy u4 4-vec-per-thread", 8.0 * n, reps, [&](int i) { hipLaunchKernelGGL((copy_stride<4, false>), dim3((unsigned)((full + 3) / 4)), dim3(256), 0, 0, (const vf4*)x[i % NB], (vf4*)y[i % NB], nvec); });
    bench("copy u4 nt 4-vec-per-thread", 8.0 * n, reps, [&](int i) { hipLaunchKernelGGL((copy_stride<4, true>), dim3((unsigned)((full + 3) / 4)), dim3(256), 0, 0, (const vf4*)x[i % NB], (vf4*)y[i % NB], nvec); });
  }
  bench("read only grid 2048", 4.0 * n, reps, [&](int i) { hipLaunchKernelGGL(read_only, dim3(2048), dim3(256), 0, 0, (const float4*)x[i % NB], grads, nvec); });
  bench("read only grid 8192", 4.0 * n, reps, [&](int i) { hipLaunchKernelGGL(read_only, dim3(8192), dim3(256), 0, 0, (const float4*)x[i % NB], grads, nvec); });
  bench("write only grid 2048", 4.0 * n, reps, [&](int i) { hipLaunchKernelGGL(write_only, dim3(2048), dim3(256), 0, 0, (float4*)y[i % NB], nvec); });
  bench("hipMemcpyAsync d2d", 8.0 * n, reps, [&](int i) { CK(hipMemcpyAsync(y[i % NB], x[i % NB], n * 4, hipMemcpyDeviceToDevice, 0)); });
  }


  {
    int64_t full = (nvec + 255) / 256;
#define RUNB(U, NT, MATH, GRID, LABEL) { char nm[96]; snprintf(nm, 96, "blockmap U%d nt%d math%d grid %s", U, NT, MATH, LABEL); \
    bench(nm, 8.0 * n, reps, [&](int i) { hipLaunchKernelGGL((fwd_blockmap<U, NT, MATH>), dim3((unsigned)(GRID)), dim3(256), 0, 0, (const vf4*)x[i % NB], (vf4*)y[i % NB], nvec, p); }); }
    RUNB(4, true, 0, (full + 3) / 4, "all")
    RUNB(4, true, 1, (full + 3) / 4, "all")
    RUNB(2, true, 1, (full + 1) / 2, "all")
    RUNB(1, true, 1, full, "all")
    RUNB(8, true, 1, (full + 7) / 8, "all")
    RUNB(4, true, 1, 2048, "2048")
    RUNB(4, true, 1, 4096, "4096")
    RUNB(4, true, 1, 1024, "1024")
    RUNB(8, true, 1, 2048, "2048")
    RUNB(8, true, 1, 1024, "1024")
    RUNB(2, true, 1, 4096, "4096")
    RUNB(2, true, 1, 8192, "8192")
    RUNB(4, false, 1, (full + 3) / 4, "all")
  }


  {
    int64_t full = (nvec + 255) / 256;
#define RUNT(U, NT) { char nm[96]; snprintf(nm, 96, "triad 2R1W U%d nt%d grid all", U, NT); \
    bench(nm, 12.0 * n, reps, [&](int i) { hipLaunchKernelGGL((triad_blockmap<U, NT>), dim3((unsigned)((full + U - 1) / U)), dim3(256), 0, 0, (const vf4*)x[i % NB], (const vf4*)g[i % NB], (vf4*)y[i % NB], nvec); }); }
    RUNT(1, true) RUNT(2, true) RUNT(4, true) RUNT(6, true) RUNT(4, false)
#define RUNX(U, NT) { char nm[96]; snprintf(nm, 96, "triad 2R1W U%d nt%d xcd-chunked", U, NT); \
    const unsigned gx = (unsigned)((((full + U - 1) / U) + 7) / 8 * 8); \
    bench(nm, 12.0 * n, reps, [&](int i) { hipLaunchKernelGGL((triad_xcdmap<U, NT>), dim3(gx), dim3(256), 0, 0, (const vf4*)x[i % NB], (const vf4*)g[i % NB], (vf4*)y[i % NB], nvec); }); }
    RUNX(1, true) RUNX(2, true)
  }

  float tf = bench("mhaq_fq_pt_fwd", 8.0 * n, reps, [&](int i) { mhaq_fq_pt_fwd(x[i % NB], y[i % NB], n, p, p + 1, p + 2, p + 3, nullptr, nullptr, nullptr, nullptr, 0, nullptr); });
  float tb = bench("mhaq_fq_pt_bwd STE (+finalize)", 12.0 * n, reps, [&](int i) { mhaq_fq_pt_bwd(x[i % NB], g[i % NB], y[i % NB], n, p, p + 1, p + 2, p + 3, MHAQ_FQ_STE, nullptr, 0, nullptr, 99, i + 1, nullptr, 0, grads, ws, wsb, nullptr); });
  float tl = bench("mhaq_fq_pt_bwd LSQ (+finalize)", 12.0 * n, reps, [&](int i) { mhaq_fq_pt_bwd(x[i % NB], g[i % NB], y[i % NB], n, p, p + 1, p + 2, p + 3, MHAQ_FQ_LSQ, nullptr, 0, nullptr, 99, i + 1, nullptr, 0, grads, ws, wsb, nullptr); });
  printf("fused fwd+bwd STE: %.1f GB/s   LSQ: %.1f GB/s (20 B/elem)\n", 20.0 * n / (tf + tb) / 1e6, 20.0 * n / (tf + tl) / 1e6);
  return 0;
}
