// Experiment (tools only): does a 2-read 1-write stream (what pt_bwd_kernel moves) go faster when the two reads are
// LDS-DMA loads (global_load_lds_dwordx4: no VGPR destination) instead of global_load_dwordx4 into registers?
// MI355X_MICROARCH.md quotes 6.4-6.8 TB/s for an LDS-DMA read stream against ~6.0-6.3 for register loads.
//   hipcc -O3 --offload-arch=gfx950 tools/glds_triad.hip -o tools/glds_triad && ./tools/glds_triad
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float vf4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int U>
__global__ __launch_bounds__(256) void triad_reg(const vf4* __restrict__ a, const vf4* __restrict__ b, vf4* __restrict__ out, int64_t nvec) {
  const int64_t i = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
  vf4 va[U], vb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) if (i + u * 256 < nvec) { va[u] = __builtin_nontemporal_load(&a[i + u * 256]); vb[u] = __builtin_nontemporal_load(&b[i + u * 256]); }
#pragma unroll
  for (int u = 0; u < U; ++u) if (i + u * 256 < nvec) __builtin_nontemporal_store(va[u] + vb[u], &out[i + u * 256]);
}

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

// AUX: 0 = default cache policy, 2 = nt
template <int U, int AUX>
__global__ __launch_bounds__(256) void triad_glds(const vf4* __restrict__ a, const vf4* __restrict__ b, vf4* __restrict__ out, int64_t nvec) {
  __shared__ vf4 sa[U * 256], sb[U * 256];
  const int64_t i = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
  const int wave = threadIdx.x >> 6;
  const bool full = ((int64_t)blockIdx.x + 1) * 256 * U <= nvec;
  if (full) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // LDS destination = wave-uniform base + lane * 16: the wave's 64 float4 land in slots [u*256 + wave*64, +64)
      __builtin_amdgcn_global_load_lds((gptr_t)(&a[i + u * 256]), (lptr_t)(&sa[u * 256 + wave * 64]), 16, 0, AUX);
      __builtin_amdgcn_global_load_lds((gptr_t)(&b[i + u * 256]), (lptr_t)(&sb[u * 256 + wave * 64]), 16, 0, AUX);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const vf4 x = sa[u * 256 + threadIdx.x], y = sb[u * 256 + threadIdx.x];   // each lane reads what its own wave loaded
      __builtin_nontemporal_store(x + y, &out[i + u * 256]);
    }
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < nvec) out[i + u * 256] = a[i + u * 256] + b[i + u * 256];
  }
}

template <class F>
static float bench(const char* name, double bytes, int reps, F f) {
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
  std::sort(ts.begin(), ts.end());
  printf("%-36s %8.2f us (min %8.2f)  %8.1f GB/s\n", name, ts[3] * 1e3, ts[0] * 1e3, bytes / ts[3] / 1e6);
  return ts[3];
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 250LL * 64 * 56 * 56;
  const int reps = 30;
  // enough rotated buffer sets that the working set is far beyond the 256 MiB Infinity Cache
  const int NB = argc > 2 ? atoi(argv[2]) : std::max(3, (int)(1.8e9 / (12.0 * n)) + 1);
  std::vector<float*> x(NB), g(NB), y(NB);
  float* yr;
  std::vector<float> h(n);
  for (int i = 0; i < NB; ++i) {
    CK(hipMalloc(&x[i], n * 4)); CK(hipMalloc(&g[i], n * 4)); CK(hipMalloc(&y[i], n * 4));
    for (int64_t j = 0; j < n; ++j) h[j] = (float)((j * 2654435761u + i * 977) % 100003) / 25000.f - 2.f;
    CK(hipMemcpy(x[i], h.data(), n * 4, hipMemcpyHostToDevice));
    for (int64_t j = 0; j < n; ++j) h[j] = (float)((j * 40503u + i * 31) % 65521) / 32760.f - 1.f;
    CK(hipMemcpy(g[i], h.data(), n * 4, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&yr, n * 4));
  const int64_t nvec = n / 4, full = (nvec + 255) / 256;
  // correctness of the LDS-DMA form against the register form
  hipLaunchKernelGGL((triad_reg<2>), dim3((unsigned)((full + 1) / 2)), dim3(256), 0, 0, (const vf4*)x[0], (const vf4*)g[0], (vf4*)yr, nvec);
  hipLaunchKernelGGL((triad_glds<2, 2>), dim3((unsigned)((full + 1) / 2)), dim3(256), 0, 0, (const vf4*)x[0], (const vf4*)g[0], (vf4*)y[0], nvec);
  CK(hipDeviceSynchronize());
  std::vector<float> r0(n), r1(n);
  CK(hipMemcpy(r0.data(), yr, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r1.data(), y[0], n * 4, hipMemcpyDeviceToHost));
  int64_t bad = 0;
  for (int64_t j = 0; j < (nvec << 2); ++j) bad += r0[j] != r1[j];
  printf("n = %lld, %d rotated buffer sets, LDS-DMA triad vs register triad: %lld mismatching elements\n", (long long)n, NB, (long long)bad);
  for (int round = 0; round < 2; ++round) {
#define RUN(NAME, K, U) bench(NAME, 12.0 * n, reps, [&](int i) { hipLaunchKernelGGL(K, dim3((unsigned)((full + U - 1) / U)), dim3(256), 0, 0, (const vf4*)x[i % NB], (const vf4*)g[i % NB], (vf4*)y[i % NB], nvec); });
    RUN("triad registers U1 nt", (triad_reg<1>), 1)
    RUN("triad registers U2 nt", (triad_reg<2>), 2)
    RUN("triad LDS-DMA U1 default", (triad_glds<1, 0>), 1)
    RUN("triad LDS-DMA U1 nt", (triad_glds<1, 2>), 1)
    RUN("triad LDS-DMA U2 default", (triad_glds<2, 0>), 2)
    RUN("triad LDS-DMA U2 nt", (triad_glds<2, 2>), 2)
    RUN("triad LDS-DMA U4 nt", (triad_glds<4, 2>), 4)
  }
  return bad ? 1 : 0;
}
