// Per-channel weight fake-quant kernels for gfx950: W viewed as [co][row], one scale per
// output channel, zero point = row minimum (gdnsq_conv2d.py:71-98).  One workgroup per
// channel; the row (and, in backward, the upstream-gradient row) is staged in LDS so that
// the row minimum, the quantizer, the per-channel reductions and the amin tie-split
// scatter all run off a single HBM read.  Reductions: registers -> wave shuffle (fp64)
// -> LDS -> thread 0; deterministic, no atomics.
#include "fq_common.hpp"
#include <atomic>

namespace mhaq {

// LDS staging budget per workgroup.  gfx950 has 160 KiB of LDS per CU; rows (forward) or row pairs (backward)
// up to 144 KiB are staged, the rest of the CU's LDS holds the reduction scratch.  Launches that ask for more
// than 64 KiB opt in through hipFuncSetAttribute and run one 1024-thread workgroup per CU.
constexpr int64_t kMaxStageFloats = 36 * 1024;
constexpr size_t kDefaultDynLds = 64 * 1024;
constexpr int kMaxWaves = 16;                   // reduction scratch is sized for up to 1024 threads

// -DMHAQ_TRACE (tools/variants.sh trace "-DMHAQ_TRACE"; never in the shipped library): the first wave of every workgroup
// of the multi-tensor launches stamps the 100 MHz wall clock at its phase boundaries -- 0 entry, 1 descriptor read,
// 2 row in registers, 3 row reduction done, 4 last store issued, 5 stores acknowledged; slot 6 = HW_ID | XCC_ID << 32 --
// into a device buffer read back by mhaq_debug_trace_read (tools/pc_multi_bench.py, MHAQ_PCMB_TRACE=1).
#ifdef MHAQ_TRACE
#ifndef MHAQ_TRACE_NOWAIT
#define MHAQ_TRACE_NOWAIT 0      // 1: stamp without draining the memory counters (the shipped kernel's own overlap of loads and arithmetic)
#endif
constexpr int kTraceBlocks = 8192;
__device__ unsigned long long mhaq_trace_buf[8 * kTraceBlocks];
#define MHAQ_TRACE_AT(k, WAIT)                                                                       \
  do {                                                                                               \
    if (threadIdx.x == 0 && blockIdx.x < kTraceBlocks) {                                             \
      if ((WAIT) && !MHAQ_TRACE_NOWAIT) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  \
      mhaq_trace_buf[blockIdx.x * 8 + (k)] = wall_clock64();                                         \
      if ((k) == 0)                                                                                  \
        mhaq_trace_buf[blockIdx.x * 8 + 6] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) |  \
                                             ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32); \
    }                                                                                                \
  } while (0)
#else
#define MHAQ_TRACE_AT(k, WAIT) do { } while (0)
#endif
// A/B knob (tools/variants.sh): the SECOND half of a model-wide grid starts its rows NS nanoseconds late.  The dispatcher deals
// workgroups round robin over XCDs and CUs, so the second half of the grid is the later half of EVERY CU's resident
// workgroups: the first half gets the memory system to itself and is computing by the time the second half's rows arrive.
// Measured (profiles/r06_pc_multi_stagger.txt): 1.5-3.5 us never pays for itself -- backward groups 13.0 / 12.9 / 7.3 -> 13.2 / 13.4 / 8.0 us
// at 1.5 us, forward 17.3 -> 17.6; a 3 us delay of half the grid costs the 19 MB groups only 0.3 us (the overlap is real, the gain is not).  0 = off.
#ifndef MHAQ_FWD_STAGGER_NS
#define MHAQ_FWD_STAGGER_NS 0
#endif
#ifndef MHAQ_BWD_STAGGER_NS
#define MHAQ_BWD_STAGGER_NS 0
#endif
template <int NS>
__device__ __forceinline__ void stagger_second_half() {
  if constexpr (NS > 0) {
    if (2 * blockIdx.x >= gridDim.x) {
      const unsigned long long t0 = wall_clock64();            // 100 MHz
      while (wall_clock64() - t0 < (unsigned long long)(NS / 10)) __builtin_amdgcn_s_sleep(16);
    }
  }
}
// Sign tile of one row (sign stream v3, fq_common.hpp): the Philox calls covering the row's elements, computed once by the
// workgroup.  296 calls = 37,888 elements: every row the staged / register-resident kernels take (<= 36 K floats) at any
// alignment of its first element inside a call; longer rows draw call by call (philox_nibble / philox_r).
constexpr int kRowTileCalls = 296;
struct RowSigns {
  int64_t rel0;      // stream index of the row's first element, relative to the tile's first element
  bool tiled;
};
static_assert(kMaxStageFloats / 128 + 2 <= kRowTileCalls, "the tile holds every staged / register-resident row");
// Fills `tile` for the row whose first element has stream index e0; the caller places a barrier before the first read.
// FITS = the launcher only sends rows of <= kMaxStageFloats here (staged and register-resident bodies): `tiled` is then a
// compile-time fact and the call-by-call fallback (a whole Philox call, and its registers, inside the element loop) drops
// out of those kernels.
template <bool FITS>
__device__ __forceinline__ RowSigns row_signs_begin(uint32_t* __restrict__ tile, int64_t e0, int64_t row, uint64_t seed,
                                                    uint64_t offset) {
  const int64_t c0 = e0 >> kSignsPerCallLog2;
  const int64_t ncalls = ((e0 + row - 1) >> kSignsPerCallLog2) - c0 + 1;
  RowSigns rs{e0 - (c0 << kSignsPerCallLog2), FITS || ncalls <= kRowTileCalls};
  if (rs.tiled) sign_tile_fill(tile, c0, (int)(FITS && ncalls > kRowTileCalls ? kRowTileCalls : ncalls), seed, offset);
  return rs;
}

// Static LDS of one backward workgroup: ONE instance per kernel, handed to whichever row body the workgroup runs.  (A
// __shared__ array declared inside a body is allocated once per body inlined into the kernel: the multi-tensor kernels,
// which carry the register-resident and both staged bodies, held three sign tiles -- 15 KB of static LDS, which capped them
// at 3-4 workgroups per CU where their registers allow 6-7; tools/pc_multi_bench.py MHAQ_PCMB_TRACE=1.)
template <bool PHILOX>
struct BwdLds {
  alignas(16) uint32_t stile[PHILOX ? 4 * kRowTileCalls : 4];      // sign_tile_fill stores 16 bytes per call (ds_write_b128)
  double sm[3 * kMaxWaves];
  double sm4[2 * kMaxWaves];
  int smt[2 * kMaxWaves];
};
static_assert(alignof(BwdLds<true>) >= 16 && offsetof(BwdLds<true>, stile) % 16 == 0, "the sign tile takes 16-byte LDS stores");
template <int METHOD, bool RSIGN>
using BwdLdsOf = BwdLds<(METHOD != MHAQ_FQ_LSQ) && !RSIGN>;

// float4 path: rows are a whole number of float4 and every row start is 16-byte aligned
__host__ __device__ inline bool vec_ok(int64_t row, const void* a, const void* b, const void* c = nullptr) {
  return ((row & 3) == 0) && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0);
}

__device__ inline float block_bcast(float v, float* slot) {
  __syncthreads();
  if (threadIdx.x == 0) *slot = v;
  __syncthreads();
  return *slot;
}

// NaN-propagating block min (torch.amin semantics); result broadcast to all threads.
__device__ inline float block_min_bcast(float mn, bool nan, float* sm /* [nw+1] */) {
  if (nan) mn = NAN;
  auto nmin = [](float a, float b) { return (a != a || b != b) ? NAN : fminf(a, b); };
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mn = nmin(mn, __shfl_down(mn, o, 64));
  const int nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mn;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < nw; ++w) mn = nmin(mn, sm[w]);
    sm[nw] = mn;
  }
  __syncthreads();
  return sm[nw];
}

// ------------------------------------------------------------------ forward
// LAYER = the whole NoisyConv2d weight forward of one layer in one launch: the scale comes from the
// learnable log2-scale (s = exp2(log_wght_s), gdnsq_conv2d.py:72) and the row maximum gives the
// regulariser input log2(max - min + s) of ModelHelper.get_model_values (model_helper.py:24-44),
// which otherwise costs a second amin/amax sweep over every weight per step.
#define MHAQ_LN2F 0.69314718055994531f
// VEC: row % 4 == 0 and 16-byte aligned tensors -> every global and LDS access moves a float4.
template <bool STAGE, bool WRITE_Q, bool LAYER, bool VEC>
__device__ __forceinline__ void pc_fwd_body(const float* __restrict__ w, float* __restrict__ wq,
                                            float* __restrict__ zp_out, float* __restrict__ q_out,
                                            const float* __restrict__ s, int64_t row,
                                            float* __restrict__ s_out, float* __restrict__ mx_out,
                                            float* __restrict__ lwq_out, const int64_t c) {
  extern __shared__ __align__(16) float smem[];
  __shared__ float red[2 * kMaxWaves];
  constexpr int W = VEC ? 4 : 1;
  const int64_t step = (int64_t)blockDim.x * W;
  const float* wrow = w + c * row;
  float mn = INFINITY, mx = -INFINITY;
  bool nan = false;
#pragma unroll 2
  for (int64_t j = (int64_t)threadIdx.x * W; j < row; j += step) {
    float v[W];
    ldvg<W>(wrow + j, v);
    if (STAGE) stv<W>(smem + j, v);
#pragma unroll
    for (int k = 0; k < W; ++k) {
      mn = fminf(mn, v[k]);
      mx = fmaxf(mx, v[k]);
      nan |= (v[k] != v[k]);
    }
  }
  block_minmax_all(mn, mx, nan, red);                   // one barrier; also orders the LDS staging above
  const float zp = mn;
  float sc;
  if (LAYER) {
    const float rmx = mx;
    sc = exp2f(ldg(s + c));                             // s holds log_wght_s here
    if (threadIdx.x == 0) {
      stg(s_out + c, sc);
      stg(mx_out + c, rmx);
      stg(lwq_out + c, log2f((rmx - zp) + sc));
    }
  } else {
    sc = ldg(s + c);
  }
  if (threadIdx.x == 0) stg(zp_out + c, zp);
  // the row's scale is uniform: the exact-quotient form of the backward pass (quant_core_w: 5 VALU instructions for the
  // IEEE division's 11, no clamp against +-inf) gives the forward the same q -- and wq = q * s + zp -- bit for bit
  const BwdCtx kx = make_bwd_ctx(sc, zp, -INFINITY, INFINITY);
  for (int64_t j = (int64_t)threadIdx.x * W; j < row; j += step) {
    float v[W], o[W], qv[W];
    if (STAGE) ldv<W>(smem + j, v);
    else ldvg<W>(wrow + j, v);
#pragma unroll
    for (int k = 0; k < W; ++k) {
      QCore q = quant_core_w(v[k], kx);
      o[k] = dequant(q.q, sc, zp);
      qv[k] = q.q;
    }
    stvg<W>(wq + c * row + j, o);
    if (WRITE_Q) stvg<W>(q_out + c * row + j, qv);
  }
}

template <bool STAGE, bool WRITE_Q, bool LAYER, bool VEC>
__global__ void pc_fwd_kernel(const float* __restrict__ w, float* __restrict__ wq, float* __restrict__ zp_out,
                              float* __restrict__ q_out, const float* __restrict__ s, int64_t row,
                              float* __restrict__ s_out, float* __restrict__ mx_out,
                              float* __restrict__ lwq_out) {
  pc_fwd_body<STAGE, WRITE_Q, LAYER, VEC>(w, wq, zp_out, q_out, s, row, s_out, mx_out, lwq_out, blockIdx.x);
}

// Multi-tensor launch: every per-channel weight layer of a model in ONE grid (SURVEY.md 8b "multi-tensor
// variants taking a device pointer table").  Block b serves channel b - first_block of the layer whose
// [first_block, first_block + co) range contains it; outputs go to two caller-owned slabs.
struct WLayerDesc {            // mirrors mhaq_wlayer_desc in include/mhaq_fq.h
  const float* w;              // [co][row]
  const float* log_s;          // [co]
  const float* G;              // backward only: dL/dwq [co][row]
  const float* g_lwq;          // backward only, nullable: dL/dlwq [co]
  int64_t co, row;
  int64_t elem_offset;         // offset of this layer in the wq / gw slab (elements)
  int64_t chan_offset;         // offset of this layer in the per-channel slabs == first block
};

// The channel a workgroup of a multi-tensor grid serves: workgroup b serves channel b.  (Measured and not adopted, round 4:
// the LAST channel first -- a CNN's rows grow with depth, so ascending order leaves the longest rows for the tail of a
// 20 us launch --: slower, ResNet-18 forward 23.5 -> 26.8 us cold, gpurun_out/r04e_pc_multi.txt: with every CU starting on
// 4608-float rows at once the workgroups run their load and compute phases in lock step.  With the row loads no longer
// serialized (second half of the round) the order stopped mattering: last-first 21.7 vs 21.9 us; both ends towards the middle
// (=2) puts all long rows on four of the eight XCDs, which the dispatcher deals workgroups to round robin -- those finish at
// 19.5 us, the others at 14.5 --; in runs of 8 workgroups (=3) -0.35 us forward, +0.7 us backward.  -DMHAQ_MULTI_REVERSE=1/2/3
// keep them as A/B knobs for tools/variants.sh.)
#ifndef MHAQ_PACKED
#define MHAQ_PACKED 1         // A/B knob: 0 = the scalar per-element code in the AEWGS backward
#endif
#ifndef MHAQ_PACKED_STE
#define MHAQ_PACKED_STE 1     // the packed-fp32 element pair (ste_pair) in the STE / LSQ backward on rows of more than 4 float4 per thread: the
                              // compute phase of a 19 MB group is all resident waves contending for the VALUs (profiles/r06_pc_multi_stagger.txt);
                              // 15 instead of 30 fp32 operations per pair, the same bits, two VGPRs fewer.  ResNet-18 STE set 50.2 -> 49.3-49.6 us
                              // cold, 43.9 -> 43.1-43.6 warm, groups 12.15 / 13.11 -> 11.85 / 12.85; LSQ 50.3 -> 49.4-50.0; nothing slower
                              // (profiles/r06_pc_multi_packed_ste.txt).  0 = the scalar element (A/B).
#endif
#ifndef MHAQ_PACKED_FWD
#define MHAQ_PACKED_FWD 1         // the packed-fp32 pair in the register-resident forward (quant_core_w + dequant, the same bits): with streaming
                                  // stores the model-wide forward's compute + store phase is what the pair shortens -- 17.5 -> 17.1 us cold, 14.45 -> 14.0
                                  // warm (round 4, with the output's write-back behind it, measured no gain); 0 = the scalar element (A/B)
#endif
#ifndef MHAQ_PACKED_AEWGS_MULTI
#define MHAQ_PACKED_AEWGS_MULTI 1   // the packed AEWGS element at <= 4 float4 per thread inside the model-wide launches (0: A/B)
#endif
#ifndef MHAQ_PACKED_STE_ALL
#define MHAQ_PACKED_STE_ALL 1 // the packed pair at every row length (unlike AEWGS, whose kept quotients compete for the registers at <= 4 float4 per
                              // thread, the STE / LSQ pair needs none more: 60 VGPRs instead of 63): the 6 MB group 7.4 -> 7.1 us cold, 6.9-7.2 -> 6.55 warm,
                              // the ResNet-18 set 43.3 -> 42.4 warm, [4096,4096] backward 37.0 -> 36.4; 0 = only above 4 float4 per thread (A/B)
#endif
#ifndef MHAQ_MULTI_REVERSE
#define MHAQ_MULTI_REVERSE 0
#endif
#ifndef MHAQ_MULTI_REG
#define MHAQ_MULTI_REG 1      // A/B knob: 0 = the LDS-staged bodies for every row of the multi-tensor launches
#endif
__device__ __forceinline__ int64_t multi_channel() {
  const int64_t b = blockIdx.x, n = gridDim.x;
  if (MHAQ_MULTI_REVERSE == 2) return (b & 1) ? (b >> 1) : n - 1 - (b >> 1);      // both ends towards the middle
  if (MHAQ_MULTI_REVERSE == 3) {      // ... in runs of 8 workgroups (one per XCD: the dispatcher deals them round robin)
    const int64_t g = b >> 3, i = ((g >> 1) << 3) + (b & 7);
    const int64_t c = (g & 1) ? i : n - 1 - i;
    return (c >= 0 && c < n && (n & 15) == 0) ? c : b;      // (grids that are not whole pairs of runs keep the identity)
  }
  return MHAQ_MULTI_REVERSE ? n - 1 - b : b;
}

// The layer whose [chan_offset, chan_offset + co) range holds channel b: binary search over the descriptor table (the
// offsets ascend).  Wave-uniform scalar loads, each a dependent round trip in front of the workgroup's first data load:
// 4 of them for ResNet-18's 16 layers where the linear scan of rounds 2-3 took up to 15 -- and the LAST layers, which
// scanned longest, hold most of a CNN's weights.
// Up to 64 layers (every model of the reference): no search at all -- lane l reads layer l's offset, all of them in ONE
// round trip, and the answer is the highest lane whose offset is <= b (a ballot; lanes beyond n repeat the last layer, which
// cannot change it).  The phase stamps of a -DMHAQ_TRACE build put the search at 1.6 us per workgroup for 16 layers (five
// dependent scalar loads + the descriptor) in front of a row whose loads take 2.  (Sixteen independent SCALAR loads issued back
// to back and a count, for tables of <= 16 layers: slower -- descriptor phase 0.7 -> 2.0 us median, forward 20.7 -> 21.4 us;
// gpurun_out/r04f_trace_scalar.txt.)
#ifndef MHAQ_FIND_BALLOT
#define MHAQ_FIND_BALLOT 1
#endif
__device__ __forceinline__ int find_layer(const WLayerDesc* __restrict__ d, int n, int64_t b) {
  if (MHAQ_FIND_BALLOT && n <= 64) {
    const int lane = threadIdx.x & 63;
    const int64_t off = gptr(d)[lane < n ? lane : n - 1].chan_offset;
    const unsigned long long m = __ballot(off <= b);
    const int top = m ? 63 - __builtin_clzll(m) : 0;
    return top < n ? top : n - 1;
  }
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (b >= d[mid].chan_offset) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <bool STAGE>
__global__ void pc_fwd_multi_kernel(const WLayerDesc* __restrict__ descs, int nlayers, float* __restrict__ wq_all,
                                    float* __restrict__ aux_all /* [4][total_co]: s, zp, mx, lwq */,
                                    int64_t total_co) {
  const WLayerDesc d = descs[find_layer(descs, nlayers, multi_channel())];
  float* a = aux_all + d.chan_offset;
  float* wq = wq_all + d.elem_offset;
  const int64_t c = multi_channel() - d.chan_offset;
  // a table that does not cover the grid must not turn into a stray access (d.w in the test: the descriptor's two halves
  // are then read back to back, not the pointers after the branch on the offsets)
  if ((c < 0) | (c >= d.co) | (d.w == nullptr)) return;
  if (vec_ok(d.row, d.w, wq))     // per layer, workgroup-uniform
    pc_fwd_body<STAGE, false, true, true>(d.w, wq, a + total_co, nullptr, d.log_s, d.row, a, a + 2 * total_co,
                                          a + 3 * total_co, c);
  else
    pc_fwd_body<STAGE, false, true, false>(d.w, wq, a + total_co, nullptr, d.log_s, d.row, a, a + 2 * total_co,
                                           a + 3 * total_co, c);
}

// ------------------------------------------------------------------ AEWGS statistics
__device__ inline void pc_stats_accumulate_div(float w, float g, float sc, float zp, double (&st)[3]) {
  const QCore q = quant_core(w, sc, zp, -INFINITY, INFINITY);     // per-element scales (quantized bias): plain division
  const float gq = g * sc;
  st[0] += (double)(sign_f(gq) * q.n);
  st[1] += (double)(q.n * q.n);
  st[2] += (double)q.n;
}
__device__ inline void pc_stats_accumulate(float w, float g, float sc, const BwdCtx& kx, double (&st)[3]) {
  const QCore q = quant_core_w(w, kx);     // the forward's bits (fq_common.hpp) at 7 VALU instructions instead of ~20
  const float gq = g * sc;
  st[0] += (double)(sign_f(gq) * q.n);
  st[1] += (double)(q.n * q.n);
  st[2] += (double)q.n;
}

template <bool VEC>
__global__ void pc_aewgs_stats_kernel(const float* __restrict__ w, const float* __restrict__ G,
                                      const float* __restrict__ s, const float* __restrict__ zp, int64_t co,
                                      int64_t row, float* __restrict__ stats) {
  __shared__ double sm[3 * 4];
  constexpr int W = VEC ? 4 : 1;
  const int64_t c = blockIdx.x;
  const float sc = s[c], z = zp[c];
  const BwdCtx kx = make_bwd_ctx(sc, z, -INFINITY, INFINITY);
  double st[3] = {0, 0, 0};
#pragma unroll 2
  for (int64_t j = (int64_t)threadIdx.x * W; j < row; j += (int64_t)blockDim.x * W) {
    float x[W], g[W];
    ldv<W>(w + c * row + j, x);
    ldv<W>(G + c * row + j, g);
#pragma unroll
    for (int k = 0; k < W; ++k) pc_stats_accumulate(x[k], g[k], sc, kx, st);
  }
  block_sum<3>(st, sm);
  if (threadIdx.x == 0) {
    const float inv = (float)row;
    stats[c] = (float)st[0] / inv;
    stats[co + c] = (float)st[1] / inv;
    stats[2 * co + c] = (float)st[2] / inv;
  }
}

// Per-row min / max (weight calibration, calib/minmaxobserver.py:73-75): one workgroup per row, read-only.
template <bool VEC>
__global__ void row_minmax_kernel(const float* __restrict__ w, int64_t row, float* __restrict__ mn_out,
                                  float* __restrict__ mx_out) {
  __shared__ float red[2 * kMaxWaves];
  constexpr int W = VEC ? 4 : 1;
  const float* wrow = w + (int64_t)blockIdx.x * row;
  float mn = INFINITY, mx = -INFINITY;
  bool nan = false;
#pragma unroll 2
  for (int64_t j = (int64_t)threadIdx.x * W; j < row; j += (int64_t)blockDim.x * W) {
    float v[W];
    ldv<W>(wrow + j, v);
#pragma unroll
    for (int k = 0; k < W; ++k) { mn = fminf(mn, v[k]); mx = fmaxf(mx, v[k]); nan |= (v[k] != v[k]); }
  }
  block_minmax_all(mn, mx, nan, red);                     // NaN-propagating like torch.amin / amax
  if (threadIdx.x == 0) { mn_out[blockIdx.x] = mn; mx_out[blockIdx.x] = mx; }
}

// ------------------------------------------------------------------ backward
template <int METHOD, bool RSIGN, bool STAGE, bool LAYER, bool VEC>
__device__ __forceinline__ void pc_bwd_body(const float* __restrict__ w, const float* __restrict__ G,
                                            float* __restrict__ gw, float* __restrict__ g_s,
                                            const float* __restrict__ s, const float* __restrict__ zp,
                                            int64_t co, int64_t row, const float* __restrict__ stats,
                                            const float* __restrict__ gzp_extra,
                                            const int8_t* __restrict__ r_sign, uint64_t seed, uint64_t offset,
                                            const float* __restrict__ mx, const float* __restrict__ g_lwq,
                                            const int64_t c, const int64_t rng_base, BwdLdsOf<METHOD, RSIGN>& lds) {
  extern __shared__ __align__(16) float smem[];
  double* const sm = lds.sm;
  constexpr int W = VEC ? 4 : 1;
  const int64_t first = (int64_t)threadIdx.x * W, step = (int64_t)blockDim.x * W;
  float* sw = smem;
  float* sg = smem + (STAGE ? row : 0);
  const float sc = ldg(s + c), z = ldg(zp + c);
  const BwdCtx kx = make_bwd_ctx(sc, z, -INFINITY, INFINITY);
  const float* wrow = w + c * row;
  const float* grow = G + c * row;
  constexpr bool PHILOX = (METHOD != MHAQ_FQ_LSQ) && !RSIGN;
  uint32_t* const stile = lds.stile;
  RowSigns rsg{0, false};
  if (PHILOX) rsg = row_signs_begin<STAGE>(stile, rng_base + c * row, row, seed, offset);
  const bool tiled = STAGE || rsg.tiled;

  if (STAGE) {
#pragma unroll 2
    for (int64_t j = first; j < row; j += step) {
      float x[W], g[W];
      ldvg<W>(wrow + j, x);
      ldvg<W>(grow + j, g);
      stv<W>(sw + j, x);
      stv<W>(sg + j, g);
    }
    // every thread only ever revisits the slots it wrote itself: no barrier needed here
  }
  if (PHILOX) __syncthreads();          // the sign tile is read across threads

  float delta = 0.f;
  if (METHOD == MHAQ_FQ_AEWGS) {
    float num, e2, me;
    if (stats) {
      num = ldg(stats + c); e2 = ldg(stats + co + c); me = ldg(stats + 2 * co + c);
    } else {
      double st[3] = {0, 0, 0};
      for (int64_t j = first; j < row; j += step) {
        float x[W], g[W];
        if (STAGE) { ldv<W>(sw + j, x); ldv<W>(sg + j, g); }
        else { ldvg<W>(wrow + j, x); ldvg<W>(grow + j, g); }
#pragma unroll
        for (int k = 0; k < W; ++k) pc_stats_accumulate(x[k], g[k], sc, kx, st);
      }
      block_sum_all<3>(st, sm);
      const float inv = (float)row;
      num = (float)st[0] / inv;
      e2 = (float)st[1] / inv;
      me = (float)st[2] / inv;
    }
    delta = aewgs_delta(num, e2, me);
  }

  // pass 1: per-channel sums; gv/s parked in LDS for pass 2
  const float rmx = LAYER ? ldg(mx + c) : 0.f;
  double acc[4] = {0, 0, 0, 0};  // d/ds, sum(G - gv/s), count(w == min), count(w == max)
  int cnt_min = 0, cnt_max = 0;  // per-lane integer tallies: exact in any order, 2 VALU instead of a 64-bit select + add
  for (int64_t j = first; j < row; j += step) {
    float xv[W], gv_[W], r[W], park[W];
    if (STAGE) { ldv<W>(sw + j, xv); ldv<W>(sg + j, gv_); }
    else { ldvg<W>(wrow + j, xv); ldvg<W>(grow + j, gv_); }
    if (METHOD != MHAQ_FQ_LSQ) {
      const int64_t i = rng_base + c * row + j;
      // r[] holds rc = r * 3^-1/2 (nibble_to_rc4): the noise term below is gq * rc, the bits of (3^-1/2 * gq) * r
      if (RSIGN) {
#pragma unroll
        for (int k = 0; k < W; ++k) r[k] = sign_to_rc(sign_half(r_sign[i + k]));
      } else if constexpr (W == 4) {
        // i % 4 == 0 on this path (launcher checks rng_base): the four signs share a word of the tile
        float r4[4];
        nibble_to_rc4(tiled ? sign_tile_nibble(stile, rsg.rel0 + j) : philox_nibble(i, seed, offset), r4);
#pragma unroll
        for (int k = 0; k < W; ++k) r[k] = r4[k];
      } else {
        r[0] = sign_to_rc(tiled ? sign_tile_r(stile, rsg.rel0 + j) : philox_r(i, seed, offset));
      }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const float x = xv[k], g = gv_[k];
      QCore q = quant_core_w(x, kx);                // same bits as the forward's IEEE division (fq_common.hpp)
      const float gq = g * sc;
      float gvs;
      if (METHOD == MHAQ_FQ_AEWGS) {
        // d/ds term g*q - gv*(v/s) with gv = gq*(1 - gsc) is g*(n + v*gsc) in real arithmetic: one product instead of
        // the difference of two ~|q| x larger ones (the form STE / LSQ use as g*(q - v), see fq_pt.hip bwd_elem)
        const float gsc = aewgs_gsc(gq, q.n, delta);
        const float gv = gq + (-gq * gsc);
        gvs = quot(gv, kx);
        acc[0] += (double)(g * (q.n + q.v * gsc) + gq * r[k]);
      } else {
        const float gv = gq + noise_grad_v<METHOD>(gq, q.n, delta);
        gvs = (METHOD == MHAQ_FQ_STE || METHOD == MHAQ_FQ_LSQ) ? quot_of_product(g, gv, kx) : quot(gv, kx);
        const float noise_s = (METHOD == MHAQ_FQ_LSQ) ? gq * q.n : gq * r[k];
        // STE/LSQ: gv == g*sc, so g*q - gv*(v/sc) == g*(q - v) exactly (see fq_pt.hip bwd_elem)
        if (METHOD == MHAQ_FQ_STE || METHOD == MHAQ_FQ_LSQ)
          acc[0] += (double)(g * q.n + noise_s);
        else
          acc[0] += (double)((g * q.q + (-gv) * quot(q.v, kx)) + noise_s);
      }
      acc[1] += (double)(g - gvs);
      if constexpr (W == 1) {
        cnt_min += (x == z) ? 1 : 0;
        if (LAYER) cnt_max += (x == rmx) ? 1 : 0;
      }
      park[k] = gvs;
    }
    if constexpr (W == 4) {       // tallies only where the float4 holds a row extreme (see pc_bwd_reg_kernel)
      const float m4 = fminf(fminf(xv[0], xv[1]), fminf(xv[2], xv[3]));
      const float M4 = fmaxf(fmaxf(xv[0], xv[1]), fmaxf(xv[2], xv[3]));
      if ((m4 == z) | (LAYER && (M4 == rmx))) {
#pragma unroll
        for (int k = 0; k < W; ++k) {
          cnt_min += (xv[k] == z) ? 1 : 0;
          if (LAYER) cnt_max += (xv[k] == rmx) ? 1 : 0;
        }
      }
    }
    if (STAGE) stv<W>(sg + j, park);
  }
  double* const sm4 = lds.sm4;
  int* const smt = lds.smt;
  {                                    // every thread holds the two row sums and the two tie counts after this one barrier
    double d2[2] = {acc[0], acc[1]}, t2[2];
    block_sum_all_tally<2>(d2, cnt_min, cnt_max, t2, sm4, smt);
    acc[0] = d2[0]; acc[1] = d2[1]; acc[2] = t2[0]; acc[3] = t2[1];
  }
  // zero-point gradient: +sum G (dequantize) - sum gv/s (before the divide) [+ grad from other users of zp]
  float gzp_local = (float)acc[1];
  if (gzp_extra) gzp_local = gzp_local + ldg(gzp_extra + c);
  float gs_local = (float)acc[0];
  float t_local = 0.f;
  if (LAYER) {
    // regulariser input lwq = log2(u), u = (max - min) + s: log2 backward g / (u * ln2) flows
    // +t to the maxima (amax backward), -t to the minima (amin backward) and +t to s
    if (g_lwq) t_local = ldg(g_lwq + c) / (((rmx - z) + sc) * MHAQ_LN2F);
    gzp_local = gzp_local - t_local;
    gs_local = gs_local + t_local;
  }
  if (threadIdx.x == 0) stg(g_s + c, LAYER ? (gs_local * sc) * MHAQ_LN2F : gs_local);   // exp2 backward when LAYER
  const float gzp = gzp_local;
  const float cnt = (float)acc[2];
  const float tie = (gzp * 1.0f) / cnt;  // amin backward: (grad * mask) / count
  float tie_max = 0.f;
  if (LAYER) tie_max = (t_local * 1.0f) / (float)acc[3];   // amax backward

  // pass 2: gW = gv/s + tie-split share of the zero-point (and range) gradient
  for (int64_t j = first; j < row; j += step) {
    float xv[W], gvs[W], o[W];
    if (STAGE) {
      ldv<W>(sw + j, xv);
      ldv<W>(sg + j, gvs);
    } else {
      float g[W];
      ldvg<W>(wrow + j, xv);
      ldvg<W>(grow + j, g);
#pragma unroll
      for (int k = 0; k < W; ++k) {
        QCore q = quant_core_w(xv[k], kx);
        const float gq = g[k] * sc;
        const float gv = (METHOD == MHAQ_FQ_AEWGS) ? gq + (-gq * aewgs_gsc(gq, q.n, delta))
                                                   : gq + noise_grad_v<METHOD>(gq, q.n, delta);
        gvs[k] = (METHOD == MHAQ_FQ_STE || METHOD == MHAQ_FQ_LSQ) ? quot_of_product(g[k], gv, kx) : quot(gv, kx);
      }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      o[k] = (xv[k] == z) ? gvs[k] + tie : gvs[k];
      if (LAYER && xv[k] == rmx) o[k] = o[k] + tie_max;
    }
    stvg<W>(gw + c * row + j, o);
  }
}

template <int METHOD, bool RSIGN, bool STAGE, bool LAYER, bool VEC>
__global__ void pc_bwd_kernel(const float* __restrict__ w, const float* __restrict__ G, float* __restrict__ gw,
                              float* __restrict__ g_s, const float* __restrict__ s, const float* __restrict__ zp,
                              int64_t co, int64_t row, const float* __restrict__ stats,
                              const float* __restrict__ gzp_extra,
                              const int8_t* __restrict__ r_sign, uint64_t seed, uint64_t offset, const uint64_t* __restrict__ offset_dev,
                              const float* __restrict__ mx, const float* __restrict__ g_lwq) {
  offset = stream_offset(offset, offset_dev);
  __shared__ BwdLdsOf<METHOD, RSIGN> lds;
  pc_bwd_body<METHOD, RSIGN, STAGE, LAYER, VEC>(w, G, gw, g_s, s, zp, co, row, stats, gzp_extra, r_sign, seed,
                                                offset, mx, g_lwq, blockIdx.x, 0, lds);
}

// Two elements per instruction: gfx950's packed fp32 operations (v_pk_add / v_pk_mul / v_pk_fma_f32: two IEEE fp32 results per
// lane and instruction, the form the chip's fp32 vector peak is quoted for) on float2 values.  Every operation below is the
// scalar code's, element by element and in the same order, so the results are its bits; v_rndne, the compares / selects and
// the fp64 row sums have no packed form.  Used where a kernel is VALU-bound: the per-channel AEWGS backward (75-87 VALU
// instructions per element before).  All of them require k.fast_div (the caller takes the scalar code otherwise).
typedef float vf2 __attribute__((ext_vector_type(2)));
// v = (x - zp) / s, the IEEE quotient (quant_core_w's corrections)
__device__ __forceinline__ vf2 exact_v2(vf2 x, float s, float rs, float zp) {
  const vf2 s2 = {s, s}, rs2 = {rs, rs};
  const vf2 v1 = x - vf2{zp, zp};
  const vf2 q0 = v1 * rs2;
  const vf2 q1 = __builtin_elementwise_fma(__builtin_elementwise_fma(-s2, q0, v1), rs2, q0);
  return __builtin_elementwise_fma(__builtin_elementwise_fma(-s2, q1, v1), rs2, q1);
}
// one element pair of the AEWGS apply pass: gv = gq - gq * gsc and the d/ds term g * (n + v * gsc) + gq * rc
struct AewgsPair { vf2 gv, term; };
__device__ __forceinline__ AewgsPair aewgs_pair(vf2 v, vf2 g, vf2 rc, float sc, float delta) {
  const vf2 rn = {rintf(v.x), rintf(v.y)};
  const vf2 n = rn - v;
  const vf2 gq = g * vf2{sc, sc};
  // aewgs_gsc: n with gq's sign bit folded in, times delta, clamped from above at 0.99 (a NaN compares false and passes)
  const vf2 nf = {__uint_as_float(__float_as_uint(n.x) ^ (__float_as_uint(gq.x) & 0x80000000u)),
                  __uint_as_float(__float_as_uint(n.y) ^ (__float_as_uint(gq.y) & 0x80000000u))};
  const vf2 t = vf2{delta, delta} * nf;
  const vf2 gsc = {(t.x > 0.99f) ? 0.99f : t.x, (t.y > 0.99f) ? 0.99f : t.y};
  AewgsPair p;
  p.gv = gq + (-gq * gsc);
  p.term = g * (n + v * gsc) + gq * rc;
  return p;
}
// one element pair of the STE / LSQ backward (the scalar code of pc_bwd_reg_body, operation by operation, two per instruction):
// v and n of quant_core_w, gv = gq + gq * 0 (noise_grad_v), gv / s as quot_of_product's single correction, the d/ds term
// g * n + noise with noise = gq * n (LSQ) or gq * rc (STE).  Only called with k.fast_div (the caller's wave-uniform test).
struct StePair { vf2 gvs, term; };
template <int METHOD>
__device__ __forceinline__ StePair ste_pair(vf2 x, vf2 g, vf2 rc, const BwdCtx& k) {
  const vf2 s2 = {k.s, k.s}, rs2 = {k.rs, k.rs};
  const vf2 v = exact_v2(x, k.s, k.rs, k.zp);
  const vf2 n = vf2{rintf(v.x), rintf(v.y)} - v;
  const vf2 gq = g * s2;
  const vf2 gv = gq + gq * vf2{0.f, 0.f};
  StePair p;
  p.gvs = __builtin_elementwise_fma(__builtin_elementwise_fma(-s2, g, gv), rs2, g);
  const vf2 noise = (METHOD == MHAQ_FQ_LSQ) ? gq * n : gq * rc;
  p.term = g * n + noise;
  return p;
}
// the four exact quotients x / s of a float4 (quot(), fq_common.hpp) behind ONE range test -- the smallest and the largest
// |x * rs| of the four: a float4 with an element outside the exact range takes quot() element by element, so the values are
// those of four quot() calls (a NaN x is dropped by min / max and runs through the fma chain: NaN either way)
__device__ __forceinline__ void exact_quot4(vf2 a, vf2 b, const BwdCtx& k, float (&out)[4]) {
  const vf2 s2 = {k.s, k.s}, rs2 = {k.rs, k.rs};
  const vf2 qa = a * rs2, qb = b * rs2;
  const float lo = fminf(fminf(fabsf(qa.x), fabsf(qa.y)), fminf(fabsf(qb.x), fabsf(qb.y)));
  const float hi = fmaxf(fmaxf(fabsf(qa.x), fabsf(qa.y)), fmaxf(fabsf(qb.x), fabsf(qb.y)));
  if (lo > 0x1p-100f && hi < 0x1p100f) {
    const vf2 a1 = __builtin_elementwise_fma(__builtin_elementwise_fma(-s2, qa, a), rs2, qa);
    const vf2 b1 = __builtin_elementwise_fma(__builtin_elementwise_fma(-s2, qb, b), rs2, qb);
    const vf2 ra = __builtin_elementwise_fma(__builtin_elementwise_fma(-s2, a1, a), rs2, a1);
    const vf2 rb = __builtin_elementwise_fma(__builtin_elementwise_fma(-s2, b1, b), rs2, b1);
    out[0] = ra.x; out[1] = ra.y; out[2] = rb.x; out[3] = rb.y;
  } else {
    out[0] = quot(a.x, k); out[1] = quot(a.y, k); out[2] = quot(b.x, k); out[3] = quot(b.y, k);
  }
}

// ------------------------------------------------------------------ register-resident rows
// Rows that are a whole number of float4 and fit T x NV float4 (T threads, NV <= 8: up to 32 K floats) never
// touch LDS with their data: thread t keeps the float4 t, t + T, ... of the row (and of the G row) in registers
// across the row reductions.  All of a row's HBM loads are issued before the first is waited for -- the same
// single-pass shape as the streaming kernels in fq_pt.hip -- and 8 workgroups of 256 threads stay resident per
// CU, where the LDS-staged form held 5 (32 KiB row pairs) down to 1 (> 64 KiB) and idled the memory pipe during
// its compute phases ([4096,4096]: 4.1 / 3.6 TB/s fwd / bwd staged; profiles/r02_pc_bench.txt for this form).
// Element -> thread mapping, per-thread accumulation order and every fp32 operation are those of the staged
// bodies above, so results are bit-identical to them.
// NT: non-temporal global accesses, chosen at launch for tensors of kPcNtBytes and more -- a stream that large
// cannot stay in the caches anyway, and as in fq_pt.hip the streaming policy is worth 4-10 % on it ([8192,8192]
// 5.35 -> 5.77 TB/s forward, 5.40 -> 5.90 backward; [50257,768] 5.46 -> 6.00 / 5.45 -> 5.87; profiles/r02_pc_bench.txt).
// The layers of a training step (<= 9.4 MB each) keep the default policy: the convolution that consumes wq and the
// optimizer that consumes gW find them in the Infinity Cache.
constexpr int64_t kPcNtBytes = 32ll << 20;
#ifndef MHAQ_PC_SMALL_NT
#define MHAQ_PC_SMALL_NT 3     // cache policy of the per-layer launches below kPcNtBytes (codes: see MHAQ_FWD_MULTI_NT): streaming
                               // STORES, like the model-wide launches and for the same reason -- [512,4608] backward 9.6-9.9 -> 8.8-9.0 us,
                               // forward 8.6-8.9 -> 8.3-8.5 (tools/pc_bench.py, profiles/r06_pc_multi_final.txt)
#endif
// (global memory by contract, whatever the pointer's origin: see gptr in fq_common.hpp)
template <bool NT>
__device__ __forceinline__ vf4 pc_ld(const vf4* p) { return NT ? __builtin_nontemporal_load(gptr(p)) : *gptr(p); }
template <bool NT>
__device__ __forceinline__ void pc_st(vf4* p, vf4 v) { if (NT) __builtin_nontemporal_store(v, gptr(p)); else *gptr(p) = v; }

template <bool WRITE_Q, bool LAYER, int NV, int NT>
__device__ __forceinline__ void pc_fwd_reg_body(
    const float* __restrict__ w, float* __restrict__ wq, float* __restrict__ zp_out, float* __restrict__ q_out,
    const float* __restrict__ s, int64_t row, float* __restrict__ s_out, float* __restrict__ mx_out,
    float* __restrict__ lwq_out, const int64_t c) {
  __shared__ float red[2 * kMaxWaves];
  const int items = (int)(row >> 2), T = blockDim.x;
  const vf4* wrow = reinterpret_cast<const vf4*>(w + c * row);
  vf4 v[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    // unconditional, index clamped into the row (items >= 1): a load under `if (j < items)` ends in a register copy at the
    // join, and the copy in an s_waitcnt per load -- the row's loads would go out one round trip after the other
    const int j = threadIdx.x + k * T;
    v[k] = pc_ld<(NT == 1 || NT == 2)>(wrow + (j < items ? j : items - 1));
  }
  // the channel's (log-)scale goes out under the row loads: behind the barrier of the row reduction its round trip
  // would sit on the workgroup's critical path
  const float s_c = ldg(s + c);
  // ... and so does everything that depends on the scale alone (exp2, the reciprocal and the range test of the exact-quotient
  // context): computed while the row is in flight, not between the row reduction and the first store
  const float sc = LAYER ? exp2f(s_c) : s_c;             // s holds log_wght_s when LAYER
  BwdCtx kx = make_bwd_ctx(sc, 0.f, -INFINITY, INFINITY);      // exact quotients without the division: see pc_fwd_body
  MHAQ_TRACE_AT(2, true);
  float mn = INFINITY, mx = -INFINITY;
  bool nan = false;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if ((int)threadIdx.x + k * T < items) {
      const float e[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mn = fminf(mn, e[q]);
        mx = fmaxf(mx, e[q]);
        nan |= (e[q] != e[q]);
      }
    }
  }
  block_minmax_all(mn, mx, nan, red);
  MHAQ_TRACE_AT(3, false);
  const float zp = mn;
  kx.zp = zp;
  vf4* orow = reinterpret_cast<vf4*>(wq + c * row);
  vf4* qrow = WRITE_Q ? reinterpret_cast<vf4*>(q_out + c * row) : nullptr;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int j = threadIdx.x + k * T;
    if (j < items) {
      const float e[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
      float o[4], qv[4];
      if (MHAQ_PACKED_FWD && kx.fast_div) {       // two elements per instruction (quant_core_w + dequant, operation by operation)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const vf2 vv = exact_v2(vf2{e[2 * h], e[2 * h + 1]}, sc, kx.rs, zp);
          const vf2 n = vf2{rintf(vv.x), rintf(vv.y)} - vv;
          const vf2 q2 = vv + n;
          const vf2 o2 = q2 * vf2{sc, sc} + vf2{zp, zp};
          qv[2 * h] = q2.x; qv[2 * h + 1] = q2.y;
          o[2 * h] = o2.x; o[2 * h + 1] = o2.y;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          QCore qc = quant_core_w(e[q], kx);
          o[q] = dequant(qc.q, sc, zp);
          qv[q] = qc.q;
        }
      }
      pc_st<(NT == 1 || NT == 3)>(orow + j, vf4{o[0], o[1], o[2], o[3]});
      if (WRITE_Q) pc_st<(NT == 1 || NT == 3)>(qrow + j, vf4{qv[0], qv[1], qv[2], qv[3]});
    }
  }
  // the channel's scalars last: thread 0's log2 does not hold back its wave's share of the row
  if (threadIdx.x == 0) {
    stg(zp_out + c, zp);
    if (LAYER) {
      stg(s_out + c, sc);
      stg(mx_out + c, mx);
      stg(lwq_out + c, log2f((mx - zp) + sc));
    }
  }
  MHAQ_TRACE_AT(4, false);
  MHAQ_TRACE_AT(5, true);
}

template <bool WRITE_Q, bool LAYER, int NV, int NT>
__global__ __launch_bounds__(64 * kMaxWaves) void pc_fwd_reg_kernel(
    const float* __restrict__ w, float* __restrict__ wq, float* __restrict__ zp_out, float* __restrict__ q_out,
    const float* __restrict__ s, int64_t row, float* __restrict__ s_out, float* __restrict__ mx_out,
    float* __restrict__ lwq_out) {
  pc_fwd_reg_body<WRITE_Q, LAYER, NV, NT>(w, wq, zp_out, q_out, s, row, s_out, mx_out, lwq_out, blockIdx.x);
}

// The model-wide forward launch with register-resident rows: every row of a training step's layers that is a whole
// number of aligned float4 and fits NV float4 per thread (ResNet-18: all 3840 rows, <= 4608 floats, NV = 5 at 256
// threads) takes the single-pass body above -- all of the row's loads in flight before the first wait, no LDS round
// trip --, any other row of the same grid the staged body (workgroup-uniform choice per layer).  The staged multi
// kernel was latency-bound: waves waiting 68 % of their cycles, VALUs active 36 % (profiles/r04_pc_multi_pmc.txt);
// same-box A/B against it (tools/variants.sh staged, gpurun_out/r04f_pc_multi_ab.txt): ResNet-18 forward 25.4 -> 24.2 us
// cold, 27.7 -> 20.9 in the training step; STE backward groups 21.9 / 21.1 -> 19.4 / 19.4 cold.
// TB = threads per workgroup: 256, or 1024 for models whose rows are whole tensors (multi_threads(): PER_TENSOR layers riding
// the launch as one channel each, e.g. ResNet-20 with `qscheme: 0`, rows up to 36,864 floats = NV 9 at 1024 threads).
// Cache policy of the model-wide launches' row accesses: 0 default, 1 non-temporal, 2 non-temporal LOADS only, 3 non-temporal
// STORES only.  Round 6: 3 for both directions.  Stores that allocate in L2 leave the launch's whole output dirty there --
// 19 MB for a backward group of ResNet-18, 44 MB for the model-wide forward against 32 MB of L2 -- and the end-of-kernel
// write-back drains it AFTER the last workgroup has retired: the phase stamps (profiles/r06_pc_multi_trace.txt) end
// 2.4-2.8 us before the launch does on these launches and 0.5 us before it on the 6 MB group.  Streaming stores drain while
// the rows are still being computed.  Loads keep the default policy: with streaming stores it beats streaming loads in both
// directions (rounds 4-5 measured the load policy with allocating stores only, where 2 won for the backward).  Same box,
// tools/pc_multi_bench.py STE resnet18, cold / the same 44 MB every launch (profiles/r06_pc_multi_nt.txt):
//   forward            policy 0 (r5)  20.8 / 15.4 us    1  17.6 / 16.5    3  17.7 / 14.2
//   backward groups    policy 2 (r5)  13.3 13.5 7.7 / 11.9 11.8 7.1    1  12.8 13.4 7.5 / 11.3 12.6 7.1    3  12.2 12.3 7.3 / 10.7 10.7 6.5
//   forward + grouped backward    r5  55.3 / 46.2 us  ->  49.5 / 42.1
// (wq and gW are re-read within the step by the convolutions / the optimizer out of the memory-side Infinity Cache, which
// the L2 policy does not bypass.)  Its 8 waves per SIMD are all needed (a cap at 6 / 4 / 2 waves: 21.7 -> 22.2 / 23.1 /
// 28.5 us cold): the launch is bound by the latency of a row's round trips, not by HBM.
// (A/B knobs for tools/variants.sh.)
#ifndef MHAQ_BWD_MULTI_NT
#define MHAQ_BWD_MULTI_NT 3
#endif
#ifndef MHAQ_FWD_MULTI_NT
#define MHAQ_FWD_MULTI_NT 3
#endif
#ifndef MHAQ_FWD_MULTI_TB128
#define MHAQ_FWD_MULTI_TB128 1
#endif
#ifndef MHAQ_FWD_MULTI_MAXW
#define MHAQ_FWD_MULTI_MAXW 8
#endif
template <int NV, int TB>
__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(((TB == 128 || TB == kBlock) ? (MHAQ_FWD_MULTI_MAXW < 8 ? 1 : 8) : 1), ((TB == 128 || TB == kBlock) ? MHAQ_FWD_MULTI_MAXW : 8))))
void pc_fwd_multi_reg_kernel(
    const WLayerDesc* __restrict__ descs, int nlayers, float* __restrict__ wq_all, float* __restrict__ aux_all,
    int64_t total_co) {
  MHAQ_TRACE_AT(0, false);
  stagger_second_half<MHAQ_FWD_STAGGER_NS>();
  const WLayerDesc d = descs[find_layer(descs, nlayers, multi_channel())];
  float* a = aux_all + d.chan_offset;
  float* wq = wq_all + d.elem_offset;
  const int64_t c = multi_channel() - d.chan_offset;
  // a table that does not cover the grid must not turn into a stray access (d.w in the test: the descriptor's two halves
  // are then read back to back, not the pointers after the branch on the offsets)
  if ((c < 0) | (c >= d.co) | (d.w == nullptr)) return;
  const bool vec = vec_ok(d.row, d.w, wq);
  MHAQ_TRACE_AT(1, true);
  // (Measured and not adopted, round 4: a launch is compiled for its LONGEST row, and a shorter row issues the surplus
  // loads of that body with a clamped index.  Sending rows of <= 2 / <= 4 float4 per thread to the NV = 2 / 4 bodies inside
  // the same kernel, and a forward NV = 5: ResNet-18 forward + grouped backward 58.8 -> 59.3 us, gpurun_out/r04d_pc_multi.txt.)
  if (vec && (d.row >> 2) <= (int64_t)NV * TB)
    pc_fwd_reg_body<false, true, NV, MHAQ_FWD_MULTI_NT>(d.w, wq, a + total_co, nullptr, d.log_s, d.row, a, a + 2 * total_co,
                                                        a + 3 * total_co, c);
  else if (vec)
    pc_fwd_body<false, false, true, true>(d.w, wq, a + total_co, nullptr, d.log_s, d.row, a, a + 2 * total_co,
                                          a + 3 * total_co, c);
  else
    pc_fwd_body<false, false, true, false>(d.w, wq, a + total_co, nullptr, d.log_s, d.row, a, a + 2 * total_co,
                                           a + 3 * total_co, c);
}

// `offset` is the effective stream offset (the caller has added *offset_dev); rng_base = stream index of the tensor's
// first element (0 for a single layer, the layer's element offset inside a multi-tensor launch).
template <int METHOD, bool RSIGN, bool LAYER, int NV, int NT, bool MULTI = false>
__device__ __forceinline__ void pc_bwd_reg_body(
    const float* __restrict__ w, const float* __restrict__ G, float* __restrict__ gw, float* __restrict__ g_s,
    const float* __restrict__ s, const float* __restrict__ zp, int64_t co, int64_t row,
    const float* __restrict__ stats, const float* __restrict__ gzp_extra, const int8_t* __restrict__ r_sign,
    uint64_t seed, uint64_t offset, const float* __restrict__ mx, const float* __restrict__ g_lwq, const int64_t c,
    const int64_t rng_base, BwdLdsOf<METHOD, RSIGN>& lds) {
  double* const sm = lds.sm;
  double* const sm4 = lds.sm4;
  int* const smt = lds.smt;
  const int items = (int)(row >> 2), T = blockDim.x;
  const vf4* wrow = reinterpret_cast<const vf4*>(w + c * row);
  const vf4* grow = reinterpret_cast<const vf4*>(G + c * row);
  vf4 xv[NV], gv4[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int j = threadIdx.x + k * T;
    const int jc = j < items ? j : items - 1;          // unconditional loads, clamped index: see pc_fwd_reg_body
    xv[k] = pc_ld<(NT == 1 || NT == 2)>(wrow + jc);
    gv4[k] = pc_ld<(NT == 1 || NT == 2)>(grow + jc);
  }
  __builtin_amdgcn_sched_barrier(0);
  // the channel's parameters go out under the row loads and in front of the sign tile's barrier (see pc_fwd_reg_body)
  const float sc = ldg(s + c), z = ldg(zp + c);
  const float rmx = LAYER ? ldg(mx + c) : 0.f;
  const float glw_c = (LAYER && g_lwq) ? ldg(g_lwq + c) : 0.f;       // (read here, used after the row sums)
  const float gzx_c = gzp_extra ? ldg(gzp_extra + c) : 0.f;
  // the row's sign bits: ceil(row / 128) (+1) Philox calls by the first threads of the workgroup, under the loads
  constexpr bool PHILOX = (METHOD != MHAQ_FQ_LSQ) && !RSIGN;
  uint32_t* const stile = lds.stile;
  const BwdCtx kx = make_bwd_ctx(sc, z, -INFINITY, INFINITY);     // (its division: in front of the tile's barrier too)
  int rel0 = 0;                        // < 128: the row's first element inside the tile (rows here always fit it)
  if (PHILOX) {
    rel0 = (int)row_signs_begin<true>(stile, rng_base + c * row, row, seed, offset).rel0;
    __syncthreads();
  }
  MHAQ_TRACE_AT(2, true);

  // AEWGS walks the row twice (statistics, then gradients): with <= 4 float4 per thread the quotients v = (w - zp) / s
  // of the first walk stay in registers for the second (16 VGPRs; at 8 float4 per thread they would cost occupancy)
  constexpr bool KEEP_V = (METHOD == MHAQ_FQ_AEWGS) && NV <= 4;
  // Packed fp32 (two elements per instruction, exact_v2 / aewgs_pair above) where it was measured faster: rows of more than
  // 4 float4 per thread ([8192,8192] 144.3 -> 140.8 us, [1024,16384] 46.1 -> 43.6, ResNet-18's two long-row groups 19.0 /
  // 20.0 -> 17.9 / 18.4 us).  At <= 4 float4 per thread the register pairs it needs cost the kept quotients or a wave per
  // SIMD, and [4096,4096] / [50257,768] lose 2 / 5 % (gpurun_out/r04f_pk_pc.txt): those keep the scalar code.
  // (MULTI: inside the model-wide launches the packed element also pays at <= 4 float4 per thread -- 256-thread rows, 93 VGPRs, the
  // 5 waves per SIMD the 6 MB group needs: 10.1 -> 9.6 us cold, 9.6 -> 9.2 warm --; the per-layer launches of such rows keep the scalar
  // element: [50257,768] loses 4 % with it.  profiles/r06_pc_multi_packed_ste.txt, last section)
  constexpr bool PACKED = (METHOD == MHAQ_FQ_AEWGS) && MHAQ_PACKED && (NV > 4 || (MULTI && MHAQ_PACKED_AEWGS_MULTI));
  float vkeep[KEEP_V ? 4 * NV : 1];
  const bool have_v = KEEP_V && !stats;
  float delta = 0.f;
  if (METHOD == MHAQ_FQ_AEWGS) {
    float num, e2, me;
    if (stats) {
      num = ldg(stats + c); e2 = ldg(stats + co + c); me = ldg(stats + 2 * co + c);
    } else {
      double st[3] = {0, 0, 0};
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        if ((int)threadIdx.x + k * T < items) {
          const float xe[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w};
          const float ge[4] = {gv4[k].x, gv4[k].y, gv4[k].z, gv4[k].w};
          if (PACKED && kx.fast_div) {                   // two elements per instruction: the same bits (see exact_v2)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const vf2 v = exact_v2(vf2{xe[2 * h], xe[2 * h + 1]}, sc, kx.rs, z);
              if (KEEP_V) { vkeep[KEEP_V ? 4 * k + 2 * h : 0] = v.x; vkeep[KEEP_V ? 4 * k + 2 * h + 1 : 0] = v.y; }
              const vf2 rn = {rintf(v.x), rintf(v.y)};
              const vf2 n = rn - v, nn = n * n;
              const vf2 gq = vf2{ge[2 * h], ge[2 * h + 1]} * vf2{sc, sc};
              st[0] += (double)(sign_f(gq.x) * n.x); st[1] += (double)nn.x; st[2] += (double)n.x;
              st[0] += (double)(sign_f(gq.y) * n.y); st[1] += (double)nn.y; st[2] += (double)n.y;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const QCore qc = quant_core_w(xe[q], kx);
              if (KEEP_V) vkeep[KEEP_V ? 4 * k + q : 0] = qc.v;
              const float gq = ge[q] * sc;
              st[0] += (double)(sign_f(gq) * qc.n);
              st[1] += (double)(qc.n * qc.n);
              st[2] += (double)qc.n;
            }
          }
        }
      }
      block_sum_all<3>(st, sm);
      const float inv = (float)row;
      num = (float)st[0] / inv;
      e2 = (float)st[1] / inv;
      me = (float)st[2] / inv;
    }
    delta = aewgs_delta(num, e2, me);
  }

  // pass 1: per-channel sums.  gW = gv/s everywhere except at the row's extremes, which also take a share of a
  // REDUCED gradient: every float4 without an extreme element is stored right here, before the row reduction, so the
  // store stream does not wait behind the barrier and the row does not stay in registers across it.  A thread's FIRST
  // float4 that holds a minimum (or maximum) waits in one 8-register slot for pass 2; a second one in the same thread
  // (rare: the row's minimum and maximum, or a tie, in one thread's share) is stored as it is and patched in pass 2
  // through a read-back (w from L2, gv/s = this thread's own store), which costs that wave a store round trip.
  vf4* orow = reinterpret_cast<vf4*>(gw + c * row);
  uint32_t deferred = 0;
  int slot_k = -1;
  vf4 slot_x = vf4{0.f, 0.f, 0.f, 0.f}, slot_p = slot_x;
  double acc[4] = {0, 0, 0, 0};  // d/ds, sum(G - gv/s), count(w == min), count(w == max)
  int cnt_min = 0, cnt_max = 0;  // per-lane integer tallies: exact in any order, 2 VALU instead of a 64-bit select + add
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int j = threadIdx.x + k * T;
    if (j < items) {
      float r[4] = {0.f, 0.f, 0.f, 0.f};
      if (METHOD != MHAQ_FQ_LSQ) {
        // r[] holds rc = r * 3^-1/2 (nibble_to_rc4): the noise term below is gq * rc, the bits of (3^-1/2 * gq) * r
        if (RSIGN) {
          const int64_t i = rng_base + c * row + ((int64_t)j << 2);
#pragma unroll
          for (int q = 0; q < 4; ++q) r[q] = sign_to_rc(sign_half(r_sign[i + q]));
        } else {
          nibble_to_rc4(sign_tile_nibble(stile, rel0 + (j << 2)), r);
        }
      }
      const float xe[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w};
      const float ge[4] = {gv4[k].x, gv4[k].y, gv4[k].z, gv4[k].w};
      float park[4];
      // a float4 holds a row extreme iff its own minimum (maximum) equals the row's: 6 min / max + 2 compares per
      // float4 instead of 8 compares, 8 counter updates and the or-chain; the exact tallies are taken only there
      // (NaN rows have no extremes either way: their minimum is NaN and compares unequal to everything)
      const float m4 = fminf(fminf(xe[0], xe[1]), fminf(xe[2], xe[3]));
      const float M4 = fmaxf(fmaxf(xe[0], xe[1]), fmaxf(xe[2], xe[3]));
      const bool extreme = (m4 == z) | (LAYER && (M4 == rmx));
      if (extreme) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          cnt_min += (xe[q] == z) ? 1 : 0;
          if (LAYER) cnt_max += (xe[q] == rmx) ? 1 : 0;
        }
      }
      constexpr bool PACKED_SL = (METHOD == MHAQ_FQ_STE || METHOD == MHAQ_FQ_LSQ) && MHAQ_PACKED_STE && (NV > 4 || MHAQ_PACKED_STE_ALL);
      const bool packed = (PACKED || PACKED_SL) && kx.fast_div;
      if (PACKED_SL && packed) {                  // two elements per instruction: the same bits (see ste_pair)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const StePair pp = ste_pair<METHOD>(vf2{xe[2 * h], xe[2 * h + 1]}, vf2{ge[2 * h], ge[2 * h + 1]},
                                              vf2{r[2 * h], r[2 * h + 1]}, kx);
          acc[0] += (double)pp.term.x; acc[1] += (double)(ge[2 * h] - pp.gvs.x);
          acc[0] += (double)pp.term.y; acc[1] += (double)(ge[2 * h + 1] - pp.gvs.y);
          park[2 * h] = pp.gvs.x; park[2 * h + 1] = pp.gvs.y;
        }
      }
      if (PACKED && packed) {                     // two elements per instruction: the same bits (see aewgs_pair)
        vf2 va, vb;
        if (have_v) {
          va = vf2{vkeep[KEEP_V ? 4 * k : 0], vkeep[KEEP_V ? 4 * k + 1 : 0]};
          vb = vf2{vkeep[KEEP_V ? 4 * k + 2 : 0], vkeep[KEEP_V ? 4 * k + 3 : 0]};
        } else {
          va = exact_v2(vf2{xe[0], xe[1]}, sc, kx.rs, z);
          vb = exact_v2(vf2{xe[2], xe[3]}, sc, kx.rs, z);
        }
        const AewgsPair pa = aewgs_pair(va, vf2{ge[0], ge[1]}, vf2{r[0], r[1]}, sc, delta);
        const AewgsPair pb = aewgs_pair(vb, vf2{ge[2], ge[3]}, vf2{r[2], r[3]}, sc, delta);
        exact_quot4(pa.gv, pb.gv, kx, park);
        acc[0] += (double)pa.term.x; acc[1] += (double)(ge[0] - park[0]);
        acc[0] += (double)pa.term.y; acc[1] += (double)(ge[1] - park[1]);
        acc[0] += (double)pb.term.x; acc[1] += (double)(ge[2] - park[2]);
        acc[0] += (double)pb.term.y; acc[1] += (double)(ge[3] - park[3]);
      }
#pragma unroll
      for (int q = 0; q < 4 && !packed; ++q) {
        const float x = xe[q], g = ge[q];
        const float gq = g * sc;
        float gvs;
        if (METHOD == MHAQ_FQ_AEWGS) {
          float v, n;
          if (have_v) {            // the statistics pass left v = (w - zp) / s in registers (rows of <= 4 float4 per thread)
            v = vkeep[KEEP_V ? 4 * k + q : 0];
            n = rintf(v) - v;
          } else {
            const QCore qc = quant_core_w(x, kx);
            v = qc.v;
            n = qc.n;
          }
          // d/ds term g*q - gv*(v/s) with gv = gq*(1 - gsc) is g*(n + v*gsc) in real arithmetic (see pc_bwd_body)
          const float gsc = aewgs_gsc(gq, n, delta);
          const float gv = gq + (-gq * gsc);
          gvs = quot(gv, kx);
          acc[0] += (double)(g * (n + v * gsc) + gq * r[q]);
        } else {
          const QCore qc = quant_core_w(x, kx);
          const float gv = gq + noise_grad_v<METHOD>(gq, qc.n, delta);
          gvs = (METHOD == MHAQ_FQ_STE || METHOD == MHAQ_FQ_LSQ) ? quot_of_product(g, gv, kx) : quot(gv, kx);
          const float noise_s = (METHOD == MHAQ_FQ_LSQ) ? gq * qc.n : gq * r[q];
          if (METHOD == MHAQ_FQ_STE || METHOD == MHAQ_FQ_LSQ)
            acc[0] += (double)(g * qc.n + noise_s);
          else
            acc[0] += (double)((g * qc.q + (-gv) * quot(qc.v, kx)) + noise_s);
        }
        acc[1] += (double)(g - gvs);
        park[q] = gvs;
      }
      const vf4 p4 = vf4{park[0], park[1], park[2], park[3]};
      if (extreme && slot_k < 0) {
        slot_k = k; slot_x = xv[k]; slot_p = p4;
      } else {
        if (extreme) deferred |= 1u << k;
        pc_st<(NT == 1 || NT == 3)>(orow + j, p4);
      }
    }
  }
  {                                    // every thread holds the two row sums and the two tie counts after this one barrier
    double d2[2] = {acc[0], acc[1]}, t2[2];
    block_sum_all_tally<2>(d2, cnt_min, cnt_max, t2, sm4, smt);
    acc[0] = d2[0]; acc[1] = d2[1]; acc[2] = t2[0]; acc[3] = t2[1];
  }
  MHAQ_TRACE_AT(3, false);
  float gzp_local = (float)acc[1];
  if (gzp_extra) gzp_local = gzp_local + gzx_c;
  float gs_local = (float)acc[0];
  float t_local = 0.f;
  if (LAYER) {
    if (g_lwq) t_local = glw_c / (((rmx - z) + sc) * MHAQ_LN2F);
    gzp_local = gzp_local - t_local;
    gs_local = gs_local + t_local;
  }
  if (threadIdx.x == 0) stg(g_s + c, LAYER ? (gs_local * sc) * MHAQ_LN2F : gs_local);   // exp2 backward when LAYER
  const float gzp = gzp_local;
  const float cnt = (float)acc[2];
  const float tie = (gzp * 1.0f) / cnt;  // amin backward: (grad * mask) / count
  float tie_max = 0.f;
  if (LAYER) tie_max = (t_local * 1.0f) / (float)acc[3];   // amax backward
  // pass 2: the deferred float4: gW = gv/s + tie-split share of the zero-point (and range) gradient
  auto with_shares = [&](const vf4 x4, const vf4 p4) {
    const float xe[4] = {x4.x, x4.y, x4.z, x4.w};
    const float pe[4] = {p4.x, p4.y, p4.z, p4.w};
    float o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      o[q] = (xe[q] == z) ? pe[q] + tie : pe[q];
      if (LAYER && xe[q] == rmx) o[q] = o[q] + tie_max;
    }
    return vf4{o[0], o[1], o[2], o[3]};
  };
  if (slot_k >= 0) pc_st<(NT == 1 || NT == 3)>(orow + (threadIdx.x + slot_k * T), with_shares(slot_x, slot_p));
  if (deferred) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int j = threadIdx.x + k * T;
      if (deferred & (1u << k))                                             // same-thread read-after-write on gW
        pc_st<(NT == 1 || NT == 3)>(orow + j, with_shares(pc_ld<false>(wrow + j), pc_ld<false>(orow + j)));
    }
  }
  MHAQ_TRACE_AT(4, false);
  MHAQ_TRACE_AT(5, true);
}

// Waves per SIMD the register allocator must leave room for.  The bodies hold nothing across the row reduction (every
// float4 is stored in pass 1), so the estimators fit 64 VGPRs up to 4 float4 per thread -- two 1024-thread workgroups per CU,
// in different phases, instead of one ([1024,16384] rows) -- and AEWGS, whose statistics walk keeps the quotients, 80.
// Each setting below compiles without scratch (tools/kernel_regs.py fq_pc.hip pc_bwd_reg_kernel).
#ifndef MHAQ_PCREG_MINW
#define MHAQ_PCREG_MINW(METHOD, NV)                                                  \
  ((METHOD) == MHAQ_FQ_AEWGS ? ((NV) <= 2 ? 6 : ((NV) <= 4 ? 5 : 4))                 \
                             : ((NV) <= 4 ? (((METHOD) == MHAQ_FQ_EWGS && (NV) == 4) ? 6 : 8) : 4))
#endif
template <int METHOD, bool RSIGN, bool LAYER, int NV, int NT>
__global__ __launch_bounds__(64 * kMaxWaves, MHAQ_PCREG_MINW(METHOD, NV)) void pc_bwd_reg_kernel(
    const float* __restrict__ w, const float* __restrict__ G, float* __restrict__ gw, float* __restrict__ g_s,
    const float* __restrict__ s, const float* __restrict__ zp, int64_t co, int64_t row,
    const float* __restrict__ stats, const float* __restrict__ gzp_extra, const int8_t* __restrict__ r_sign,
    uint64_t seed, uint64_t offset, const uint64_t* __restrict__ offset_dev, const float* __restrict__ mx,
    const float* __restrict__ g_lwq) {
  __shared__ BwdLdsOf<METHOD, RSIGN> lds;
  pc_bwd_reg_body<METHOD, RSIGN, LAYER, NV, NT>(w, G, gw, g_s, s, zp, co, row, stats, gzp_extra, r_sign, seed,
                                                stream_offset(offset, offset_dev), mx, g_lwq, blockIdx.x, 0, lds);
}

// NV (float4 per thread) and the thread count for a register-resident row; 0 = the row does not qualify.
// Forward: few threads, up to 8 float4 each (60 VGPRs at NV = 8: full occupancy either way, and fewer, longer waves
// measured faster: [4096,4096] 25.6 us at 128 x 8 against 30.0 at 256 x 4).  Backward holds two rows per thread
// (104 VGPRs at NV = 8 = 4 waves per SIMD): at most 4 float4 per row and thread where the row allows it
// ([4096,4096] 40.0 us at 256 x 4 against 43.4 at 128 x 8; [1024,16384] 42.1 at 1024 x 4 against 44.5 at 512 x 8).
// AEWGS backward makes two row reductions (statistics, then sums): wide workgroups pay for both barriers, so its
// plan stops growing at 256 threads and takes 8 float4 per thread beyond ([1024,16384]: 68 us at 512 x 8, 85 at 1024 x 4).
#ifndef MHAQ_AEWGS_CAP
#define MHAQ_AEWGS_CAP 256    // A/B knob (tools/variants.sh): widest workgroup the two-reduction AEWGS rows grow to at 4 float4 per thread
#endif
static inline int reg_plan(int64_t row, bool vec, int* threads, bool backward = false, bool two_reductions = false) {
  if (!vec) return 0;
  const int64_t items = row >> 2;
  int t;
  if (backward) {
    const int cap = two_reductions ? MHAQ_AEWGS_CAP : 64 * kMaxWaves;
    t = 64;
    while (t < cap && items > (int64_t)t * 4) t *= 2;
    while (t < 64 * kMaxWaves && items > (int64_t)t * 8) t *= 2;
  } else {
    t = items <= 256 ? 64 : (items <= 1024 ? 128 : 256);
    if (items > (int64_t)t * 8) t = 512;                        // two workgroups per CU stay resident
    if (items > (int64_t)t * 8) t = 64 * kMaxWaves;
  }
  if (items > (int64_t)t * 8) return 0;
  const int64_t per = (items + t - 1) / t;
  *threads = t;
  return per <= 2 ? 2 : (per <= 4 ? 4 : 8);
}

// Multi-tensor backward: aux_all is the forward's [4][total_co] slab; gw_all / g_log_s_all are slabs laid out
// like wq_all / one aux row.  stats_all: nullable [3][total_co] AEWGS statistics (after the all-reduce).
// The sign stream of layer L is the single-layer stream shifted by the layer's element offset.
template <int METHOD, bool STAGE>
__global__ void pc_bwd_multi_kernel(const WLayerDesc* __restrict__ descs, int nlayers,
                                    const float* __restrict__ aux_all, int64_t aux_stride,
                                    float* __restrict__ gw_all, float* __restrict__ g_log_s_all,
                                    const float* __restrict__ stats_all, int64_t stats_stride, uint64_t seed,
                                    uint64_t offset, const uint64_t* __restrict__ offset_dev) {
  offset = stream_offset(offset, offset_dev);
  const WLayerDesc d = descs[find_layer(descs, nlayers, multi_channel())];
  const float* a = aux_all + d.chan_offset;
  // AEWGS statistics are indexed stats[c], stats[co + c], stats[2co + c] inside the body: pass a view whose
  // "co" stride is the slab's row stride by pointing at this layer's first channel
  float* gw = gw_all + d.elem_offset;
  const float* st = stats_all ? stats_all + d.chan_offset : nullptr;
  const int64_t sco = stats_all ? stats_stride : d.co, c = multi_channel() - d.chan_offset;
  // a table that does not cover the grid must not turn into a stray access (d.w in the test: the descriptor's two halves
  // are then read back to back, not the pointers after the branch on the offsets)
  if ((c < 0) | (c >= d.co) | (d.w == nullptr)) return;
  __shared__ BwdLdsOf<METHOD, false> lds;
  if (vec_ok(d.row, d.w, d.G, gw) && (d.elem_offset & 3) == 0)     // per layer, workgroup-uniform
    pc_bwd_body<METHOD, false, STAGE, true, true>(d.w, d.G, gw, g_log_s_all + d.chan_offset, a, a + aux_stride, sco,
                                                  d.row, st, nullptr, nullptr, seed, offset, a + 2 * aux_stride,
                                                  d.g_lwq, c, d.elem_offset, lds);
  else
    pc_bwd_body<METHOD, false, STAGE, true, false>(d.w, d.G, gw, g_log_s_all + d.chan_offset, a, a + aux_stride, sco,
                                                   d.row, st, nullptr, nullptr, seed, offset, a + 2 * aux_stride,
                                                   d.g_lwq, c, d.elem_offset, lds);
}

// The same grid with register-resident rows (see pc_fwd_multi_reg_kernel): rows that are whole aligned float4s and fit NV
// float4 per thread take pc_bwd_reg_body -- one HBM read of W and G, stores of the non-extreme float4 before the row
// reduction --, the others the staged body.  Same element -> thread mapping per row, fp64 row sums: the per-layer bits.
// (min waves per SIMD as for pc_bwd_reg_kernel; this kernel also carries the staged bodies: 7 where that one has 8)
#ifndef MHAQ_PCMULTI_MINW
#define MHAQ_PCMULTI_MINW(METHOD, NV)                                                \
  ((METHOD) == MHAQ_FQ_AEWGS ? ((NV) <= 2 ? 6 : ((NV) <= 5 ? 5 : 4))                 \
                             : ((NV) <= 4 ? (((METHOD) == MHAQ_FQ_EWGS && (NV) == 4) ? 6 : 7) : ((NV) == 5 ? ((METHOD) == MHAQ_FQ_EWGS ? 5 : 6) : 4)))
#endif
template <int METHOD, int NV, int TB>
__global__ __launch_bounds__(TB, (TB != kBlock ? 1 : MHAQ_PCMULTI_MINW(METHOD, NV))) void pc_bwd_multi_reg_kernel(
    const WLayerDesc* __restrict__ descs, int nlayers, const float* __restrict__ aux_all, int64_t aux_stride,
    float* __restrict__ gw_all, float* __restrict__ g_log_s_all, const float* __restrict__ stats_all,
    int64_t stats_stride, uint64_t seed, uint64_t offset, const uint64_t* __restrict__ offset_dev) {
  MHAQ_TRACE_AT(0, false);
  stagger_second_half<MHAQ_BWD_STAGGER_NS>();
  offset = stream_offset(offset, offset_dev);
  const WLayerDesc d = descs[find_layer(descs, nlayers, multi_channel())];
  const float* a = aux_all + d.chan_offset;
  float* gw = gw_all + d.elem_offset;
  const float* st = stats_all ? stats_all + d.chan_offset : nullptr;
  const int64_t sco = stats_all ? stats_stride : d.co, c = multi_channel() - d.chan_offset;
  // a table that does not cover the grid must not turn into a stray access (d.w in the test: the descriptor's two halves
  // are then read back to back, not the pointers after the branch on the offsets)
  if ((c < 0) | (c >= d.co) | (d.w == nullptr)) return;
  const bool vec = vec_ok(d.row, d.w, d.G, gw) && (d.elem_offset & 3) == 0;
  MHAQ_TRACE_AT(1, true);
  __shared__ BwdLdsOf<METHOD, false> lds;
  if (vec && (d.row >> 2) <= (int64_t)NV * TB)
    pc_bwd_reg_body<METHOD, false, true, NV, MHAQ_BWD_MULTI_NT, true>(d.w, d.G, gw, g_log_s_all + d.chan_offset, a, a + aux_stride, sco,
                                                    d.row, st, nullptr, nullptr, seed, offset, a + 2 * aux_stride,
                                                    d.g_lwq, c, d.elem_offset, lds);
  else if (vec)
    pc_bwd_body<METHOD, false, false, true, true>(d.w, d.G, gw, g_log_s_all + d.chan_offset, a, a + aux_stride, sco,
                                                  d.row, st, nullptr, nullptr, seed, offset, a + 2 * aux_stride,
                                                  d.g_lwq, c, d.elem_offset, lds);
  else
    pc_bwd_body<METHOD, false, false, true, false>(d.w, d.G, gw, g_log_s_all + d.chan_offset, a, a + aux_stride, sco,
                                                   d.row, st, nullptr, nullptr, seed, offset, a + 2 * aux_stride,
                                                   d.g_lwq, c, d.elem_offset, lds);
}

// AEWGS statistics of a GROUP of layers in one grid (the data-parallel trainer's exchange: ONE packed all-reduce
// per group instead of one per layer): stats[3][group_co], scales and zero points from the forward's aux slab.
template <bool VEC>
__device__ __forceinline__ void pc_stats_row(const float* __restrict__ w, const float* __restrict__ G, float sc,
                                             float z, int64_t row, double (&st)[3]) {
  constexpr int W = VEC ? 4 : 1;
  const BwdCtx kx = make_bwd_ctx(sc, z, -INFINITY, INFINITY);
#pragma unroll 2
  for (int64_t j = (int64_t)threadIdx.x * W; j < row; j += (int64_t)blockDim.x * W) {
    float x[W], g[W];
    ldvg<W>(w + j, x);
    ldvg<W>(G + j, g);
#pragma unroll
    for (int k = 0; k < W; ++k) pc_stats_accumulate(x[k], g[k], sc, kx, st);
  }
}

__global__ __launch_bounds__(kBlock) void pc_aewgs_stats_multi_kernel(
    const WLayerDesc* __restrict__ descs, int nlayers, const float* __restrict__ aux_all, int64_t aux_stride,
    float* __restrict__ stats, int64_t group_co) {
  __shared__ double sm[3 * 4];
  const WLayerDesc d = descs[find_layer(descs, nlayers, multi_channel())];
  const int64_t c = multi_channel() - d.chan_offset;
  if (c < 0 || c >= d.co) return;
  const int64_t ch = multi_channel();
  const float sc = aux_all[ch], z = aux_all[aux_stride + ch];
  const float* wrow = d.w + c * d.row;
  const float* grow = d.G + c * d.row;
  double st[3] = {0, 0, 0};
  if (vec_ok(d.row, d.w, d.G)) pc_stats_row<true>(wrow, grow, sc, z, d.row, st);
  else pc_stats_row<false>(wrow, grow, sc, z, d.row, st);
  block_sum<3>(st, sm);
  if (threadIdx.x == 0) {
    const float inv = (float)d.row;
    stats[ch] = (float)st[0] / inv;
    stats[group_co + ch] = (float)st[1] / inv;
    stats[2 * group_co + ch] = (float)st[2] / inv;
  }
}

// ------------------------------------------------------------------ quantize with GIVEN per-row parameters
// Quantizer.quantize (gdnsq.py:197-208) on a [co][row] tensor whose scale AND zero point are given per row: what the
// two-method facade does to a weight outside the step loop -- utils/model_stats.py:118,123 quantizes the detached weights
// with the zero point the layer's last forward left in Q (NOT the row minimum of the weights as they are now, which
// mhaq_fq_pc_fwd would take).  Bounds are the weight quantizer's -inf / +inf (gdnsq_conv2d.py:76-77).  Cold path: plain
// coalesced dwords, any row length and alignment, IEEE division.  flags (nullable, zeroed by the caller): the eval asserts
// of gdnsq.py:211-217 -- with infinite bounds only "not an integer" can fire, and q = v + (rne(v) - v) is not an integer
// exactly when it is NaN (see fwd_elem in fq_pt.hip).
__global__ __launch_bounds__(kBlock) void pc_quantize_kernel(const float* __restrict__ x, float* __restrict__ q_out,
                                                             float* __restrict__ y_out, const float* __restrict__ s,
                                                             const float* __restrict__ zp, int64_t row, int chunks,
                                                             int32_t* __restrict__ flags) {
  const int64_t c = blockIdx.x / chunks;
  const int64_t j0 = (int64_t)(blockIdx.x % chunks) * (kBlock * 4);
  const float sc = s[c], z = zp[c];
  bool bad = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t j = j0 + k * kBlock + threadIdx.x;
    if (j < row) {
      const QCore qc = quant_core(x[c * row + j], sc, z, -INFINITY, INFINITY);
      q_out[c * row + j] = qc.q;
      if (y_out) y_out[c * row + j] = dequant(qc.q, sc, z);
      bad |= (qc.q != qc.q);
    }
  }
  if (flags && __ballot(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(flags, (int32_t)MHAQ_FQ_FLAG_NOT_INTEGER);
}

// ------------------------------------------------------------------ per-element parameters
// Quantizer with one (scale, zero point) PER ELEMENT: the quant_bias=True branch of
// gdnsq_conv2d.py:86-94 (bias[c] uses s.ravel()[c], zp.ravel()[c]).  n is small (= C_out).
template <bool WRITE_Q>
__global__ __launch_bounds__(kBlock) void vec_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ q_out, const float* __restrict__ s,
                                                         const float* __restrict__ zp, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    QCore q = quant_core(x[i], s[i], zp[i], -INFINITY, INFINITY);
    y[i] = dequant(q.q, s[i], zp[i]);
    if (WRITE_Q) q_out[i] = q.q;
  }
}

// stats[0..2] = means over ALL n elements (reduce_to_shape with no unit dims reduces everything)
__global__ __launch_bounds__(kBlock) void vec_aewgs_stats_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ g,
                                                                 const float* __restrict__ s,
                                                                 const float* __restrict__ zp, int64_t n,
                                                                 float* __restrict__ stats) {
  __shared__ double sm[3 * 4];
  double st[3] = {0, 0, 0};
  for (int64_t i = threadIdx.x; i < n; i += kBlock) pc_stats_accumulate_div(x[i], g[i], s[i], zp[i], st);
  block_sum<3>(st, sm);
  if (threadIdx.x == 0) {
    const float inv = (float)n;
    stats[0] = (float)st[0] / inv;
    stats[1] = (float)st[1] / inv;
    stats[2] = (float)st[2] / inv;
  }
}

template <int METHOD, bool RSIGN>
__global__ __launch_bounds__(kBlock) void vec_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                         float* __restrict__ gx, float* __restrict__ g_s,
                                                         float* __restrict__ g_zp, const float* __restrict__ s,
                                                         const float* __restrict__ zp, int64_t n,
                                                         const float* __restrict__ stats,
                                                         const int8_t* __restrict__ r_sign, uint64_t seed,
                                                         uint64_t offset, const uint64_t* __restrict__ offset_dev) {
  offset = stream_offset(offset, offset_dev);
  float delta = 0.f;
  if (METHOD == MHAQ_FQ_AEWGS) delta = aewgs_delta(stats[0], stats[1], stats[2]);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const float sc = s[i];
    QCore q = quant_core(x[i], sc, zp[i], -INFINITY, INFINITY);
    const float gq = g[i] * sc;
    const float gv = gq + noise_grad_v<METHOD>(gq, q.n, delta);
    const float gvs = gv / sc;
    float noise_s;
    if (METHOD == MHAQ_FQ_LSQ) {
      noise_s = gq * q.n;
    } else {
      const float r = RSIGN ? sign_half(r_sign[i]) : philox_r(i, seed, offset);
      noise_s = (MHAQ_INV_SQRT3 * gq) * r;
    }
    gx[i] = gvs;
    g_s[i] = (g[i] * q.q + (-gv) * (q.v / sc)) + noise_s;
    g_zp[i] = g[i] - gvs;
  }
}

template <int METHOD>
static int launch_vec_bwd(const float* x, const float* g, float* gx, float* g_s, float* g_zp, const float* s,
                          const float* zp, int64_t n, const float* stats, const int8_t* r_sign, uint64_t seed,
                          uint64_t offset, const uint64_t* offset_dev, hipStream_t st) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b > kMaxBlocks) b = kMaxBlocks;
  if (r_sign) MHAQ_LAUNCH((vec_bwd_kernel<METHOD, true>), dim3((unsigned)b), dim3(kBlock), 0, st, x, g, gx, g_s, g_zp, s, zp, n, stats, r_sign, seed, offset, offset_dev);
  else MHAQ_LAUNCH((vec_bwd_kernel<METHOD, false>), dim3((unsigned)b), dim3(kBlock), 0, st, x, g, gx, g_s, g_zp, s, zp, n, stats, r_sign, seed, offset, offset_dev);
  return launch_status();
}


// ------------------------------------------------------------------ QN* autograd Functions (unfused facade)
// The reference's own custom Functions, QNoise.forward and QN{STE,LSQ,EWGS,AEWGS}.backward
// (gdnsq.py:11-147), for callers that use Quantizer.quantize / dequantize / _get_rnoise
// separately (utils/model_stats.py:116-132).  `groups` scales, each owning `len` consecutive
// elements: groups == 1 is a per-tensor scale, groups == C_out a per-channel one.
__global__ __launch_bounds__(kBlock) void noise_fwd_kernel(const float* __restrict__ v, float* __restrict__ out,
                                                           int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    out[i] = rintf(v[i]) - v[i];
}

// grid = (groups, slices) -- the groups on x (the y dimension stops at 65535): up to 2^24 - 1 of them, HIP's bound on
// gridDim.x * blockDim.x at 256 threads (a per-element scale of a wide bias, a per-channel scale of a [50257, 768] Linear); partial[group * slices + slice] = fp64 sum of the scale-gradient terms.
// stats layout: per-group [3][groups] when period == 0, per-position [3][period] otherwise.
template <int METHOD, bool RSIGN>
__global__ __launch_bounds__(kBlock) void noise_bwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                           float* __restrict__ gv, int64_t groups, int64_t len,
                                                           const float* __restrict__ stats, int64_t period,
                                                           const int8_t* __restrict__ r_sign, uint64_t seed,
                                                           uint64_t offset, const uint64_t* __restrict__ offset_dev, double* __restrict__ partial) {
  __shared__ double sm[4];
  offset = stream_offset(offset, offset_dev);
  const int64_t grp = blockIdx.x;
  float delta_g = 0.f;
  if (METHOD == MHAQ_FQ_AEWGS && period == 0)
    delta_g = aewgs_delta(stats[grp], stats[groups + grp], stats[2 * groups + grp]);
  double acc[1] = {0.0};
  for (int64_t j = (int64_t)blockIdx.y * kBlock + threadIdx.x; j < len; j += (int64_t)gridDim.y * kBlock) {
    const int64_t i = grp * len + j;
    const float x = v[i], go = g[i];
    const float e = rintf(x) - x;
    float delta = delta_g;
    if (METHOD == MHAQ_FQ_AEWGS && period > 0) {
      const int64_t c = i % period;
      delta = aewgs_delta(stats[c], stats[period + c], stats[2 * period + c]);
    }
    gv[i] = noise_grad_v<METHOD>(go, e, delta);
    float t;
    if (METHOD == MHAQ_FQ_LSQ) {
      t = go * e;
    } else {
      const float r = RSIGN ? sign_half(r_sign[i]) : philox_r(i, seed, offset);
      t = (MHAQ_INV_SQRT3 * go) * r;
    }
    acc[0] += (double)t;
  }
  block_sum<1>(acc, sm);
  if (threadIdx.x == 0) partial[grp * gridDim.y + blockIdx.y] = acc[0];
}

__global__ __launch_bounds__(kBlock) void noise_bwd_finalize_kernel(const double* __restrict__ partial, int slices,
                                                                    float* __restrict__ gs) {
  __shared__ double sm[4];
  double acc[1] = {0.0};
  for (int i = threadIdx.x; i < slices; i += kBlock) acc[0] += partial[(int64_t)blockIdx.x * slices + i];
  block_sum<1>(acc, sm);
  if (threadIdx.x == 0) gs[blockIdx.x] = (float)acc[0];
}

static inline int noise_slices(int64_t groups, int64_t len) {
  int64_t s = (len + kBlock * 8 - 1) / (kBlock * 8);
  const int64_t cap = groups >= kMaxBlocks ? 1 : kMaxBlocks / groups;
  if (s > cap) s = cap;
  return (int)(s < 1 ? 1 : s);
}

template <int METHOD>
static int launch_noise_bwd(const float* v, const float* g, float* gv, int64_t groups, int64_t len,
                            const float* stats, int64_t period, const int8_t* r_sign, uint64_t seed,
                            uint64_t offset, const uint64_t* offset_dev, double* partial, int slices, hipStream_t st) {
  dim3 grid((unsigned)groups, (unsigned)slices);      // slices <= kMaxBlocks / groups, at most 2048
  if (r_sign) MHAQ_LAUNCH((noise_bwd_kernel<METHOD, true>), grid, dim3(kBlock), 0, st, v, g, gv, groups, len, stats, period, r_sign, seed, offset, offset_dev, partial);
  else MHAQ_LAUNCH((noise_bwd_kernel<METHOD, false>), grid, dim3(kBlock), 0, st, v, g, gv, groups, len, stats, period, r_sign, seed, offset, offset_dev, partial);
  return launch_status();
}


// ------------------------------------------------------------------ small PER_TENSOR weight layers
// A PER_TENSOR weight quantizer of a CIFAR-sized layer (<= 64 K elements, L2-resident) needs a global
// minimum before the first element can be quantized; instead of minmax + finalize + forward launches
// (and torch's exp2 / amin / amax / log2 chain for the regulariser input) ONE workgroup of 1024
// threads does the whole layer: s = exp2(log_s), min, max, lwq = log2(max - min + s), quantize.
constexpr int kSmallThreads = 1024;
constexpr int64_t kSmallMaxElems = 64 * 1024;

__global__ __launch_bounds__(kSmallThreads) void wt_small_fwd_kernel(const float* __restrict__ w,
                                                                     float* __restrict__ wq,
                                                                     const float* __restrict__ log_s, int64_t n,
                                                                     float* __restrict__ aux /* s zp mx lwq */) {
  __shared__ float red[kSmallThreads / 64 + 1];
  float mn = INFINITY, mx = -INFINITY;
  bool nan = false;
  for (int64_t i = threadIdx.x; i < n; i += kSmallThreads) {
    const float v = w[i];
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
    nan |= (v != v);
  }
  const float zp = block_min_bcast(mn, nan, red);
  const float rmx = -block_min_bcast(-mx, nan, red);
  const float sc = exp2f(*log_s);
  if (threadIdx.x == 0) {
    aux[0] = sc; aux[1] = zp; aux[2] = rmx; aux[3] = log2f((rmx - zp) + sc);
  }
  for (int64_t i = threadIdx.x; i < n; i += kSmallThreads) {
    QCore q = quant_core(w[i], sc, zp, -INFINITY, INFINITY);
    wq[i] = dequant(q.q, sc, zp);
  }
}

template <int METHOD, bool RSIGN>
__global__ __launch_bounds__(kSmallThreads) void wt_small_bwd_kernel(
    const float* __restrict__ w, const float* __restrict__ G, float* __restrict__ gw,
    float* __restrict__ g_log_s, const float* __restrict__ aux, const float* __restrict__ g_lwq, int64_t n,
    const int8_t* __restrict__ r_sign, uint64_t seed, uint64_t offset, const uint64_t* __restrict__ offset_dev) {
  __shared__ double sm[4 * (kSmallThreads / 64)];
  __shared__ float bc[4];
  offset = stream_offset(offset, offset_dev);
  // the layer's sign bits: <= 512 Philox calls (64 K elements), one per thread of the first waves, instead of one per element
  constexpr bool PHILOX = (METHOD != MHAQ_FQ_LSQ) && !RSIGN;
  __shared__ __align__(16) uint32_t stile[PHILOX ? 4 * (kSmallMaxElems / 128) : 4];      // 16-byte stores (sign_tile_fill)
  if (PHILOX) {
    sign_tile_fill(stile, 0, (int)((n + 127) >> kSignsPerCallLog2), seed, offset);
    __syncthreads();
  }
  const float sc = aux[0], z = aux[1], rmx = aux[2];
  double acc[4] = {0, 0, 0, 0};   // d/ds, sum(G - gv/s), count(w == min), count(w == max)
  for (int64_t i = threadIdx.x; i < n; i += kSmallThreads) {
    const float x = w[i], g = G[i];
    QCore q = quant_core(x, sc, z, -INFINITY, INFINITY);
    const float gq = g * sc;
    const float gv = gq + noise_grad_v<METHOD>(gq, q.n, 0.f);
    const float gvs = gv / sc;
    float noise_s;
    if (METHOD == MHAQ_FQ_LSQ) {
      noise_s = gq * q.n;
    } else {
      const float r = RSIGN ? sign_half(r_sign[i]) : sign_tile_r(stile, i);
      noise_s = (MHAQ_INV_SQRT3 * gq) * r;
    }
    if (METHOD == MHAQ_FQ_STE || METHOD == MHAQ_FQ_LSQ)
      acc[0] += (double)(g * q.n + noise_s);
    else
      acc[0] += (double)((g * q.q + (-gv) * (q.v / sc)) + noise_s);
    acc[1] += (double)(g - gvs);
    acc[2] += (x == z) ? 1.0 : 0.0;
    acc[3] += (x == rmx) ? 1.0 : 0.0;
  }
  block_sum<4>(acc, sm);
  float t_local = 0.f;
  if (g_lwq) t_local = g_lwq[0] / (((rmx - z) + sc) * MHAQ_LN2F);
  if (threadIdx.x == 0) g_log_s[0] = (((float)acc[0] + t_local) * sc) * MHAQ_LN2F;
  const float gzp = block_bcast((float)acc[1] - t_local, &bc[0]);
  const float cnt = block_bcast((float)acc[2], &bc[1]);
  const float t = block_bcast(t_local, &bc[2]);
  const float cmax = block_bcast((float)acc[3], &bc[3]);
  const float tie = (gzp * 1.0f) / cnt, tie_max = (t * 1.0f) / cmax;
  for (int64_t i = threadIdx.x; i < n; i += kSmallThreads) {
    const float x = w[i];
    QCore q = quant_core(x, sc, z, -INFINITY, INFINITY);
    const float gq = G[i] * sc;
    float o = (gq + noise_grad_v<METHOD>(gq, q.n, 0.f)) / sc;
    if (x == z) o = o + tie;
    if (x == rmx) o = o + tie_max;
    gw[i] = o;
  }
}


// ------------------------------------------------------------------ PotentialLoss
// gdnsq_loss.py:47-71 / 129-153 over the concatenated regulariser vectors (<= a few thousand floats): one
// workgroup, deterministic fp64 sums, replaces ~25 scalar launches forward and ~30 backward per step.
//   out[0] ploss  [1] wloss  [2] aloss  [3] rloss  [4] cw  [5] ca  [6] cb     (cw/ca: d ploss / d hinge_i,
//   cb: d ploss / d base)   [7] -mean lws  [8] mean lwq  [9] -mean las  [10] mean laq  [11] max(lwq - lws)

__device__ inline float hinge_pow(float h, float p) { return (h > 0.f) ? ((p == 1.f) ? h : powf(h, p)) : 0.f; }
// d/dh of max(0, h)^p; torch.max(0, h) splits the gradient at the tie h == 0
__device__ inline float hinge_grad(float h, float p) {
  if (h < 0.f) return 0.f;
  const float d = (p == 1.f) ? 1.f : p * powf(h, p - 1.f);
  return (h > 0.f) ? d : 0.5f * d;
}

__global__ __launch_bounds__(kBlock) void potential_loss_fwd_kernel(
    const float* __restrict__ base, const float* __restrict__ las, const float* __restrict__ laq, int64_t na,
    const float* __restrict__ lws, const float* __restrict__ lwq, int64_t nw, float a_bits, float w_bits, float p,
    int lossless, float* __restrict__ state /* {loss_sum, cnt, t} */, int update_state,
    float* __restrict__ out) {
  __shared__ double sm[8 * 4];
  __shared__ float smax[4];
  const float wt = w_bits - 1e-3f, at = a_bits - 1e-3f;
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // sum wloss0, count w, sum aloss0, count a, sum lws, sum lwq, sum las, sum laq
  float mx = -INFINITY;
  // Four trips' loads in flight at once (a thread's 15 trips over ResNet-18's 3840 channels each waited for their own two
  // loads: 9.8 us for 30 KB); the sums run in the same order.
  constexpr int kU = 4;
  for (int64_t i0 = threadIdx.x; i0 < nw; i0 += kU * kBlock) {
    float q[kU], sv[kU];
#pragma unroll
    for (int k = 0; k < kU; ++k) {
      const int64_t i = i0 + k * kBlock, ic = i < nw ? i : i0;
      q[k] = lwq[ic];
      sv[k] = lws[ic];
    }
#pragma unroll
    for (int k = 0; k < kU; ++k) {
      if (i0 + k * kBlock < nw) {
        const float d = q[k] - sv[k];
        const float h = hinge_pow(d - wt, p);
        acc[0] += (double)h;
        acc[1] += (h > 0.f) ? 1.0 : 0.0;
        acc[4] += (double)sv[k];
        acc[5] += (double)q[k];
        mx = fmaxf(mx, d);
      }
    }
  }
  for (int64_t i0 = threadIdx.x; i0 < na; i0 += kU * kBlock) {
    float q[kU], sv[kU];
#pragma unroll
    for (int k = 0; k < kU; ++k) {
      const int64_t i = i0 + k * kBlock, ic = i < na ? i : i0;
      q[k] = laq[ic];
      sv[k] = las[ic];
    }
#pragma unroll
    for (int k = 0; k < kU; ++k) {
      if (i0 + k * kBlock < na) {
        const float h = hinge_pow((q[k] - sv[k]) - at, p);
        acc[2] += (double)h;
        acc[3] += (h > 0.f) ? 1.0 : 0.0;
        acc[6] += (double)sv[k];
        acc[7] += (double)q[k];
      }
    }
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = mx;
  block_sum<8>(acc, sm);          // contains the barriers that also publish smax
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; ++w) mx = fmaxf(mx, smax[w]);
    const float wloss = (float)acc[0] / (float)nw, aloss = (float)acc[2] / (float)na;
    const float wact = (float)acc[1], aact = (float)acc[3];
    const float b = *base;
    const float rloss = (p == 1.f) ? b : powf(b, p);
    const float loss_sum = state[0], cnt = state[1], t = state[2];
    const float calib = loss_sum / cnt;
    const float wmul = (wact + 1e-3f) / ((wact + aact) + 1e-3f);
    const float amul = (aact + 1e-3f) / ((wact + aact) + 1e-3f);
    const float l1 = lossless ? 1.0f : t, l2 = lossless ? t : 1.0f;
    out[0] = (calib * l1) * (wmul * wloss + amul * aloss) + l2 * rloss;
    out[1] = wloss; out[2] = aloss; out[3] = rloss;
    out[4] = ((calib * l1) * wmul) / (float)nw;
    out[5] = ((calib * l1) * amul) / (float)na;
    out[6] = l2 * ((p == 1.f) ? 1.f : p * powf(b, p - 1.f));
    out[7] = -(float)acc[4] / (float)nw; out[8] = (float)acc[5] / (float)nw;
    out[9] = -(float)acc[6] / (float)na; out[10] = (float)acc[7] / (float)na;
    out[11] = mx;
    if (update_state) {                                   // training mode: loss_sum += rloss.detach(); cnt += 1
      state[0] = loss_sum + rloss;
      state[1] = cnt + 1.0f;
    }
  }
}

__global__ __launch_bounds__(kBlock) void potential_loss_bwd_kernel(
    const float* __restrict__ g, const float* __restrict__ out, const float* __restrict__ las,
    const float* __restrict__ laq, int64_t na, const float* __restrict__ lws, const float* __restrict__ lwq,
    int64_t nw, float a_bits, float w_bits, float p, float* __restrict__ g_base, float* __restrict__ g_las,
    float* __restrict__ g_laq, float* __restrict__ g_lws, float* __restrict__ g_lwq) {
  const float wt = w_bits - 1e-3f, at = a_bits - 1e-3f;
  const float go = *g, cw = out[4], ca = out[5];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nw; i += (int64_t)gridDim.x * kBlock) {
    const float d = (go * cw) * hinge_grad((lwq[i] - lws[i]) - wt, p);
    g_lwq[i] = d;
    g_lws[i] = -d;
  }
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < na; i += (int64_t)gridDim.x * kBlock) {
    const float d = (go * ca) * hinge_grad((laq[i] - las[i]) - at, p);
    g_laq[i] = d;
    g_las[i] = -d;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *g_base = go * out[6];
}

static inline int threads_for_row(int64_t row, bool vec = false, size_t lds = 0) {
  if (lds > kDefaultDynLds) return 64 * kMaxWaves;   // one workgroup per CU: make it a full one
  const int64_t items = vec ? row >> 2 : row;        // accesses per row
  if (items <= 256) return 64;
  if (items <= 1024) return 128;
  return 256;
}

// Largest dynamic LDS request a workgroup may make on the current device, minus the static scratch.
static inline size_t stage_budget_bytes() {
  // a memo of one device attribute per device (a constant of the hardware, not state of the library): relaxed atomics, so
  // two threads asking at once both compute the same value and either store wins
  static std::atomic<size_t> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return kDefaultDynLds - 8192;
  size_t have = cached[dev].load(std::memory_order_relaxed);
  if (have == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0)
      v = (int)kDefaultDynLds;
    size_t b = (size_t)v - 8192;                     // static LDS: red / sm / sm4 / bc and the 4.7 KB sign tile
    const size_t cap = (size_t)kMaxStageFloats * sizeof(float);
    have = b < cap ? b : cap;
    cached[dev].store(have, std::memory_order_relaxed);
  }
  return have;
}

template <class K>
static inline int opt_in_lds(K kernel, size_t lds) {
  if (lds <= kDefaultDynLds - 8192) return 0;       // static scratch + sign tile ride on top of the dynamic request
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
}

constexpr int64_t kMultiStageFloats = 12 * 1024;   // the multi-tensor grids stage rows up to 48 KiB
// ... and run 256 threads per row -- except where a model's rows are WHOLE TENSORS (PER_TENSOR layers riding the
// model-wide launches as one channel each: ResNet-20 with `qscheme: 0`, rows up to 36,864 floats on 18 workgroups): a row
// of 8 K floats and more gets a full 1024-thread workgroup (measured on that set, tools/pc_multi_bench.py STE resnet20_pt:
// forward 21.9 -> 12.4 us, backward 37.4 -> 17.1; profiles/r04_pc_multi_pmc.txt) and, up to 36,864 floats, the
// register-resident bodies at 9 float4 per thread (11.1 -> 9.9 / 15.9 -> 12.5 us, gpurun_out/r04ac_pt_rows.txt).
// "Whole tensors" is a property of the LAUNCH, not of its longest row: every layer of such a table is one channel (total_co ==
// nlayers; up to 2 channels per layer on average still counts).  A per-channel model with ONE long-row layer -- a Linear of 8 K
// inputs and more, a 3x3 convolution on 1024 channels -- stays at 256 threads: its thousands of short rows would otherwise
// run 16 waves through two 16-wave barriers and issue 9 (18) clamped, redundant float4 loads per thread, and only its long
// rows take the unstaged body of the 256-thread grid.
// (-DMHAQ_MULTI_WIDE_ANY=1: the round-4 rule -- any launch whose longest row has 8 K floats -- as an A/B knob for
// tools/variants.sh; measured on VGG-16's convolutions + a [64, 25088] Linear, tools/pc_multi_bench.py STE vggfc:
// profiles/r05_ab_logs.txt)
#ifndef MHAQ_MULTI_WIDE_ANY
#define MHAQ_MULTI_WIDE_ANY 0
#endif
static inline int multi_threads(int64_t max_row, int64_t total_co, int nlayers) {
  return (max_row >= 8192 && (MHAQ_MULTI_WIDE_ANY || total_co <= 2 * (int64_t)nlayers)) ? 64 * kMaxWaves : kBlock;
}
// float4 per thread of the register-resident multi-tensor bodies for a model whose longest row is max_row floats
// (256 threads): 2, 4, 5 (4608-float rows: ResNet-18 / -34 / -50 3x3 layers) or 8.
// (the forward has no 5: its NV = 8 instantiation compiles to 56 VGPRs -- 8 waves per SIMD --, an NV = 5 one to 88.)
// A longest row beyond 8 float4 per thread (a per-channel model with one long-row layer, see multi_threads) keeps 8: the
// rows that fit stay single-pass, the long ones take the unstaged body of the same grid.
static inline int multi_reg_nv(int64_t max_row, bool backward) {
  const int64_t per = ((max_row + 3) / 4 + kBlock - 1) / kBlock;
  if (MHAQ_MULTI_WIDE_ANY && per > 8) return 0;      // (the round-4 rule of the A/B knob above)
  return per <= 2 ? 2 : (per <= 4 ? 4 : ((per <= 5 && backward) ? 5 : 8));
}

}  // namespace mhaq

using namespace mhaq;

template <int METHOD>
static int launch_pc_bwd(const float* w, const float* G, float* gw, float* g_s, const float* s, const float* zp,
                         int64_t co, int64_t row, const float* stats, const float* gzp_extra,
                         const int8_t* r_sign, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, hipStream_t st,
                         bool layer = false, const float* mx = nullptr, const float* g_lwq = nullptr) {
  const bool vec = vec_ok(row, w, G, gw);
  int rt = 0;
  if (const int nv = reg_plan(row, vec, &rt, true, METHOD == MHAQ_FQ_AEWGS && !stats)) {
    const bool nt = co * row * (int64_t)sizeof(float) >= kPcNtBytes;
#define MHAQ_LAUNCH_PCR_(RS, LY, NV, NT)                                                                            \
  MHAQ_LAUNCH((pc_bwd_reg_kernel<METHOD, RS, LY, NV, NT>), dim3((unsigned)co), dim3(rt), 0, st, w, G, gw, g_s, \
                     s, zp, co, row, stats, gzp_extra, r_sign, seed, offset, offset_dev, mx, g_lwq)
#define MHAQ_LAUNCH_PCR(RS, LY, NV) do { if (nt) MHAQ_LAUNCH_PCR_(RS, LY, NV, 1); else MHAQ_LAUNCH_PCR_(RS, LY, NV, MHAQ_PC_SMALL_NT); } while (0)
#define MHAQ_LAUNCH_PCR_NV(RS, LY)                                                                              \
  do { if (nv == 2) MHAQ_LAUNCH_PCR(RS, LY, 2); else if (nv == 4) MHAQ_LAUNCH_PCR(RS, LY, 4);                   \
       else MHAQ_LAUNCH_PCR(RS, LY, 8); } while (0)
    if (layer) { if (r_sign) MHAQ_LAUNCH_PCR_NV(true, true); else MHAQ_LAUNCH_PCR_NV(false, true); }
    else       { if (r_sign) MHAQ_LAUNCH_PCR_NV(true, false); else MHAQ_LAUNCH_PCR_NV(false, false); }
#undef MHAQ_LAUNCH_PCR_NV
#undef MHAQ_LAUNCH_PCR
#undef MHAQ_LAUNCH_PCR_
    return launch_status();
  }
  const bool stage = (size_t)row * 2 * sizeof(float) <= stage_budget_bytes();
  const size_t lds = stage ? (size_t)row * 2 * sizeof(float) : 0;
  const int threads = threads_for_row(row, vec, lds);
#define MHAQ_LAUNCH_PC(RS, SG, LY)                                                                                \
  do {                                                                                                            \
    auto kv = pc_bwd_kernel<METHOD, RS, SG, LY, true>;                                                            \
    auto ks = pc_bwd_kernel<METHOD, RS, SG, LY, false>;                                                           \
    if (int rc = vec ? opt_in_lds(kv, lds) : opt_in_lds(ks, lds)) return rc;                                      \
    if (vec) MHAQ_LAUNCH(kv, dim3((unsigned)co), dim3(threads), lds, st, w, G, gw, g_s, s, zp, co, row,      \
                                stats, gzp_extra, r_sign, seed, offset, offset_dev, mx, g_lwq);                               \
    else MHAQ_LAUNCH(ks, dim3((unsigned)co), dim3(threads), lds, st, w, G, gw, g_s, s, zp, co, row, stats,  \
                            gzp_extra, r_sign, seed, offset, offset_dev, mx, g_lwq);                                          \
  } while (0)
  if (layer) {
    if (r_sign) { if (stage) MHAQ_LAUNCH_PC(true, true, true); else MHAQ_LAUNCH_PC(true, false, true); }
    else        { if (stage) MHAQ_LAUNCH_PC(false, true, true); else MHAQ_LAUNCH_PC(false, false, true); }
  } else {
    if (r_sign) { if (stage) MHAQ_LAUNCH_PC(true, true, false); else MHAQ_LAUNCH_PC(true, false, false); }
    else        { if (stage) MHAQ_LAUNCH_PC(false, true, false); else MHAQ_LAUNCH_PC(false, false, false); }
  }
#undef MHAQ_LAUNCH_PC
  return launch_status();
}

static_assert(sizeof(WLayerDesc) == sizeof(mhaq_wlayer_desc), "descriptor layout must match the C header");

template <int METHOD>
static int launch_pc_bwd_multi(const WLayerDesc* d, int nlayers, const float* aux_all, int64_t total_co,
                               int64_t aux_stride, int64_t max_row, float* gw_all, float* g_log_s_all,
                               const float* stats_all, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                               hipStream_t st) {
  const bool stage = 2 * max_row <= kMultiStageFloats;
  const size_t lds = stage ? (size_t)max_row * 2 * sizeof(float) : 0;
  // the statistics slab is [3][total_co] of THIS launch (a group's own), the aux slab may be a window of a wider one
  const int threads = multi_threads(max_row, total_co, nlayers);
  const int nv = MHAQ_MULTI_REG ? multi_reg_nv(max_row, true) : 0;
  // (no dynamic LDS: the few rows of such a launch that are not whole aligned float4s -- a first convolution's 27-float
  // rows -- take the unstaged body and read their row a second time from L2.  The staged fallback had every workgroup of the
  // launch reserve 2 x max_row floats it never touched: 37 KB for ResNet-18's last group, 3 workgroups per CU instead of
  // the 6 its registers allow; tools/pc_multi_bench.py MHAQ_PCMB_TRACE=1)
#define MHAQ_LAUNCH_MBR(NV, TB)                                                                                       \
  MHAQ_LAUNCH((pc_bwd_multi_reg_kernel<METHOD, NV, TB>), dim3((unsigned)total_co), dim3(TB), 0, st, d,         \
                     nlayers, aux_all, aux_stride, gw_all, g_log_s_all, stats_all, total_co, seed, offset, offset_dev)
  // (128-thread workgroups, which the forward takes, measured slower here: ResNet-18 groups 13.5 / 13.6 / 7.8 -> 15.2 / 15.4 /
  // 8.6 us cold at 109 VGPRs and 9 float4 of W and of G per thread; gpurun_out/r04f_whatif.txt)
  if (nv && threads == kBlock) {
    if (nv == 2) MHAQ_LAUNCH_MBR(2, kBlock); else if (nv == 4) MHAQ_LAUNCH_MBR(4, kBlock);
    else if (nv == 5) MHAQ_LAUNCH_MBR(5, kBlock); else MHAQ_LAUNCH_MBR(8, kBlock);
    return launch_status();
  }
  // whole-tensor rows (1024 threads): rows up to 36,864 floats keep their data in registers (9 float4 of W and of G per
  // thread); longer ones, odd lengths and unaligned tensors take the unstaged body of the same grid
  if (MHAQ_MULTI_REG && threads != kBlock && (max_row + 3) / 4 <= 9 * (int64_t)(64 * kMaxWaves)) {
    MHAQ_LAUNCH_MBR(9, 64 * kMaxWaves);
    return launch_status();
  }
#undef MHAQ_LAUNCH_MBR
  if (stage) MHAQ_LAUNCH((pc_bwd_multi_kernel<METHOD, true>), dim3((unsigned)total_co), dim3(threads), lds, st, d, nlayers, aux_all, aux_stride, gw_all, g_log_s_all, stats_all, total_co, seed, offset, offset_dev);
  else MHAQ_LAUNCH((pc_bwd_multi_kernel<METHOD, false>), dim3((unsigned)total_co), dim3(threads), 0, st, d, nlayers, aux_all, aux_stride, gw_all, g_log_s_all, stats_all, total_co, seed, offset, offset_dev);
  return launch_status();
}

static int dispatch_pc_bwd_multi(const WLayerDesc* d, int nlayers, const float* aux_all, int64_t total_co,
                                 int64_t aux_stride, int64_t max_row, float* gw_all, float* g_log_s_all, int method,
                                 const float* stats_all, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                                 hipStream_t st) {
  switch (method) {
    case MHAQ_FQ_STE: return launch_pc_bwd_multi<MHAQ_FQ_STE>(d, nlayers, aux_all, total_co, aux_stride, max_row, gw_all, g_log_s_all, stats_all, seed, offset, offset_dev, st);
    case MHAQ_FQ_EWGS: return launch_pc_bwd_multi<MHAQ_FQ_EWGS>(d, nlayers, aux_all, total_co, aux_stride, max_row, gw_all, g_log_s_all, stats_all, seed, offset, offset_dev, st);
    case MHAQ_FQ_AEWGS: return launch_pc_bwd_multi<MHAQ_FQ_AEWGS>(d, nlayers, aux_all, total_co, aux_stride, max_row, gw_all, g_log_s_all, stats_all, seed, offset, offset_dev, st);
    default: return launch_pc_bwd_multi<MHAQ_FQ_LSQ>(d, nlayers, aux_all, total_co, aux_stride, max_row, gw_all, g_log_s_all, stats_all, seed, offset, offset_dev, st);
  }
}


extern "C" {

static int launch_pc_fwd(const float* w, float* wq, float* zp_out, float* q_out, const float* s, int64_t co,
                         int64_t row, bool layer, float* s_out, float* mx_out, float* lwq_out, hipStream_t st) {
  const bool vec = vec_ok(row, w, wq, q_out);
  int rt = 0;
  if (const int nv = reg_plan(row, vec, &rt)) {
    const bool nt = co * row * (int64_t)sizeof(float) >= kPcNtBytes;
#define MHAQ_LAUNCH_PCFR_(WQ, LY, NV, NT)                                                                        \
  MHAQ_LAUNCH((pc_fwd_reg_kernel<WQ, LY, NV, NT>), dim3((unsigned)co), dim3(rt), 0, st, w, wq, zp_out, q_out, \
                     s, row, s_out, mx_out, lwq_out)
#define MHAQ_LAUNCH_PCFR(WQ, LY, NV) do { if (nt) MHAQ_LAUNCH_PCFR_(WQ, LY, NV, 1); else MHAQ_LAUNCH_PCFR_(WQ, LY, NV, MHAQ_PC_SMALL_NT); } while (0)
#define MHAQ_LAUNCH_PCFR_NV(WQ, LY)                                                                          \
  do { if (nv == 2) MHAQ_LAUNCH_PCFR(WQ, LY, 2); else if (nv == 4) MHAQ_LAUNCH_PCFR(WQ, LY, 4);              \
       else MHAQ_LAUNCH_PCFR(WQ, LY, 8); } while (0)
    if (layer) MHAQ_LAUNCH_PCFR_NV(false, true);
    else if (q_out) MHAQ_LAUNCH_PCFR_NV(true, false);
    else MHAQ_LAUNCH_PCFR_NV(false, false);
#undef MHAQ_LAUNCH_PCFR_NV
#undef MHAQ_LAUNCH_PCFR
#undef MHAQ_LAUNCH_PCFR_
    return launch_status();
  }
  const bool stage = (size_t)row * sizeof(float) <= stage_budget_bytes();
  const size_t lds = stage ? (size_t)row * sizeof(float) : 0;
  const int threads = threads_for_row(row, vec, lds);
#define MHAQ_LAUNCH_PCF(SG, WQ, LY)                                                                            \
  do {                                                                                                         \
    auto kv = pc_fwd_kernel<SG, WQ, LY, true>;                                                                 \
    auto ks = pc_fwd_kernel<SG, WQ, LY, false>;                                                                \
    if (int rc = vec ? opt_in_lds(kv, lds) : opt_in_lds(ks, lds)) return rc;                                   \
    if (vec) MHAQ_LAUNCH(kv, dim3((unsigned)co), dim3(threads), lds, st, w, wq, zp_out, q_out, s, row,   \
                                s_out, mx_out, lwq_out);                                                       \
    else MHAQ_LAUNCH(ks, dim3((unsigned)co), dim3(threads), lds, st, w, wq, zp_out, q_out, s, row,       \
                            s_out, mx_out, lwq_out);                                                           \
  } while (0)
  if (layer) {
    if (stage) MHAQ_LAUNCH_PCF(true, false, true); else MHAQ_LAUNCH_PCF(false, false, true);
  } else if (stage) {
    if (q_out) MHAQ_LAUNCH_PCF(true, true, false); else MHAQ_LAUNCH_PCF(true, false, false);
  } else {
    if (q_out) MHAQ_LAUNCH_PCF(false, true, false); else MHAQ_LAUNCH_PCF(false, false, false);
  }
#undef MHAQ_LAUNCH_PCF
  return launch_status();
}

int mhaq_fq_pc_fwd(const float* w, float* wq, float* zp_out, float* q_out, const float* s, int64_t co,
                   int64_t row, void* stream) {
  if (co < 0 || row <= 0 || !s || !zp_out || (co > 0 && (!w || !wq))) return MHAQ_FQ_EINVAL;
  if (co == 0) return 0;
  if (co > 0x7fffffff) return MHAQ_FQ_EUNSUPPORTED;
  return launch_pc_fwd(w, wq, zp_out, q_out, s, co, row, false, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

int mhaq_fq_wlayer_fwd(const float* w, float* wq, const float* log_s, int64_t co, int64_t row, float* s_out,
                       float* zp_out, float* mx_out, float* lwq_out, void* stream) {
  if (co < 0 || row <= 0 || !log_s || !s_out || !zp_out || !mx_out || !lwq_out || (co > 0 && (!w || !wq)))
    return MHAQ_FQ_EINVAL;
  if (co == 0) return 0;
  if (co > 0x7fffffff) return MHAQ_FQ_EUNSUPPORTED;
  return launch_pc_fwd(w, wq, zp_out, nullptr, log_s, co, row, true, s_out, mx_out, lwq_out, (hipStream_t)stream);
}

int mhaq_fq_pc_aewgs_stats(const float* w, const float* G, const float* s, const float* zp, int64_t co,
                           int64_t row, float* stats, void* stream) {
  if (co <= 0 || row <= 0 || !w || !G || !s || !zp || !stats) return MHAQ_FQ_EINVAL;
  if (co > 0x7fffffff) return MHAQ_FQ_EUNSUPPORTED;
  if (vec_ok(row, w, G))
    MHAQ_LAUNCH(pc_aewgs_stats_kernel<true>, dim3((unsigned)co), dim3(threads_for_row(row, true)), 0,
                       (hipStream_t)stream, w, G, s, zp, co, row, stats);
  else
    MHAQ_LAUNCH(pc_aewgs_stats_kernel<false>, dim3((unsigned)co), dim3(threads_for_row(row)), 0,
                       (hipStream_t)stream, w, G, s, zp, co, row, stats);
  return launch_status();
}


int mhaq_fq_row_minmax(const float* w, int64_t co, int64_t row, float* mn_out, float* mx_out, void* stream) {
  if (co < 0 || row <= 0 || (co > 0 && (!w || !mn_out || !mx_out))) return MHAQ_FQ_EINVAL;
  if (co == 0) return 0;
  if (co > 0x7fffffff) return MHAQ_FQ_EUNSUPPORTED;
  if (vec_ok(row, w, w))
    MHAQ_LAUNCH(row_minmax_kernel<true>, dim3((unsigned)co), dim3(threads_for_row(row, true)), 0,
                       (hipStream_t)stream, w, row, mn_out, mx_out);
  else
    MHAQ_LAUNCH(row_minmax_kernel<false>, dim3((unsigned)co), dim3(threads_for_row(row)), 0,
                       (hipStream_t)stream, w, row, mn_out, mx_out);
  return launch_status();
}

int mhaq_fq_pc_bwd(const float* w, const float* G, float* gw, float* g_s, const float* s, const float* zp,
                   int64_t co, int64_t row, int method, const float* stats, const float* gzp_extra,
                   const int8_t* r_sign, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, void* stream) {
  if (co < 0 || row <= 0 || !s || !zp || !g_s || (co > 0 && (!w || !G || !gw))) return MHAQ_FQ_EINVAL;
  if (method < 0 || method > 3) return MHAQ_FQ_EINVAL;
  if (co == 0) return 0;
  if (co > 0x7fffffff) return MHAQ_FQ_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  switch (method) {
    case MHAQ_FQ_STE: return launch_pc_bwd<MHAQ_FQ_STE>(w, G, gw, g_s, s, zp, co, row, stats, gzp_extra, r_sign, seed, offset, offset_dev, st);
    case MHAQ_FQ_EWGS: return launch_pc_bwd<MHAQ_FQ_EWGS>(w, G, gw, g_s, s, zp, co, row, stats, gzp_extra, r_sign, seed, offset, offset_dev, st);
    case MHAQ_FQ_AEWGS: return launch_pc_bwd<MHAQ_FQ_AEWGS>(w, G, gw, g_s, s, zp, co, row, stats, gzp_extra, r_sign, seed, offset, offset_dev, st);
    default: return launch_pc_bwd<MHAQ_FQ_LSQ>(w, G, gw, g_s, s, zp, co, row, stats, gzp_extra, r_sign, seed, offset, offset_dev, st);
  }
}

int mhaq_fq_wlayer_bwd(const float* w, const float* G, float* gw, float* g_log_s, const float* s,
                       const float* zp, const float* mx, const float* g_lwq, int64_t co, int64_t row, int method,
                       const float* stats, const float* gzp_extra, const int8_t* r_sign, uint64_t seed,
                       uint64_t offset, const uint64_t* offset_dev, void* stream) {
  if (co < 0 || row <= 0 || !s || !zp || !mx || !g_log_s || (co > 0 && (!w || !G || !gw))) return MHAQ_FQ_EINVAL;
  if (method < 0 || method > 3) return MHAQ_FQ_EINVAL;
  if (co == 0) return 0;
  if (co > 0x7fffffff) return MHAQ_FQ_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  switch (method) {
    case MHAQ_FQ_STE: return launch_pc_bwd<MHAQ_FQ_STE>(w, G, gw, g_log_s, s, zp, co, row, stats, gzp_extra, r_sign, seed, offset, offset_dev, st, true, mx, g_lwq);
    case MHAQ_FQ_EWGS: return launch_pc_bwd<MHAQ_FQ_EWGS>(w, G, gw, g_log_s, s, zp, co, row, stats, gzp_extra, r_sign, seed, offset, offset_dev, st, true, mx, g_lwq);
    case MHAQ_FQ_AEWGS: return launch_pc_bwd<MHAQ_FQ_AEWGS>(w, G, gw, g_log_s, s, zp, co, row, stats, gzp_extra, r_sign, seed, offset, offset_dev, st, true, mx, g_lwq);
    default: return launch_pc_bwd<MHAQ_FQ_LSQ>(w, G, gw, g_log_s, s, zp, co, row, stats, gzp_extra, r_sign, seed, offset, offset_dev, st, true, mx, g_lwq);
  }
}

int mhaq_fq_wlayer_fwd_multi(const mhaq_wlayer_desc* descs_device, int nlayers, int64_t total_co, int64_t max_row,
                             float* wq_all, float* aux_all, void* stream) {
  if (nlayers <= 0 || total_co <= 0 || max_row <= 0 || !descs_device || !wq_all || !aux_all) return MHAQ_FQ_EINVAL;
  if (total_co > 0x7fffffff) return MHAQ_FQ_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const WLayerDesc* d = reinterpret_cast<const WLayerDesc*>(descs_device);
  const bool stage = max_row <= kMultiStageFloats;
  const size_t lds = stage ? (size_t)max_row * sizeof(float) : 0;
  const int threads = multi_threads(max_row, total_co, nlayers);
  const int nv = MHAQ_MULTI_REG ? multi_reg_nv(max_row, false) : 0;
  // (no dynamic LDS, unstaged fallback for odd rows: see launch_pc_bwd_multi)
#define MHAQ_LAUNCH_MFR(NV, TB)                                                                                      \
  MHAQ_LAUNCH((pc_fwd_multi_reg_kernel<NV, TB>), dim3((unsigned)total_co), dim3(TB), 0, st, d, nlayers,        \
                     wq_all, aux_all, total_co)
  // Rows up to 4608 floats (every 3x3 layer up to 512 input channels): 128-thread workgroups, up to 9 float4 per thread.  The
  // launch is bound by the latency of a row's round trips (see MHAQ_FWD_MULTI_MAXW), and at the same registers -- the same
  // bytes in flight -- per CU, sixteen two-wave rows overlap their phases better than eight four-wave ones: ResNet-18 forward
  // 21.9 -> 20.7 us cold, 17.2 -> 15.1 warm (one wave per row, 18 float4 per thread, 4 waves per SIMD: 22.1 / 16.5).
  const int64_t per128 = ((max_row + 3) / 4 + 127) / 128;
  if (MHAQ_FWD_MULTI_TB128 && nv && threads == kBlock && per128 <= 9) {
    if (per128 <= 2) MHAQ_LAUNCH_MFR(2, 128); else if (per128 <= 4) MHAQ_LAUNCH_MFR(4, 128); else MHAQ_LAUNCH_MFR(9, 128);
    return launch_status();
  }
  if (nv && threads == kBlock) {
    if (nv == 2) MHAQ_LAUNCH_MFR(2, kBlock); else if (nv == 4) MHAQ_LAUNCH_MFR(4, kBlock); else MHAQ_LAUNCH_MFR(8, kBlock);
    return launch_status();
  }
  if (MHAQ_MULTI_REG && threads != kBlock && (max_row + 3) / 4 <= 9 * (int64_t)(64 * kMaxWaves)) {   // whole-tensor rows
    MHAQ_LAUNCH_MFR(9, 64 * kMaxWaves);
    return launch_status();
  }
#undef MHAQ_LAUNCH_MFR
  if (stage) MHAQ_LAUNCH((pc_fwd_multi_kernel<true>), dim3((unsigned)total_co), dim3(threads), lds, st, d, nlayers, wq_all, aux_all, total_co);
  else MHAQ_LAUNCH((pc_fwd_multi_kernel<false>), dim3((unsigned)total_co), dim3(threads), 0, st, d, nlayers, wq_all, aux_all, total_co);
  return launch_status();
}

int mhaq_fq_wlayer_bwd_multi(const mhaq_wlayer_desc* descs_device, int nlayers, int64_t total_co, int64_t max_row,
                             const float* aux_all, float* gw_all, float* g_log_s_all, int method,
                             const float* stats_all, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, void* stream) {
  if (nlayers <= 0 || total_co <= 0 || max_row <= 0 || !descs_device || !aux_all || !gw_all || !g_log_s_all)
    return MHAQ_FQ_EINVAL;
  if (method < 0 || method > 3) return MHAQ_FQ_EINVAL;
  if (total_co > 0x7fffffff) return MHAQ_FQ_EUNSUPPORTED;
  return dispatch_pc_bwd_multi(reinterpret_cast<const WLayerDesc*>(descs_device), nlayers, aux_all, total_co, total_co,
                               max_row, gw_all, g_log_s_all, method, stats_all, seed, offset, offset_dev,
                               (hipStream_t)stream);
}

int mhaq_fq_wlayer_bwd_group(const mhaq_wlayer_desc* descs_device, int nlayers, int64_t group_co, int64_t max_row,
                             const float* aux, int64_t aux_stride, float* gw, float* g_log_s, int method,
                             const float* stats, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                             void* stream) {
  if (nlayers <= 0 || group_co <= 0 || max_row <= 0 || aux_stride < group_co || !descs_device || !aux || !gw ||
      !g_log_s)
    return MHAQ_FQ_EINVAL;
  if (method < 0 || method > 3) return MHAQ_FQ_EINVAL;
  if (group_co > 0x7fffffff) return MHAQ_FQ_EUNSUPPORTED;
  return dispatch_pc_bwd_multi(reinterpret_cast<const WLayerDesc*>(descs_device), nlayers, aux, group_co, aux_stride,
                               max_row, gw, g_log_s, method, stats, seed, offset, offset_dev, (hipStream_t)stream);
}

int mhaq_fq_wlayer_aewgs_stats_group(const mhaq_wlayer_desc* descs_device, int nlayers, int64_t group_co,
                                     const float* aux, int64_t aux_stride, float* stats, void* stream) {
  if (nlayers <= 0 || group_co <= 0 || aux_stride < group_co || !descs_device || !aux || !stats)
    return MHAQ_FQ_EINVAL;
  if (group_co > 0x7fffffff) return MHAQ_FQ_EUNSUPPORTED;
  MHAQ_LAUNCH(pc_aewgs_stats_multi_kernel, dim3((unsigned)group_co), dim3(kBlock), 0, (hipStream_t)stream,
                     reinterpret_cast<const WLayerDesc*>(descs_device), nlayers, aux, aux_stride, stats, group_co);
  return launch_status();
}

int mhaq_fq_pc_quantize(const float* x, float* q_out, float* y_out, const float* s, const float* zp, int64_t co,
                        int64_t row, int32_t* flags, void* stream) {
  if (co < 0 || row < 0 || !s || !zp || ((co > 0 && row > 0) && (!x || !q_out))) return MHAQ_FQ_EINVAL;
  if (co == 0 || row == 0) return 0;
  const int64_t chunks = (row + kBlock * 4 - 1) / (kBlock * 4);
  if (chunks > 0x7fffffff || co * chunks > (int64_t)(0xffffffffull / kBlock)) return MHAQ_FQ_EUNSUPPORTED;
  MHAQ_LAUNCH(pc_quantize_kernel, dim3((unsigned)(co * chunks)), dim3(kBlock), 0, (hipStream_t)stream, x, q_out, y_out, s, zp,
              row, (int)chunks, flags);
  return launch_status();
}

int mhaq_fq_vec_fwd(const float* x, float* y, float* q_out, const float* s, const float* zp, int64_t n,
                    void* stream) {
  if (n < 0 || (n > 0 && (!x || !y || !s || !zp))) return MHAQ_FQ_EINVAL;
  if (n == 0) return 0;
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b > kMaxBlocks) b = kMaxBlocks;
  if (q_out) MHAQ_LAUNCH((vec_fwd_kernel<true>), dim3((unsigned)b), dim3(kBlock), 0, (hipStream_t)stream, x, y, q_out, s, zp, n);
  else MHAQ_LAUNCH((vec_fwd_kernel<false>), dim3((unsigned)b), dim3(kBlock), 0, (hipStream_t)stream, x, y, q_out, s, zp, n);
  return launch_status();
}

int mhaq_fq_vec_aewgs_stats(const float* x, const float* g, const float* s, const float* zp, int64_t n,
                            float* stats, void* stream) {
  if (n <= 0 || !x || !g || !s || !zp || !stats) return MHAQ_FQ_EINVAL;
  MHAQ_LAUNCH(vec_aewgs_stats_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, x, g, s, zp, n, stats);
  return launch_status();
}

int mhaq_fq_vec_bwd(const float* x, const float* g, float* gx, float* g_s, float* g_zp, const float* s,
                    const float* zp, int64_t n, int method, const float* stats, const int8_t* r_sign,
                    uint64_t seed, uint64_t offset, const uint64_t* offset_dev, void* stream) {
  if (n < 0 || (n > 0 && (!x || !g || !gx || !g_s || !g_zp || !s || !zp))) return MHAQ_FQ_EINVAL;
  if (method < 0 || method > 3) return MHAQ_FQ_EINVAL;
  if (method == MHAQ_FQ_AEWGS && !stats) return MHAQ_FQ_EINVAL;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  switch (method) {
    case MHAQ_FQ_STE: return launch_vec_bwd<MHAQ_FQ_STE>(x, g, gx, g_s, g_zp, s, zp, n, stats, r_sign, seed, offset, offset_dev, st);
    case MHAQ_FQ_EWGS: return launch_vec_bwd<MHAQ_FQ_EWGS>(x, g, gx, g_s, g_zp, s, zp, n, stats, r_sign, seed, offset, offset_dev, st);
    case MHAQ_FQ_AEWGS: return launch_vec_bwd<MHAQ_FQ_AEWGS>(x, g, gx, g_s, g_zp, s, zp, n, stats, r_sign, seed, offset, offset_dev, st);
    default: return launch_vec_bwd<MHAQ_FQ_LSQ>(x, g, gx, g_s, g_zp, s, zp, n, stats, r_sign, seed, offset, offset_dev, st);
  }
}

int mhaq_fq_noise_fwd(const float* v, float* out, int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (!v || !out))) return MHAQ_FQ_EINVAL;
  if (n == 0) return 0;
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b > kMaxBlocks) b = kMaxBlocks;
  MHAQ_LAUNCH(noise_fwd_kernel, dim3((unsigned)b), dim3(kBlock), 0, (hipStream_t)stream, v, out, n);
  return launch_status();
}

size_t mhaq_fq_noise_bwd_workspace_bytes(int64_t groups, int64_t len) {
  if (groups <= 0 || len <= 0) return sizeof(double);
  return (size_t)groups * noise_slices(groups, len) * sizeof(double);
}

int mhaq_fq_noise_bwd(const float* v, const float* g, float* gv, float* gs, int64_t groups, int64_t len,
                      int method, const float* stats, int64_t period, const int8_t* r_sign, uint64_t seed,
                      uint64_t offset, const uint64_t* offset_dev, void* workspace, size_t workspace_bytes, void* stream) {
  if (groups <= 0 || len <= 0 || !v || !g || !gv || !gs) return MHAQ_FQ_EINVAL;
  if (method < 0 || method > 3) return MHAQ_FQ_EINVAL;
  // one 256-thread workgroup per group on grid.x, here and in the finalize: HIP validates gridDim.x * blockDim.x <= UINT32_MAX,
  // i.e. at most 2^24 - 1 groups -- beyond that this is an argument the library does not serve, not a launch error
  if (groups > (int64_t)(0xffffffffull / kBlock)) return MHAQ_FQ_EUNSUPPORTED;
  if (method == MHAQ_FQ_AEWGS && !stats) return MHAQ_FQ_EINVAL;
  if (!workspace || workspace_bytes < mhaq_fq_noise_bwd_workspace_bytes(groups, len)) return MHAQ_FQ_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int slices = noise_slices(groups, len);
  double* partial = (double*)workspace;
  int rc;
  switch (method) {
    case MHAQ_FQ_STE: rc = launch_noise_bwd<MHAQ_FQ_STE>(v, g, gv, groups, len, stats, period, r_sign, seed, offset, offset_dev, partial, slices, st); break;
    case MHAQ_FQ_EWGS: rc = launch_noise_bwd<MHAQ_FQ_EWGS>(v, g, gv, groups, len, stats, period, r_sign, seed, offset, offset_dev, partial, slices, st); break;
    case MHAQ_FQ_AEWGS: rc = launch_noise_bwd<MHAQ_FQ_AEWGS>(v, g, gv, groups, len, stats, period, r_sign, seed, offset, offset_dev, partial, slices, st); break;
    default: rc = launch_noise_bwd<MHAQ_FQ_LSQ>(v, g, gv, groups, len, stats, period, r_sign, seed, offset, offset_dev, partial, slices, st); break;
  }
  if (rc) return rc;
  MHAQ_LAUNCH(noise_bwd_finalize_kernel, dim3((unsigned)groups), dim3(kBlock), 0, st, partial, slices, gs);
  return launch_status();
}

int64_t mhaq_fq_wlayer_pt_max_elements(void) { return kSmallMaxElems; }

int mhaq_fq_wlayer_pt_fwd(const float* w, float* wq, const float* log_s, int64_t n, float* aux, void* stream) {
  if (n <= 0 || n > kSmallMaxElems) return n <= 0 ? MHAQ_FQ_EINVAL : MHAQ_FQ_EUNSUPPORTED;
  if (!w || !wq || !log_s || !aux) return MHAQ_FQ_EINVAL;
  MHAQ_LAUNCH(wt_small_fwd_kernel, dim3(1), dim3(kSmallThreads), 0, (hipStream_t)stream, w, wq, log_s, n, aux);
  return launch_status();
}

int mhaq_fq_wlayer_pt_bwd(const float* w, const float* G, float* gw, float* g_log_s, const float* aux,
                          const float* g_lwq, int64_t n, int method, const int8_t* r_sign, uint64_t seed,
                          uint64_t offset, const uint64_t* offset_dev, void* stream) {
  if (n <= 0 || n > kSmallMaxElems) return n <= 0 ? MHAQ_FQ_EINVAL : MHAQ_FQ_EUNSUPPORTED;
  if (!w || !G || !gw || !g_log_s || !aux) return MHAQ_FQ_EINVAL;
  if (method == MHAQ_FQ_AEWGS) return MHAQ_FQ_EUNSUPPORTED;
  if (method < 0 || method > 3) return MHAQ_FQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
#define MHAQ_LAUNCH_WS(M)                                                                                       \
  do {                                                                                                          \
    if (r_sign) MHAQ_LAUNCH((wt_small_bwd_kernel<M, true>), dim3(1), dim3(kSmallThreads), 0, st, w, G, gw, \
                                   g_log_s, aux, g_lwq, n, r_sign, seed, offset, offset_dev);                                \
    else MHAQ_LAUNCH((wt_small_bwd_kernel<M, false>), dim3(1), dim3(kSmallThreads), 0, st, w, G, gw,       \
                            g_log_s, aux, g_lwq, n, r_sign, seed, offset, offset_dev);                                       \
  } while (0)
  switch (method) {
    case MHAQ_FQ_STE: MHAQ_LAUNCH_WS(MHAQ_FQ_STE); break;
    case MHAQ_FQ_EWGS: MHAQ_LAUNCH_WS(MHAQ_FQ_EWGS); break;
    default: MHAQ_LAUNCH_WS(MHAQ_FQ_LSQ); break;
  }
#undef MHAQ_LAUNCH_WS
  return launch_status();
}

int mhaq_fq_potential_loss_fwd(const float* base, const float* las, const float* laq, int64_t na, const float* lws,
                               const float* lwq, int64_t nw, float a_bits, float w_bits, float p, int lossless,
                               float* state, int update_state, float* out, void* stream) {
  if (na <= 0 || nw <= 0 || !base || !las || !laq || !lws || !lwq || !state || !out) return MHAQ_FQ_EINVAL;
  MHAQ_LAUNCH(potential_loss_fwd_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, base, las, laq, na,
                     lws, lwq, nw, a_bits, w_bits, p, lossless, state, update_state, out);
  return launch_status();
}

int mhaq_fq_potential_loss_bwd(const float* g, const float* out, const float* las, const float* laq, int64_t na,
                               const float* lws, const float* lwq, int64_t nw, float a_bits, float w_bits, float p,
                               float* g_base, float* g_las, float* g_laq, float* g_lws, float* g_lwq,
                               void* stream) {
  if (na <= 0 || nw <= 0 || !g || !out || !las || !laq || !lws || !lwq || !g_base || !g_las || !g_laq || !g_lws ||
      !g_lwq)
    return MHAQ_FQ_EINVAL;
  int64_t b = ((na > nw ? na : nw) + kBlock - 1) / kBlock;
  if (b > 64) b = 64;
  MHAQ_LAUNCH(potential_loss_bwd_kernel, dim3((unsigned)b), dim3(kBlock), 0, (hipStream_t)stream, g, out, las,
                     laq, na, lws, lwq, nw, a_bits, w_bits, p, g_base, g_las, g_laq, g_lws, g_lwq);
  return launch_status();
}

#ifdef MHAQ_TRACE
// trace builds only (see MHAQ_TRACE_AT): copies the stamps of the last multi-tensor launch to the host and clears them
int mhaq_debug_trace_read(unsigned long long* host, int nblocks) {
  if (nblocks > kTraceBlocks) nblocks = kTraceBlocks;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(mhaq_trace_buf), (size_t)nblocks * 8 * sizeof(unsigned long long)) != hipSuccess)
    return 1;
  static unsigned long long zeros[8 * kTraceBlocks];
  return hipMemcpyToSymbol(HIP_SYMBOL(mhaq_trace_buf), zeros, sizeof(zeros)) == hipSuccess ? 0 : 1;
}
#endif
}  // extern "C"
