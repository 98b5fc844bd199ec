// Shared device helpers for the gfx950 fake-quant kernels (wave64, CDNA4).
// Compiled with -ffp-contract=off: every fp32 op below is a separately rounded
// IEEE operation so elementwise results are bit-identical to the reference's
// eager op chain (gdnsq.py:189-229).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mhaq_fq.h"

namespace mhaq {

constexpr int kWave = 64;
constexpr int kBlock = 256;              // 4 waves: one per SIMD
constexpr int kMaxBlocks = 256 * 8;      // 256 CUs x 8 resident 256-thread blocks

// fp32(3^-1/2): the reference multiplies an fp32 tensor by the python double 3.0**-0.5,
// which aten rounds to fp32 first (gdnsq.py:55).
#define MHAQ_INV_SQRT3 0.57735026918962584f

// ---------------------------------------------------------------- Philox4x32-10
struct Philox4 { uint32_t w[4]; };

__host__ __device__ inline Philox4 philox4x32_10(uint64_t ctr01, uint64_t ctr23, uint64_t key) {
  uint32_t c0 = (uint32_t)ctr01, c1 = (uint32_t)(ctr01 >> 32);
  uint32_t c2 = (uint32_t)ctr23, c3 = (uint32_t)(ctr23 >> 32);
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{{c0, c1, c2, c3}};
}

// Sign stream layout, ABI v3 (see include/mhaq_fq.h): ONE Philox call yields the signs of 128 CONSECUTIVE elements --
// all 128 bits of its output are spent.  Element i of a stream (seed, offset):
//   call c = i >> 7;  counter = {lo(c), hi(c), lo(offset), hi(offset)}, key = {lo(seed), hi(seed)};
//   bit j = i & 127 of the 128 output bits = bit (j & 31) of output word (j >> 5);  r = bit ? +0.5 : -0.5.
// A pure function of (seed, offset, i), independent of the launch geometry.  (v2 drew one call per lane and kept 8 of its
// bits -- 4 on 64-thread rows --, which made the per-channel AEWGS / STE backward VALU-bound: a Philox call is ~40
// quarter-rate integer multiplies.)  A workgroup computes the calls its elements need ONCE, cooperatively, into an LDS
// tile (sign_tile_fill) and every lane then picks its nibbles out of it: the streaming backward runs 16 calls per
// 2048-element block in ONE wave instead of one call per lane in all four.
constexpr int kSignsPerCallLog2 = 7;
__host__ __device__ inline Philox4 philox_call(int64_t call, uint64_t seed, uint64_t offset) {
  return philox4x32_10((uint64_t)call, offset, seed);
}
__host__ __device__ inline uint32_t philox_word_of(const Philox4& p, int w) {
  return w == 0 ? p.w[0] : (w == 1 ? p.w[1] : (w == 2 ? p.w[2] : p.w[3]));
}
// one element (tails, unaligned views, the small latency-bound kernels): a whole call for one bit
__host__ __device__ inline float philox_r(int64_t i, uint64_t seed, uint64_t offset) {
  const Philox4 p = philox_call(i >> kSignsPerCallLog2, seed, offset);
  const int j = (int)(i & 127);
  return ((philox_word_of(p, j >> 5) >> (j & 31)) & 1u) ? 0.5f : -0.5f;
}
// the four elements i0 .. i0+3 (i0 % 4 == 0: they share a word): bit k of the result = sign bit of element i0 + k
__host__ __device__ inline uint32_t philox_nibble(int64_t i0, uint64_t seed, uint64_t offset) {
  const Philox4 p = philox_call(i0 >> kSignsPerCallLog2, seed, offset);
  const int j = (int)(i0 & 127);
  return (philox_word_of(p, j >> 5) >> (j & 31)) & 15u;
}

// The effective stream offset of a backward launch: the host argument plus, when the caller keeps a device-resident
// base (nullable `offset_dev`, include/mhaq_fq.h), the 64-bit word it points at.  A captured hipGraph freezes the
// host argument; the word in memory is read at every replay, so a caller that advances it between replays (one
// 8-byte add per step) draws fresh signs each time.  Wave-uniform: one s_load_dwordx2.
__device__ inline uint64_t stream_offset(uint64_t offset, const uint64_t* __restrict__ offset_dev) {
  return offset_dev ? offset + *offset_dev : offset;
}

// explicit signs (r_sign, int8): any positive value is +0.5, zero or negative is -0.5 -- so both a +-1 coding
// (mhaq_fq_fill_r, the golden vectors) and a 0/1 coding (torch.randint(0, 2), one launch) are accepted
__host__ __device__ inline float sign_half(int8_t v) { return v > 0 ? 0.5f : -0.5f; }

// ---- LDS sign tile: the sign bits of the stream elements [128 * c0, 128 * (c0 + ncalls)), 4 words per call.
// Filled by the whole workgroup (thread t computes call c0 + t, c0 + t + blockDim, ...: only the waves that hold a call
// run the Philox rounds at all); the caller places ONE barrier between the fill and the first read.  `tile` must be
// 16-byte aligned (every caller declares it __align__(16) / alignas(16)): a call's four words go out as one ds_write_b128.
__device__ inline void sign_tile_fill(uint32_t* __restrict__ tile, int64_t c0, int ncalls, uint64_t seed,
                                      uint64_t offset) {
  for (int t = threadIdx.x; t < ncalls; t += blockDim.x) {
    const Philox4 p = philox_call(c0 + t, seed, offset);
    typedef uint32_t vu4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<vu4*>(tile + 4 * t) = vu4{p.w[0], p.w[1], p.w[2], p.w[3]};
  }
}
// rel = element index relative to the tile's first element (128 * c0); rel % 4 == 0 for the nibble form
__device__ inline uint32_t sign_tile_nibble(const uint32_t* __restrict__ tile, int64_t rel) {
  return (tile[rel >> 5] >> (uint32_t)(rel & 31)) & 15u;
}
__device__ inline uint32_t sign_tile_nibble(const uint32_t* __restrict__ tile, int rel) {
  return (tile[rel >> 5] >> (uint32_t)(rel & 31)) & 15u;
}
__device__ inline float sign_tile_r(const uint32_t* __restrict__ tile, int64_t rel) {
  return ((tile[rel >> 5] >> (uint32_t)(rel & 31)) & 1u) ? 0.5f : -0.5f;
}
__device__ inline void nibble_to_r4(uint32_t nib, float (&r)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = ((nib >> k) & 1u) ? 0.5f : -0.5f;
}
// The same signs already multiplied by 3^-1/2: rc = r * 3^-1/2 = +-(3^-1/2 / 2), so that the scale-gradient noise term
// (3^-1/2 * gq) * r  (gdnsq.py:55) is ONE product gq * rc -- the SAME bits (halving and sign commute with the rounding of
// the product; only a subnormal product could differ) at 2 + 1 instructions per element instead of 3 + 2.
#define MHAQ_HALF_INV_SQRT3_BITS 0x3E93CD3Au      // bits of 0.57735026918962584f * 0.5f
__device__ inline void nibble_to_rc4(uint32_t nib, float (&rc)[4]) {
  const uint32_t inv = ~nib;                     // stream bit set = +, i.e. sign bit clear
#pragma unroll
  for (int k = 0; k < 4; ++k) rc[k] = __uint_as_float(((inv << (31 - k)) & 0x80000000u) | MHAQ_HALF_INV_SQRT3_BITS);
}
__device__ inline float sign_to_rc(float r_half) { return r_half * MHAQ_INV_SQRT3; }      // exact: +-0.5 * c

// W consecutive floats as one access (W = 4: a 16-byte load/store, global or LDS; W = 1: a dword)
typedef float vf4 __attribute__((ext_vector_type(4)));
template <int W>
__device__ __forceinline__ void ldv(const float* p, float (&v)[W]) {
  if constexpr (W == 4) {
    const vf4 t = *reinterpret_cast<const vf4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = *p;
  }
}
template <int W>
__device__ __forceinline__ void stv(float* p, const float (&v)[W]) {
  if constexpr (W == 4) {
    vf4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<vf4*>(p) = t;
  } else {
    *p = v[0];
  }
}
// The same accesses on pointers the HOST guarantees to be device global memory.  To the compiler only kernel arguments
// are known global: a pointer read out of a descriptor table in memory (the multi-tensor launches) is FLAT, its loads
// become flat_load -- which count on vmcnt AND lgkmcnt, so every LDS read (sign tile, staged row, reduction scratch) waits
// for all of the row's loads in flight -- and a uniform one cannot be a scalar load.  Through address space 1 they are
// global_load / s_load whatever the pointer's origin.
#define MHAQ_GLOBAL_AS __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ T MHAQ_GLOBAL_AS* gptr(T* p) { return (T MHAQ_GLOBAL_AS*)p; }
__device__ __forceinline__ float ldg(const float* p) { return *gptr(p); }
__device__ __forceinline__ void stg(float* p, float v) { *gptr(p) = v; }
template <int W>
__device__ __forceinline__ void ldvg(const float* p, float (&v)[W]) {
  if constexpr (W == 4) {
    const vf4 t = *gptr(reinterpret_cast<const vf4*>(p));
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = *gptr(p);
  }
}
template <int W>
__device__ __forceinline__ void stvg(float* p, const float (&v)[W]) {
  if constexpr (W == 4) {
    vf4 t = {v[0], v[1], v[2], v[3]};
    *gptr(reinterpret_cast<vf4*>(p)) = t;
  } else {
    *gptr(p) = v[0];
  }
}

// ---------------------------------------------------------------- reductions
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ inline float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_down(v, o, 64));
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
  return v;
}

// fp32 wave64 sum on the DPP cross-lane network (no LDS traffic): 6 v_add_f32 with DPP
// operands per value; the total lands in lane 63 and is returned wave-uniform.
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_src(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true));
}
__device__ inline float wave_sum_dpp(float v) {
  v += dpp_src<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v += dpp_src<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v += dpp_src<0x141, 0xF>(v);   // row_half_mirror
  v += dpp_src<0x140, 0xF>(v);   // row_mirror        -> every lane of a 16-lane row holds the row sum
  v += dpp_src<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
  v += dpp_src<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Block sum of K per-thread fp32 partials: DPP within the wave, fp64 across the waves.
// Result valid in thread 0.  `sm` holds K*(blockDim/64) floats.
template <int K>
__device__ inline void block_sum_f32(const float (&v)[K], double (&out)[K], float* sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  float w[K];
#pragma unroll
  for (int k = 0; k < K; ++k) w[k] = wave_sum_dpp(v[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) sm[wave * K + k] = w[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = 0.0;
    for (int i = 0; i < nw; ++i) {
#pragma unroll
      for (int k = 0; k < K; ++k) out[k] += (double)sm[i * K + k];
    }
  }
}

// Block-wide fp64 sum of K values; result valid in thread 0.  `sm` holds K*(blockDim/64) doubles.
template <int K>
__device__ inline void block_sum(double (&v)[K], double* sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
  if (nw == 1) return;
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) sm[wave * K + k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < nw; ++w) {
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] += sm[w * K + k];
    }
  }
}

// The same sum, valid in EVERY thread after ONE barrier: each wave leaves its partial in LDS and every thread adds
// the (at most 16) wave partials itself, in wave order -- the order block_sum's thread 0 uses, so the totals are
// the same bits.  Replaces a reduce + broadcast pair (4 barriers) per value.  `sm` holds K*(blockDim/64) doubles
// and must not still be read by an earlier reduction (give each call site its own buffer).
template <int K>
__device__ inline void block_sum_all(double (&v)[K], double* sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) sm[wave * K + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = sm[k];
  for (int w = 1; w < nw; ++w) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += sm[w * K + k];
  }
}

// int32 wave64 sum on the DPP network (see wave_sum_dpp): wave-uniform result.
__device__ inline int wave_sum_dpp_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);   // row_bcast:15
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);   // row_bcast:31
  return __builtin_amdgcn_readlane(v, 63);
}

// block_sum_all for K fp64 sums AND two integer tallies behind the SAME single barrier: the tallies (tie counts: small
// exact integers) ride six 1-instruction DPP adds each instead of six 64-bit shuffle + fp64-add steps.  `smd` holds
// K * (blockDim / 64) doubles, `smi` 2 * (blockDim / 64) ints.  The totals come back in d[] and (as doubles: the callers'
// arithmetic is unchanged) in t[].
template <int K>
__device__ inline void block_sum_all_tally(double (&d)[K], int c0, int c1, double (&t)[2], double* smd, int* smi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) d[k] = wave_sum(d[k]);
  c0 = wave_sum_dpp_i32(c0);
  c1 = wave_sum_dpp_i32(c1);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) smd[wave * K + k] = d[k];
    smi[2 * wave] = c0;
    smi[2 * wave + 1] = c1;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) d[k] = smd[k];
  int s0 = smi[0], s1 = smi[1];
  for (int w = 1; w < nw; ++w) {
#pragma unroll
    for (int k = 0; k < K; ++k) d[k] += smd[w * K + k];
    s0 += smi[2 * w];
    s1 += smi[2 * w + 1];
  }
  t[0] = (double)s0;
  t[1] = (double)s1;
}

// NaN-propagating block min and max (torch.amin / amax semantics) in every thread after one barrier.
// `sm` holds 2*(blockDim/64) floats.
__device__ inline void block_minmax_all(float& mn, float& mx, bool nan, float* sm) {
  if (nan) { mn = NAN; mx = NAN; }
  auto nmin = [](float a, float b) { return (a != a || b != b) ? NAN : fminf(a, b); };
  auto nmax = [](float a, float b) { return (a != a || b != b) ? NAN : fmaxf(a, b); };
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = nmin(mn, __shfl_down(mn, o, 64));
    mx = nmax(mx, __shfl_down(mx, o, 64));
  }
  const int nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) { sm[2 * (threadIdx.x >> 6)] = mn; sm[2 * (threadIdx.x >> 6) + 1] = mx; }
  __syncthreads();
  mn = sm[0]; mx = sm[1];
  for (int w = 1; w < nw; ++w) { mn = nmin(mn, sm[2 * w]); mx = nmax(mx, sm[2 * w + 1]); }
}

// ---------------------------------------------------------------- quantizer core
// One element of Quantizer.quantize (gdnsq.py:197-208).  NaN x propagates like torch.clamp.
struct QCore { float v0, v1, v, n, q; };
__device__ inline QCore quant_core(float x, float s, float zp, float lo, float hi) {
  QCore c;
  float t = fmaxf(x, lo);             // clamp = min(max(x, lo), hi); hi wins when lo > hi
  t = fminf(t, hi);
  c.v0 = (x != x) ? x : t;
  c.v1 = c.v0 - zp;
  c.v = c.v1 / s;                     // IEEE-correct division (never x * (1/s))
  c.n = rintf(c.v) - c.v;             // QNoise.forward: round-half-even noise
  c.q = c.v + c.n;                    // == rne(v) for finite v, NaN for +-inf like the reference
  return c;
}
__device__ inline float dequant(float q, float s, float zp) { return q * s + zp; }

// ---- backward-pass form of the core: divisions by a wave-uniform scale as exact FMA corrections
struct BwdCtx {
  float s, zp, lo, hi;
  float rs;        // RN(1/s)
  bool lo_lt_hi, hi_lt_lo;
  bool fast_div;   // s normal and its significand not all ones: Markstein correction is exact
};

__device__ inline BwdCtx make_bwd_ctx(float s, float zp, float lo, float hi) {
  BwdCtx k;
  k.s = s; k.zp = zp; k.lo = lo; k.hi = hi;
  k.rs = 1.0f / s;
  k.lo_lt_hi = lo < hi;
  k.hi_lt_lo = hi < lo;
  const uint32_t sb = __float_as_uint(s), rb = __float_as_uint(k.rs);
  const uint32_t se = (sb >> 23) & 0xff, re = (rb >> 23) & 0xff;
  k.fast_div = se != 0 && se != 255 && re != 0 && re != 255 && (sb & 0x7FFFFFu) != 0x7FFFFFu;
  return k;
}

// Quantizer core for the backward pass: as quant_core, with the division by the wave-uniform
// scale done as  q0 = v1*rs;  q1 = q0 + (v1 - s*q0)*rs;  v = q1 + (v1 - s*q1)*rs  (residuals
// exact by FMA).  q1 is within 1/2 ulp (+2^-24 ulp) of v1/s, so by Markstein's theorem the last
// step is the correctly rounded quotient: the same bits as the forward's IEEE division, at 5
// VALU instructions instead of 11.  Degenerate scales (fast_div == false) take the division.
__device__ inline QCore quant_core_bwd(float x, const BwdCtx& k) {
  if (!k.fast_div) return quant_core(x, k.s, k.zp, k.lo, k.hi);
  QCore c;
  float t = fmaxf(x, k.lo);
  t = fminf(t, k.hi);
  c.v0 = (x != x) ? x : t;
  c.v1 = c.v0 - k.zp;
  const float q0 = c.v1 * k.rs;
  const float q1 = __fmaf_rn(__fmaf_rn(-k.s, q0, c.v1), k.rs, q0);
  c.v = __fmaf_rn(__fmaf_rn(-k.s, q1, c.v1), k.rs, q1);
  c.n = rintf(c.v) - c.v;
  c.q = c.v + c.n;
  return c;
}

// The same core for a WEIGHT quantizer, whose clamp bounds are -inf / +inf (gdnsq_conv2d.py:76-77): clamp(x) == x for
// every x (NaN and +-inf included), so v0 = x and the four clamp / NaN-select instructions of quant_core_bwd drop out.
// Same bits as quant_core_bwd(x, make_bwd_ctx(s, zp, -inf, +inf)) for every finite x and NaN.  Range of the exact-
// quotient claim (also quant_core_bwd's): |v1 / s| inside the normal range and v1 finite -- a weight of +-inf turns
// into NaN one operation earlier than in the reference (whose q = inf + (round(inf) - inf) is NaN as well), and a
// quotient below 2^-126 may differ in its last denormal bit (q = 0 and gW are unaffected).
__device__ inline QCore quant_core_w(float x, const BwdCtx& k) {
  QCore c;
  c.v0 = x;
  c.v1 = x - k.zp;
  if (k.fast_div) {
    const float q0 = c.v1 * k.rs;
    const float q1 = __fmaf_rn(__fmaf_rn(-k.s, q0, c.v1), k.rs, q0);
    c.v = __fmaf_rn(__fmaf_rn(-k.s, q1, c.v1), k.rs, q1);
  } else {
    c.v = c.v1 / k.s;
  }
  c.n = rintf(c.v) - c.v;
  c.q = c.v + c.n;
  return c;
}

// x / s to within one ulp (faithfully rounded: one Newton step on x * RN(1/s), residual exact by FMA) -- for terms that
// only enter a REDUCED gradient (the AEWGS d/ds term gv * (v / s)), where a last-bit difference per term is far inside
// the 1e-6 * sum|terms| bar; 3 VALU instructions against the 8 of quot().
__device__ inline float quot_faithful(float x, const BwdCtx& k) {
  if (!k.fast_div) return x / k.s;
  const float q0 = x * k.rs;
  return __fmaf_rn(__fmaf_rn(-k.s, q0, x), k.rs, q0);
}

// (G*s)/s for the STE / LSQ estimators, where gv == g*s: g is a faithful estimate of that quotient, so one
// Markstein correction with rs = RN(1/s) gives the correctly rounded quotient -- the bits of the IEEE division.
__device__ inline float quot_of_product(float g, float gv, const BwdCtx& k) {
  return k.fast_div ? __fmaf_rn(__fmaf_rn(-k.s, g, gv), k.rs, g) : gv / k.s;
}

// x / s for ANY x with the wave-uniform scale of a BwdCtx (the EWGS / AEWGS estimators' gv / s and v / s): the two
// Markstein steps of quant_core_bwd, valid while the quotient stays well inside the normal range -- the residuals
// are then exact and the second step rounds correctly; zeros, infinities, NaNs, quotients near the ends of the
// exponent range and degenerate scales take the IEEE division (same bits either way; 5 VALU instructions instead
// of ~12 on the common path).
__device__ inline float quot(float x, const BwdCtx& k) {
  const float q0 = x * k.rs;
  const float a = fabsf(q0);
  if (!k.fast_div || !(a > 0x1p-100f && a < 0x1p100f)) return x / k.s;
  const float q1 = __fmaf_rn(__fmaf_rn(-k.s, q0, x), k.rs, q0);
  return __fmaf_rn(__fmaf_rn(-k.s, q1, x), k.rs, q1);
}

__device__ inline float sign_f(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

// Estimator-specific d(noise)/dv contribution (QN*.backward, grad_input):
//   STE/LSQ: gq*0 ; EWGS: -|gq|*e*0.01 ; AEWGS: -gq*min(delta*sign(gq)*e, 0.99)
template <int METHOD>
__device__ inline float noise_grad_v(float gq, float e, float delta) {
  if (METHOD == MHAQ_FQ_EWGS) return -fabsf(gq) * e * 0.01f;
  if (METHOD == MHAQ_FQ_AEWGS) {
    float num_full = sign_f(gq) * e;
    float gsc = fminf(1.0f * delta * num_full, 0.99f);
    // torch.clamp_max propagates NaN
    if (delta * num_full != delta * num_full) gsc = delta * num_full;
    return -gq * gsc;
  }
  return gq * 0.f;
}

// The AEWGS gradient scale of one weight in the per-channel kernels: gsc = clamp_max(delta * sign(gq) * e, 0.99),
// NaN-propagating like torch.clamp_max (gdnsq.py:135-139).  sign(gq) * e is e with gq's sign bit folded in (2 bit ops
// instead of two compares, two selects and a multiply); for gq == +-0 that yields +-e where the reference has +-0, which
// cannot be seen in gv = gq - gq * gsc (= +-0 - (+-0) * finite: the same zero either way).  The AEWGS *statistics* keep
// sign_f(): there a zero gradient must contribute exactly 0 to the mean.
__device__ inline float aewgs_gsc(float gq, float e, float delta) {
  const float nf = __uint_as_float(__float_as_uint(e) ^ (__float_as_uint(gq) & 0x80000000u));
  const float t = delta * nf;
  return (t > 0.99f) ? 0.99f : t;         // NaN compares false and is passed on
}

// delta = num / clamp_min(e2 - me^2, 1e-3)   (gdnsq.py:131-134).  torch.clamp_min passes a NaN on (fmaxf would return the
// bound): a compare + select, NaN compares false and stays -- reachable with externally supplied statistics whose e2 or
// me is not finite while num is.
__device__ inline float aewgs_delta(float num, float e2, float me) {
  const float d = e2 - me * me;
  const float den = (d < 1e-3f) ? 1e-3f : d;
  return num / den;
}

// Launch + status.  Every kernel of the library goes through hipLaunchKernel, whose RETURN VALUE is this launch's own
// status (configuration / argument errors): the thread's sticky "last error" is neither consulted nor cleared, so an error
// some earlier HIP call of the caller left behind is not reported as ours -- and not swallowed either (rounds 1-5 returned
// hipGetLastError() after a <<<>>> launch, which does both).  The status travels from the launch to the entry point's
// return statement in a thread-local word: per calling thread, overwritten by every launch, nothing a second thread can
// observe -- the library still keeps no state between calls.
template <class T> struct launch_arg { using type = T; };
inline int& launch_rc() { static thread_local int rc = 0; return rc; }
template <class... KArgs>
inline void launch_k(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st,
                     typename launch_arg<KArgs>::type... args) {
  void* ptrs[] = {(void*)&args...};
  launch_rc() = (int)hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, lds, st);
}
#define MHAQ_LAUNCH(kernel, grid, block, lds, st, ...) ::mhaq::launch_k(kernel, grid, block, lds, st, __VA_ARGS__)
inline int launch_status() { return launch_rc(); }

}  // namespace mhaq
