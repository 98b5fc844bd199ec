// Per-tensor fake-quant kernels for gfx950 (activation quantizer NoisyAct and the
// elementwise half of PER_TENSOR weight quantizers).  HBM-bound streaming kernels:
// block b owns the 256*U consecutive float4 starting at b*256*U and makes ONE pass over
// them (no grid-stride loop: measured on MI355X the single-pass, block-contiguous mapping
// with non-temporal 16 B/lane accesses reaches the float4-copy ceiling, 6.0-6.2 TB/s, where
// a persistent grid-stride form of the same math reached 5.1-5.3).  Scalars live in SGPRs;
// reductions go registers (fp32, <= 4*U + 1 terms) -> fp32 DPP wave reduction -> LDS -> fp64 over
// the 4 waves -> one fp32 partial row per block -> fixed-order fp64 finalize kernel (one workgroup
// per output; deterministic, no float atomics).
//
// Reference op chains replaced: gdnsq.py:189-229 (forward), the autograd graph of the
// same lines + QN*.backward gdnsq.py:35-147 (backward), gdnsq_act.py:51-54 (bw stats),
// gdnsq.py:211-217 (eval asserts).
#include "fq_common.hpp"

namespace mhaq {

// 16 B/lane streaming accesses.  Every tensor of this path is read once and written once per
// launch and is larger than the caches at BASELINE sizes, so loads and stores are
// non-temporal: measured on MI355X a float4 copy runs 6.0-6.1 TB/s with nt vs 5.5-5.7 TB/s
// without (tools/kbench.hip).
// The MHAQ_* knobs below exist for tools/variants.sh (A/B builds of the library); defaults are the
// measured optimum on MI355X.
#ifndef MHAQ_FWD_NT_LD
#define MHAQ_FWD_NT_LD 1
#endif
#ifndef MHAQ_FWD_NT_ST
#define MHAQ_FWD_NT_ST 1
#endif
#ifndef MHAQ_FWD_U
#define MHAQ_FWD_U 1   // float4 per lane (forward): block = 256*U float4 (U=1: 6.6 TB/s, U=4: 6.1)
#endif
#ifndef MHAQ_BWD_U
#define MHAQ_BWD_U 2   // float4 per lane per stream (x and g) in backward (measured: U=1 127 us, U=2 96, U=4 100-102 at 50 M)
#endif
#ifndef MHAQ_BWD_NT_LD
#define MHAQ_BWD_NT_LD 1
#endif
#ifndef MHAQ_BWD_NT_ST
#define MHAQ_BWD_NT_ST 1
#endif
template <bool NT>
__device__ inline vf4 ld4(const float* p, int64_t vidx) {
  const vf4* q = reinterpret_cast<const vf4*>(p) + vidx;
  return NT ? __builtin_nontemporal_load(q) : *q;
}
template <bool NT>
__device__ inline void st4(float* p, int64_t vidx, vf4 v) {
  vf4* q = reinterpret_cast<vf4*>(p) + vidx;
  if (NT) __builtin_nontemporal_store(v, q); else *q = v;
}

static inline int64_t blocks_for(int64_t n, int u) {
  const int64_t nvec = n >> 2;
  const int64_t per = (int64_t)kBlock * u;
  int64_t blocks = (nvec + per - 1) / per;
  return blocks < 1 ? 1 : blocks;
}

// =============================================================== forward
struct FwdStats { float qmin, qmax; int flags; bool bad; };

template <bool WRITE_Q, bool STATS>
__device__ inline float fwd_elem(float x, float s, float zp, float lo, float hi, float qlo, float qhi,
                                 float& q_out, FwdStats& st) {
  QCore c = quant_core(x, s, zp, lo, hi);
  if (WRITE_Q) q_out = c.q;
  if (STATS) {
    // gdnsq.py:211-217 per element costs five instructions here: the two range asserts are taken from the running
    // minimum / maximum after the loop (any(q < qlo) == min q < qlo; fminf / fmaxf skip NaNs exactly like the
    // reference's comparisons, which are false for them), and q is integral iff q == rne(q) (false for NaN and inf - inf,
    // like (q == floor(q)) | (q == ceil(q)))
    st.qmin = fminf(st.qmin, c.q);
    st.qmax = fmaxf(st.qmax, c.q);
#if MHAQ_FWD_STATS_RINT
    if (!(c.q == rintf(c.q))) st.flags |= MHAQ_FQ_FLAG_NOT_INTEGER;
#else
    // q = v + (rne(v) - v) IS rne(v) for every finite v (the subtraction is exact: |rne(v) - v| <= 1/2 and the two are
    // within a factor of two of each other, or rne(v) = 0; the sum is then the representable integer itself) and NaN for
    // v = NaN / +-inf (inf - inf).  So "q is not an integer" -- (q == floor(q)) | (q == ceil(q)) false, gdnsq.py:213-214 --
    // is exactly "q is NaN": one compare, kept as a lane mask that the wave ORs in a scalar register (round 6; rounds 2-5
    // spent rintf + compare + select + or per element on it; pinned by test_eval_flag_word_equals_the_three_reference_asserts
    // and the special-value tests)
    st.bad |= (c.q != c.q);
#endif
  }
  return dequant(c.q, s, zp);
}

// LOGP = NoisyAct.forward from its learnable parameters (gdnsq_act.py:42-48): ps/pzp/plo point at
// log_act_s / log_act_q / act_b; every wave derives s = exp2(log_s), qr = exp2(log_q), zp = lo = b,
// hi = (b + qr) - s in scalar registers (no separate exp2/add/sub launches) and block 0 publishes
// {s, zp, lo, hi, qr} in params_out for the backward and for side consumers of Quantizer.scale etc.
// NTLD: non-temporal loads of x.  Chosen at launch by size: tensors above kFwdPlainLoadElems stream through with
// the non-temporal policy (50.2 M elements: 63.3 us against 64.2 with default-policy loads, 25.1 M: 32.1 against 33.1),
// smaller ones load with the default policy (6.3 M: 8.7 us against 10.0, 12.5 M: 17.2 against 17.5; tools/ab_kernels.py
// over library variants, two rounds, gpurun_out/r02_ab1.txt).  Stores are non-temporal at every size.
constexpr int64_t kFwdPlainLoadElems = 16ll << 20;
// FU: float4 per lane.  1 for the training forward (measured optimum); the eval-mode variant (STATS) takes kFwdStatsU
// so that its per-block epilogue -- three wave reductions, a barrier, three partials -- is paid once per 4096 elements.
// Re-measured in round 6 on the round-6 library (profiles/r06_eval_fwd.txt, 50.2 M elements, training forward 63.2 us on
// that box): U = 4 70.3 us with the finalize, U = 2 71.8, U = 1 84.3 (49 000 blocks each ending in a barrier and a serial
// tail of one thread); with the barrier replaced by an LDS ticket (the last wave to arrive combines, the others retire)
// U = 1 / 2 / 4: 88.5 / 73.4 / 72.3.  4 stays.  The finalize launch is ~3.5 us of the 70.
#ifndef MHAQ_FWD_STATS_U
#define MHAQ_FWD_STATS_U 4
#endif
#ifndef MHAQ_FWD_STATS_RINT
#define MHAQ_FWD_STATS_RINT 0      // A/B knob: 1 = the rounds 2-5 integrality test (q == rne(q))
#endif
constexpr int kFwdStatsU = MHAQ_FWD_STATS_U;
// (-DMHAQ_FWD_MAXWAVES=n: an A/B knob for tools/variants.sh, an upper bound on the resident waves per SIMD)
#ifdef MHAQ_FWD_MAXWAVES
#define MHAQ_FWD_OCC __attribute__((amdgpu_waves_per_eu(1, MHAQ_FWD_MAXWAVES))) __launch_bounds__(kBlock)
#else
#define MHAQ_FWD_OCC __launch_bounds__(kBlock)
#endif
template <bool WRITE_Q, bool STATS, bool ALIGNED, bool LOGP, bool NTLD, int FU>
__global__ MHAQ_FWD_OCC void pt_fwd_kernel(
    const float* __restrict__ x, float* __restrict__ y, float* __restrict__ q_out, int64_t n,
    const float* __restrict__ ps, const float* __restrict__ pzp, const float* __restrict__ plo,
    const float* __restrict__ phi, float* __restrict__ partials /* [grid][3] */,
    float* __restrict__ params_out) {
  // The data loads go out FIRST: a wave lives for one float4 per lane, so every scalar-memory round trip in
  // front of its global load (kernel arguments -> parameter pointers -> parameters -> exp2) is time the wave
  // occupies a slot with nothing in flight.  The quantizer parameters are fetched and derived under the load.
  const int64_t nvec = n >> 2;
  const int64_t base = (int64_t)blockIdx.x * (kBlock * FU) + threadIdx.x;
  const bool full = ((int64_t)blockIdx.x + 1) * (kBlock * FU) <= nvec;
  vf4 a[FU];
  if (ALIGNED) {
#pragma unroll
    for (int u = 0; u < FU; ++u) {
      const int64_t idx = base + u * kBlock;
      if (full || idx < nvec) a[u] = ld4<NTLD>(x, idx);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s, zp, lo, hi;
  if (LOGP) {
    s = exp2f(*ps);
    const float qr = exp2f(*pzp);
    zp = lo = *plo;
    hi = (zp + qr) - s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      params_out[0] = s; params_out[1] = zp; params_out[2] = lo; params_out[3] = hi; params_out[4] = qr;
    }
  } else {
    s = *ps; zp = *pzp; lo = *plo; hi = *phi;
  }
  float qlo = 0.f, qhi = 0.f;
  if (STATS) {
    qlo = floorf((lo - zp) / s);
    qhi = ceilf((hi - zp) / s);
  }
  FwdStats st{INFINITY, -INFINITY, 0, false};
  const int lane = threadIdx.x & 63;

  if (ALIGNED) {
#pragma unroll
    for (int u = 0; u < FU; ++u) {
      const int64_t idx = base + u * kBlock;
      if (full || idx < nvec) {
        vf4 o;
        float q0, q1, q2, q3;
        o.x = fwd_elem<WRITE_Q, STATS>(a[u].x, s, zp, lo, hi, qlo, qhi, q0, st);
        o.y = fwd_elem<WRITE_Q, STATS>(a[u].y, s, zp, lo, hi, qlo, qhi, q1, st);
        o.z = fwd_elem<WRITE_Q, STATS>(a[u].z, s, zp, lo, hi, qlo, qhi, q2, st);
        o.w = fwd_elem<WRITE_Q, STATS>(a[u].w, s, zp, lo, hi, qlo, qhi, q3, st);
        st4<MHAQ_FWD_NT_ST>(y, idx, o);
        if (WRITE_Q) { vf4 qq = {q0, q1, q2, q3}; st4<MHAQ_FWD_NT_ST>(q_out, idx, qq); }
      }
    }
    // scalar tail (n % 4 elements)
    const int64_t t = (nvec << 2) + threadIdx.x;
    if (blockIdx.x == 0 && t < n) {
      float qq;
      y[t] = fwd_elem<WRITE_Q, STATS>(x[t], s, zp, lo, hi, qlo, qhi, qq, st);
      if (WRITE_Q) q_out[t] = qq;
    }
  } else {
    // unaligned pointers (tensor views): plain coalesced dword accesses
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
      float qq;
      y[i] = fwd_elem<WRITE_Q, STATS>(x[i], s, zp, lo, hi, qlo, qhi, qq, st);
      if (WRITE_Q) q_out[i] = qq;
    }
  }

  if (STATS) {
    __shared__ float smn[kBlock / 64], smx[kBlock / 64];
    __shared__ int sfl[kBlock / 64];
    float mn = wave_min(st.qmin), mx = wave_max(st.qmax);
    int fl = st.flags | (st.bad ? MHAQ_FQ_FLAG_NOT_INTEGER : 0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) fl |= __shfl_down(fl, o, 64);
    if (lane == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; sfl[threadIdx.x >> 6] = fl; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < kBlock / 64; ++w) { mn = fminf(mn, smn[w]); mx = fmaxf(mx, smx[w]); fl |= sfl[w]; }
      if (mn < qlo) fl |= MHAQ_FQ_FLAG_BELOW_MIN;
      if (mx > qhi) fl |= MHAQ_FQ_FLAG_ABOVE_MAX;
      // columns [3][grid]: the finalize reads each with its own workgroup, coalesced
      const int64_t nb = gridDim.x;
      partials[blockIdx.x] = mn;
      partials[nb + blockIdx.x] = mx;
      partials[2 * nb + blockIdx.x] = __int_as_float(fl);
    }
  }
}

// One workgroup per column of the [3][nparts] partials (min q, max q, flag words): a single workgroup pulls only
// ~35 GB/s, which made this launch 17 us behind a 50 M-element tensor's 49 000 partial rows.
constexpr int kFwdFinalThreads = 1024;
__global__ __launch_bounds__(kFwdFinalThreads) void pt_fwd_finalize_kernel(const float* __restrict__ partials,
                                                                            int nparts, float* __restrict__ qstats,
                                                                            int32_t* __restrict__ flags) {
  const int col = blockIdx.x;
  const float* p = partials + (int64_t)col * nparts;
  float v = (col == 0) ? INFINITY : -INFINITY;
  int fl = 0;
  // 8 independent loads in flight per thread and trip (one load per trip pulled ~35 GB/s per workgroup: a round trip per
  // 4 KB; 49 000 partial rows behind a 50 M-element tensor at U = 1, 2 M behind 2^31 elements)
  constexpr int kUnroll = 8;
  for (int i0 = threadIdx.x; i0 < nparts; i0 += kUnroll * kFwdFinalThreads) {
    float t[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      const int i = i0 + k * kFwdFinalThreads;
      t[k] = p[i < nparts ? i : i0];                  // (clamped: a duplicate changes no minimum, maximum or OR)
    }
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      if (col == 0) v = fminf(v, t[k]);
      else if (col == 1) v = fmaxf(v, t[k]);
      else fl |= __float_as_int(t[k]);
    }
  }
  __shared__ float sv[kFwdFinalThreads / 64];
  __shared__ int sf[kFwdFinalThreads / 64];
  v = (col == 0) ? wave_min(v) : wave_max(v);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) fl |= __shfl_down(fl, o, 64);
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = v; sf[threadIdx.x >> 6] = fl; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kFwdFinalThreads / 64; ++w) {
      v = (col == 0) ? fminf(v, sv[w]) : fmaxf(v, sv[w]);
      fl |= sf[w];
    }
    if (col == 0) { if (qstats) qstats[0] = v; }
    else if (col == 1) { if (qstats) qstats[1] = v; }
    else if (flags) *flags = fl;
  }
}

// =============================================================== backward
constexpr int kNAcc = 5;  // d/ds, d/dzp, d/dlo, d/dhi, count(x == zp)

// One element of the fused backward.  The quantizer core is recomputed so q and the rounding
// noise are bit-identical to the forward.  The other two divisions of the reference's graph are
// by the same wave-uniform scale:
//   * g1 = (g*s)/s : g is a faithful estimate of that quotient, and with rs = RN(1/s) one
//     Markstein correction  g + (gv - s*g)*rs  (two FMAs, residual exact) rounds to the
//     correctly rounded quotient -- the same bits as the IEEE division (STE/LSQ; the other
//     estimators and degenerate scales take the division);
//   * (v1/s)/s only feeds the reduced scale gradient; for STE/LSQ it cancels analytically
//     against g*q (see below), the other estimators take the division.
template <int METHOD, bool COUNT>
__device__ inline float bwd_elem(float x, float g, float r, float delta, const BwdCtx& k, float (&acc)[kNAcc]) {
  QCore c = quant_core_bwd(x, k);
  const float gq = g * k.s;                                   // dequantize: d(q*s)/dq
  const float gv = gq + noise_grad_v<METHOD>(gq, c.n, delta); // q = v + noise(v)
  float g1;                                                   // v = v1 / s
  if ((METHOD == MHAQ_FQ_STE || METHOD == MHAQ_FQ_LSQ) && k.fast_div)
    g1 = __fmaf_rn(__fmaf_rn(-k.s, g, gv), k.rs, g);
  else
    g1 = gv / k.s;
  const float noise_s = (METHOD == MHAQ_FQ_LSQ) ? gq * c.n : (MHAQ_INV_SQRT3 * gq) * r;
  // d/ds: mul-backward g*q, div-backward -gv*((v1/s)/s), noise estimator term.  For STE/LSQ
  // gv == g*s, so g*q - gv*(v/s) == g*(q - v) == g*noise exactly: one product instead of the
  // difference of two ~|q|-times larger ones (the reference sums those separately in fp32 and
  // loses ~1e-7*sum|g*q| to cancellation; measured on a ResNet-20 layer this form lands 10x closer
  // to the fp64 value than the eager chain does).  EWGS/AEWGS keep the general form.
  if (METHOD == MHAQ_FQ_STE || METHOD == MHAQ_FQ_LSQ)
    acc[0] += g * c.n + noise_s;
  else
    acc[0] += (g * c.q + (-gv) * (c.v / k.s)) + noise_s;
  acc[1] += g - g1;                                           // +zp in dequantize, -zp before the divide
  const bool lt = x < k.lo, gt = x > k.hi;
  acc[2] += (lt && k.lo_lt_hi) ? g1 : 0.f;                    // clamp_backward_min_max
  if (COUNT) {
    // weight quantizers: the bounds never clip (lo = -inf; hi = +inf, or the tensor's own maximum when the caller
    // wants the amax tie count of the regulariser input, mhaq_fq_wlayer_ptl_bwd), so dL/dhi is identically 0 and
    // its slot carries count(x == hi) instead
    acc[3] += (x == k.hi) ? 1.f : 0.f;
    acc[4] += (x == k.zp) ? 1.f : 0.f;                        // amin tie count
  } else {
    acc[3] += (gt || k.hi_lt_lo) ? g1 : 0.f;
  }
  return ((x >= k.lo) && (x <= k.hi)) ? g1 : 0.f;             // clamp_backward
}

// The same element for the estimators of the shipped configurations (STE, LSQ) on a well-formed quantizer -- `fast`:
// scale positive, normal and Markstein-exact (fast_div), lo < hi -- at ~30 VALU instructions instead of ~45 (the small
// activation tensors are latency- / VALU-limited: all of a 4 M-element tensor's workgroups are resident at once, and
// every instruction of a wave's life is on the launch's critical path; profiles/r04_size_ceilings.txt).  Same bits for
// gx (the elementwise output) as bwd_elem; the REDUCED sums are the same quantities through fewer roundings:
//   * clamp: v0 = x > hi ? hi : (x < lo ? lo : x) re-uses the two compares the clamp masks need anyway (== min(max(x, lo),
//     hi) for lo < hi; a NaN x compares false twice and stays NaN, as torch.clamp has it);
//   * gv = gq + gq*0 as ONE fma (the product is an exact +-0 or NaN: the same value as the two-step form);
//   * d/ds: STE  g*n + (3^-1/2 g s) r  as fma(g, n + rsc, .) with rsc = +-(3^-1/2 s / 2) = (3^-1/2 s) r built from the
//     sign bit by one shift and one bit-field insert;  LSQ  g*n + (g s) n  as fma(g + g s, n, .).
template <int METHOD, bool COUNT>
__device__ __forceinline__ float bwd_elem_fast(float x, float g, float rsc, const BwdCtx& k, float (&acc)[kNAcc]) {
  const bool lt = x < k.lo, gt = x > k.hi, ord = (x == x);
  const float v0 = gt ? k.hi : (lt ? k.lo : x);
  const float v1 = v0 - k.zp;
  const float q0 = v1 * k.rs;
  const float q1 = __fmaf_rn(__fmaf_rn(-k.s, q0, v1), k.rs, q0);
  const float v = __fmaf_rn(__fmaf_rn(-k.s, q1, v1), k.rs, q1);       // == v1 / s, correctly rounded (quant_core_bwd)
  const float n = rintf(v) - v;
  const float gq = g * k.s;
  const float gv = __fmaf_rn(gq, 0.f, gq);
  const float g1 = __fmaf_rn(__fmaf_rn(-k.s, g, gv), k.rs, g);        // == gv / s, correctly rounded
  if (METHOD == MHAQ_FQ_LSQ) acc[0] = __fmaf_rn(g + gq, n, acc[0]);
  else acc[0] = __fmaf_rn(g, n + rsc, acc[0]);
  acc[1] += g - g1;
  acc[2] += lt ? g1 : 0.f;
  if (COUNT) {
    acc[3] += (x == k.hi) ? 1.f : 0.f;
    acc[4] += (x == k.zp) ? 1.f : 0.f;
  } else {
    acc[3] += gt ? g1 : 0.f;
  }
  return (ord && !lt && !gt) ? g1 : 0.f;
}

// +-h with the sign of stream bit `bit` of `nb` INVERTED (nb = ~sign word): bit set in the stream = +h.  h >= 0.
__device__ __forceinline__ float signed_half_scale(uint32_t nb, int bit, float h) {
  const uint32_t sgn = nb << (31 - bit);
  return __uint_as_float((sgn & 0x80000000u) | (__float_as_uint(h) & 0x7fffffffu));     // one v_bfi_b32
}

template <int METHOD>
__device__ inline float col_delta_at(const float* __restrict__ cs, int64_t period, int64_t j) {   // j < period
  if (METHOD != MHAQ_FQ_AEWGS) return 0.f;
  return aewgs_delta(cs[j], cs[period + j], cs[2 * period + j]);
}
template <int METHOD>
__device__ inline float col_delta(const float* __restrict__ cs, int64_t period, int64_t i) {
  if (METHOD != MHAQ_FQ_AEWGS) return 0.f;
  return col_delta_at<METHOD>(cs, period, i % period);
}

// Per-block partials are stored as fp32 (a block sums 16*U*256 terms in fp32/fp64 first; rounding one
// partial to fp32 costs 6e-8 of ITS magnitude, ~1e-9 of sum|terms| after the fp64 final sum) so the
// latency-bound finalize reads half the bytes.  K = number of live accumulators (4 without the tie counters).
template <bool ACT, int K, class T>
__device__ inline void write_partials(float* __restrict__ partials, const T (&t)[K], int64_t nb, int64_t b) {
  if (ACT) {
    partials[0 * nb + b] = (float)(t[0] - t[3]);
    partials[1 * nb + b] = (float)t[3];
    partials[2 * nb + b] = (float)((t[1] + t[2]) + t[3]);
  } else {
#pragma unroll
    for (int q = 0; q < K; ++q) partials[(int64_t)q * nb + b] = (float)t[q];
    if (K < kNAcc) partials[(int64_t)(kNAcc - 1) * nb + b] = 0.f;      // the tie counter column of a launch without counters
  }
}

// ACT: block 0 leaves {s, qr} of the forward's parameter block behind the three partial columns (the workspace
// has room: it is sized for kNAcc columns), so a deferred multi-quantizer finalize needs nothing but the
// workspace pointer -- which a caller can keep stable across steps.
__device__ inline void publish_act_scales(float* __restrict__ partials, const float* __restrict__ params, int64_t nb) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    partials[3 * nb] = params[0];
    partials[3 * nb + 1] = params[4];
  }
}

// ACT = NoisyAct backward: the per-block partials are already combined into the three learnable
// parameters' columns {d/ds - d/dhi, d/dhi, d/dzp + d/dlo + d/dhi} (hi = b + qr - s, zp = lo = b), so the
// finalize emits d/dlog_act_s, d/dlog_act_q, d/dact_b directly (no scalar autograd launches).
// Random signs (sign stream v3, fq_common.hpp): a block's 1024*U elements are 8*U consecutive Philox calls; the
// first 8*U lanes of wave 0 run them while all four waves' loads are in flight, the 128 bits of each call go to an LDS
// tile behind ONE barrier and every lane shifts its nibbles out of it -- a quarter of the Philox work of one call per
// lane, and none of it in three of the four waves.
// Occupancy by tensor size (BIG, chosen at launch).  Left to itself the register allocator takes 70 VGPRs for the STE
// instantiation (7 waves per SIMD) where 48 do without a spill.  The small tensors are latency-limited and want every
// wave slot: bounded to 8 waves, 12.5 M elements run 26.3-26.5 us against 26.8-27.5, 4.1 M 10.7 against 11.2.  The large
// ones are bandwidth-limited and want FEWER concurrent streams (DRAM pages stay open longer): at 50.2 M, 8 waves 98.1-98.4
// us, the allocator's 7 94.3-96.3, a cap of 6 93.6-95.1, 5 93.6-94.7, 4 97.6-98.8; at 25.1 M a cap of 5-6 49.0 against
// 49.3-49.5 (tools/run_ab.sh over tools/variants.sh builds, two interleaved rounds each: gpurun_out/r04d_ab.txt,
// gpurun_out/r04i_ab.txt).  Hence: at most 6 waves per SIMD from 20 Mi elements up, at least 8 below.  No instantiation
// spills under either bound.
constexpr int64_t kBwdBigElems = 20ll << 20;
#ifndef MHAQ_BWD_MINWAVES
#define MHAQ_BWD_MINWAVES 8
#endif
#ifndef MHAQ_BWD_BIG_MAXWAVES
#define MHAQ_BWD_BIG_MAXWAVES 6
#endif
// (-DMHAQ_BWD_MAXWAVES=n: an A/B knob for tools/variants.sh, the same cap on every instantiation)
#ifdef MHAQ_BWD_MAXWAVES
#define MHAQ_BWD_OCC __attribute__((amdgpu_waves_per_eu(1, MHAQ_BWD_MAXWAVES))) __launch_bounds__(kBlock)
#else
#define MHAQ_BWD_OCC                                                                                              \
  __attribute__((amdgpu_waves_per_eu((BIG ? 1 : MHAQ_BWD_MINWAVES), (BIG ? MHAQ_BWD_BIG_MAXWAVES : MHAQ_BWD_MINWAVES)))) \
  __launch_bounds__(kBlock)
#endif
template <int METHOD, bool RSIGN, bool ALIGNED, bool COUNT, bool ACT, bool BIG>
__global__ MHAQ_BWD_OCC void pt_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ gx, int64_t n,
    const float* __restrict__ ps, const float* __restrict__ pzp, const float* __restrict__ plo,
    const float* __restrict__ phi, const float* __restrict__ col_stats, int64_t period,
    const int8_t* __restrict__ r_sign, uint64_t seed, uint64_t offset, const uint64_t* __restrict__ offset_dev,
    float* __restrict__ partials) {
  constexpr bool NEED_R = (METHOD != MHAQ_FQ_LSQ);
  constexpr bool FAST_METHOD = (METHOD == MHAQ_FQ_STE || METHOD == MHAQ_FQ_LSQ);
  constexpr int K = COUNT ? kNAcc : kNAcc - 1;     // live accumulators
  constexpr int kTileCalls = 8 * MHAQ_BWD_U;       // 1024 * U elements per block / 128 per call
  float acc[kNAcc] = {0.f, 0.f, 0.f, 0.f, 0.f};   // <= 4*U (+1) terms per thread in the aligned path
  // data loads first, parameters under them (see pt_fwd_kernel)
  const int64_t nvec = n >> 2;
  const int64_t base = (int64_t)blockIdx.x * (kBlock * MHAQ_BWD_U) + threadIdx.x;
  const bool full = ((int64_t)blockIdx.x + 1) * (kBlock * MHAQ_BWD_U) <= nvec;
  vf4 a[MHAQ_BWD_U], b[MHAQ_BWD_U];
  uint32_t rs[MHAQ_BWD_U];
  if (ALIGNED) {
    // unconditional loads (lanes past the end of a ragged last block re-read the last float4 and drop it): straight-line
    // code, all 2*U loads in flight before anything waits -- a load under `if (idx < nvec)` made the register allocator
    // wait for the first pair before issuing the second.  The launcher sends n < 4 to the dword kernel (nvec >= 1 here).
#pragma unroll
    for (int u = 0; u < MHAQ_BWD_U; ++u) {
      const int64_t idx = base + u * kBlock;
      const int64_t idc = (full || idx < nvec) ? idx : nvec - 1;
      a[u] = ld4<MHAQ_BWD_NT_LD>(x, idc);
      b[u] = ld4<MHAQ_BWD_NT_LD>(g, idc);
      if (NEED_R && RSIGN) rs[u] = reinterpret_cast<const uint32_t*>(r_sign)[idc];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  offset = stream_offset(offset, offset_dev);
  // the parameter loads go out before the sign tile: their round trip runs under wave 0's Philox rounds, and the one
  // wait in front of the barrier covers both
  const float p_s = *ps, p_zp = *pzp, p_lo = *plo, p_hi = *phi;
  __shared__ __align__(16) uint32_t stile[4 * kTileCalls];      // sign_tile_fill stores 16 bytes per call
  if (ALIGNED && NEED_R && !RSIGN) {
    sign_tile_fill(stile, (int64_t)blockIdx.x * kTileCalls, kTileCalls, seed, offset);
    __syncthreads();
  }
  const BwdCtx k = make_bwd_ctx(p_s, p_zp, p_lo, p_hi);

  if (ALIGNED) {
    // this lane's sign nibbles: float4 u*256 + t of the block = bits [4*(u*256 + t), +4) of the tile, inverted once
    uint32_t nb[MHAQ_BWD_U];
#pragma unroll
    for (int u = 0; u < MHAQ_BWD_U; ++u) {
      nb[u] = 0;
      if (NEED_R && !RSIGN) nb[u] = ~(stile[(u * kBlock + (int)threadIdx.x) >> 3] >> (((int)threadIdx.x & 7) * 4));
    }
    const bool fast = FAST_METHOD && k.fast_div && (k.lo < k.hi) && (k.s > 0.f);      // wave-uniform
    if (FAST_METHOD && fast) {
      const float hcs = (MHAQ_INV_SQRT3 * k.s) * 0.5f;
#pragma unroll
      for (int u = 0; u < MHAQ_BWD_U; ++u) {
        const int64_t idx = base + u * kBlock;
        if (full || idx < nvec) {
          float rc[4] = {0.f, 0.f, 0.f, 0.f};
          if (NEED_R) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              rc[q] = RSIGN ? (((int8_t)((rs[u] >> (8 * q)) & 0xff) > 0) ? hcs : -hcs) : signed_half_scale(nb[u], q, hcs);
          }
          vf4 o;
          o.x = bwd_elem_fast<METHOD, COUNT>(a[u].x, b[u].x, rc[0], k, acc);
          o.y = bwd_elem_fast<METHOD, COUNT>(a[u].y, b[u].y, rc[1], k, acc);
          o.z = bwd_elem_fast<METHOD, COUNT>(a[u].z, b[u].z, rc[2], k, acc);
          o.w = bwd_elem_fast<METHOD, COUNT>(a[u].w, b[u].w, rc[3], k, acc);
          st4<MHAQ_BWD_NT_ST>(gx, idx, o);
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < MHAQ_BWD_U; ++u) {
        const int64_t idx = base + u * kBlock;
        if (full || idx < nvec) {
          float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
          if (NEED_R) {
            if (RSIGN) {
              r0 = sign_half((int8_t)(rs[u] & 0xff));
              r1 = sign_half((int8_t)((rs[u] >> 8) & 0xff));
              r2 = sign_half((int8_t)((rs[u] >> 16) & 0xff));
              r3 = sign_half((int8_t)((rs[u] >> 24) & 0xff));
            } else {
              const uint32_t nib = ~nb[u];
              r0 = (nib & 1u) ? 0.5f : -0.5f; r1 = (nib & 2u) ? 0.5f : -0.5f;
              r2 = (nib & 4u) ? 0.5f : -0.5f; r3 = (nib & 8u) ? 0.5f : -0.5f;
            }
          }
          // AEWGS: column of the first element (one 64-bit modulo per float4), then step with wrap-around
          int64_t j0 = 0, j1 = 0, j2 = 0, j3 = 0;
          if (METHOD == MHAQ_FQ_AEWGS) {
            j0 = (idx << 2) % period;
            j1 = (j0 + 1 == period) ? 0 : j0 + 1;
            j2 = (j1 + 1 == period) ? 0 : j1 + 1;
            j3 = (j2 + 1 == period) ? 0 : j2 + 1;
          }
          vf4 o;
          o.x = bwd_elem<METHOD, COUNT>(a[u].x, b[u].x, r0, col_delta_at<METHOD>(col_stats, period, j0), k, acc);
          o.y = bwd_elem<METHOD, COUNT>(a[u].y, b[u].y, r1, col_delta_at<METHOD>(col_stats, period, j1), k, acc);
          o.z = bwd_elem<METHOD, COUNT>(a[u].z, b[u].z, r2, col_delta_at<METHOD>(col_stats, period, j2), k, acc);
          o.w = bwd_elem<METHOD, COUNT>(a[u].w, b[u].w, r3, col_delta_at<METHOD>(col_stats, period, j3), k, acc);
          st4<MHAQ_BWD_NT_ST>(gx, idx, o);
        }
      }
    }
    const int64_t t = (nvec << 2) + threadIdx.x;
    if (blockIdx.x == 0 && t < n) {   // n % 4 tail elements
      float r = 0.f;
      if (NEED_R) r = RSIGN ? sign_half(r_sign[t]) : philox_r(t, seed, offset);
      gx[t] = bwd_elem<METHOD, COUNT>(x[t], g[t], r, col_delta<METHOD>(col_stats, period, t), k, acc);
    }
    if (BIG) {
      // bandwidth-limited launches: one partial row per block (the block's epilogue hides behind the other blocks' streams,
      // and the finalize behind a 50 M-element tensor reads 24,500 rows per column instead of 98,000)
      __shared__ float smf[K * (kBlock / 64)];
      float live[K];
#pragma unroll
      for (int q = 0; q < K; ++q) live[q] = acc[q];
      double tot[K];
      block_sum_f32<K>(live, tot, smf);
      if (threadIdx.x == 0) write_partials<ACT, K>(partials, tot, (int64_t)gridDim.x, (int64_t)blockIdx.x);
      if (ACT) publish_act_scales(partials, ps, (int64_t)gridDim.x);
    } else {
      // latency-limited launches: one partial row per WAVE -- no LDS, no second barrier, no serial tail in thread 0 (the
      // block-level form cost 0.6-0.75 us of a 10-33 us launch: timing-only build without it, gpurun_out/r04j_noreduce.txt).
      // The finalize sums 4x the rows in fp64, in the same fixed order.
      float wsum[K];
#pragma unroll
      for (int q = 0; q < K; ++q) wsum[q] = wave_sum_dpp(acc[q]);
      const int64_t nrows = (int64_t)gridDim.x * (kBlock / 64);
      if ((threadIdx.x & 63) == 0)
        write_partials<ACT, K>(partials, wsum, nrows, (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
      if (ACT) publish_act_scales(partials, ps, nrows);
    }
  } else {
    // unaligned tensor views: dword accesses, grid-stride, fp64 per-thread accumulators
    double dacc[kNAcc] = {0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
      float r = 0.f;
      if (NEED_R) r = RSIGN ? sign_half(r_sign[i]) : philox_r(i, seed, offset);
      float a1[kNAcc] = {0.f, 0.f, 0.f, 0.f, 0.f};
      gx[i] = bwd_elem<METHOD, COUNT>(x[i], g[i], r, col_delta<METHOD>(col_stats, period, i), k, a1);
#pragma unroll
      for (int q = 0; q < kNAcc; ++q) dacc[q] += (double)a1[q];
    }
    __shared__ double sm[kNAcc * (kBlock / 64)];
    block_sum<kNAcc>(dacc, sm);
    if (threadIdx.x == 0) write_partials<ACT, kNAcc>(partials, dacc, (int64_t)gridDim.x, (int64_t)blockIdx.x);
    if (ACT) publish_act_scales(partials, ps, (int64_t)gridDim.x);
  }
}

// Fixed-order final sum of the per-block fp64 partials -> K fp32 outputs.  Partials are stored
// transposed, partials[q * nparts + block]; block q of this kernel sums column q.  One CU pulls
// only ~35 GB/s, so a single workgroup took 14 us for the 490 KB a 50 M-element tensor leaves;
// one workgroup per column reads 98 KB each, coalesced, 8 independent loads in flight per lane.
constexpr int kFinalThreads = 1024;
__global__ __launch_bounds__(kFinalThreads) void sum_finalize_kernel(const float* __restrict__ partials,
                                                                       int nparts, float* __restrict__ out) {
  const float* col = partials + (int64_t)blockIdx.x * nparts;
  double v[1] = {0.0};
  int i = threadIdx.x;
  for (; i + 7 * kFinalThreads < nparts; i += 8 * kFinalThreads) {
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = col[i + j * kFinalThreads];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[0] += (double)t[j];
  }
  for (; i < nparts; i += kFinalThreads) v[0] += (double)col[i];
  __shared__ double sm[kFinalThreads / 64];
  block_sum<1>(v, sm);
  if (threadIdx.x == 0) out[blockIdx.x] = (float)v[0];
}

// NoisyAct: column sums -> d/dlog_act_s = (sum * s) * ln2, d/dlog_act_q = (sum * qr) * ln2 (exp2 backward,
// gdnsq_act.py:42-43), d/dact_b = sum.  params = {s, zp, lo, hi, qr} from the forward.
__global__ __launch_bounds__(kFinalThreads) void act_finalize_kernel(const float* __restrict__ partials,
                                                                       int nparts,
                                                                       const float* __restrict__ params,
                                                                       float* __restrict__ out) {
  const float* col = partials + (int64_t)blockIdx.x * nparts;
  double v[1] = {0.0};
  int i = threadIdx.x;
  for (; i + 7 * kFinalThreads < nparts; i += 8 * kFinalThreads) {
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = col[i + j * kFinalThreads];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[0] += (double)t[j];
  }
  for (; i < nparts; i += kFinalThreads) v[0] += (double)col[i];
  __shared__ double sm[kFinalThreads / 64];
  block_sum<1>(v, sm);
  if (threadIdx.x == 0) {
    const float g = (float)v[0];
    if (blockIdx.x == 0) out[0] = (g * params[0]) * 0.69314718055994531f;
    else if (blockIdx.x == 1) out[1] = (g * params[4]) * 0.69314718055994531f;
    else out[2] = g;
  }
}

// The same finalize for EVERY activation quantizer of a backward pass in one launch: workgroup 3*i + c sums
// column c of quantizer i's partials (same partition and order as act_finalize_kernel: identical bits) and
// reads {s, qr} from behind the columns.  grads_out is [nquant][3].
struct ActFinalizeDesc { const float* partials; int64_t nparts; };
static_assert(sizeof(ActFinalizeDesc) == sizeof(mhaq_act_finalize_desc), "descriptor layout must match the C header");
__global__ __launch_bounds__(kFinalThreads) void act_finalize_multi_kernel(const ActFinalizeDesc* __restrict__ descs,
                                                                             float* __restrict__ grads_out) {
  const int qi = blockIdx.x / 3, c = blockIdx.x % 3;
  const ActFinalizeDesc d = descs[qi];
  const int nparts = (int)d.nparts;
  // (a pointer read out of the table is flat to the compiler: through address space 1 the loads are global_load -- gptr, fq_common.hpp)
  const float* col = d.partials + (int64_t)c * nparts;
  double v[1] = {0.0};
  int i = threadIdx.x;
  for (; i + 7 * kFinalThreads < nparts; i += 8 * kFinalThreads) {
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = ldg(col + i + j * kFinalThreads);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[0] += (double)t[j];
  }
  for (; i < nparts; i += kFinalThreads) v[0] += (double)ldg(col + i);
  __shared__ double sm[kFinalThreads / 64];
  block_sum<1>(v, sm);
  if (threadIdx.x == 0) {
    const float g = (float)v[0];
    float* out = grads_out + 3 * (int64_t)qi;
    if (c == 0) out[0] = (g * ldg(d.partials + 3 * (int64_t)nparts)) * 0.69314718055994531f;
    else if (c == 1) out[1] = (g * ldg(d.partials + 3 * (int64_t)nparts + 1)) * 0.69314718055994531f;
    else out[2] = g;
  }
}

// =============================================================== min / max
template <bool ALIGNED>
__global__ __launch_bounds__(kBlock) void minmax_kernel(const float* __restrict__ x, int64_t n,
                                                        float* __restrict__ partials) {
  float mn = INFINITY, mx = -INFINITY;
  bool nan = false;
  auto upd = [&](float v) { mn = fminf(mn, v); mx = fmaxf(mx, v); nan |= (v != v); };
  if (ALIGNED) {
    const int64_t nvec = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * kBlock) {
      vf4 a = ld4<true>(x, i);
      upd(a.x); upd(a.y); upd(a.z); upd(a.w);
    }
    const int64_t t = (nvec << 2) + threadIdx.x;
    if (blockIdx.x == 0 && t < n) upd(x[t]);
  } else {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) upd(x[i]);
  }
  if (nan) { mn = NAN; mx = NAN; }   // torch.amin/amax propagate NaN
  __shared__ float smn[kBlock / 64], smx[kBlock / 64];
  // NaN-propagating wave reduce
  auto nmin = [](float a, float b) { return (a != a || b != b) ? NAN : fminf(a, b); };
  auto nmax = [](float a, float b) { return (a != a || b != b) ? NAN : fmaxf(a, b); };
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = nmin(mn, __shfl_down(mn, o, 64));
    mx = nmax(mx, __shfl_down(mx, o, 64));
  }
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; ++w) { mn = nmin(mn, smn[w]); mx = nmax(mx, smx[w]); }
    partials[blockIdx.x * 2 + 0] = mn;
    partials[blockIdx.x * 2 + 1] = mx;
  }
}

__global__ __launch_bounds__(kBlock) void minmax_finalize_kernel(const float* __restrict__ partials, int nparts,
                                                                 float* __restrict__ out) {
  auto nmin = [](float a, float b) { return (a != a || b != b) ? NAN : fminf(a, b); };
  auto nmax = [](float a, float b) { return (a != a || b != b) ? NAN : fmaxf(a, b); };
  float mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < nparts; i += kBlock) {
    mn = nmin(mn, partials[i * 2]);
    mx = nmax(mx, partials[i * 2 + 1]);
  }
  __shared__ float smn[kBlock / 64], smx[kBlock / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = nmin(mn, __shfl_down(mn, o, 64));
    mx = nmax(mx, __shfl_down(mx, o, 64));
  }
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; ++w) { mn = nmin(mn, smn[w]); mx = nmax(mx, smx[w]); }
    out[0] = mn;
    out[1] = mx;
  }
}

// =============================================================== amin tie scatter
__global__ __launch_bounds__(kBlock) void tie_scatter_kernel(const float* __restrict__ w, float* __restrict__ gw,
                                                             int64_t n, const float* __restrict__ pzp,
                                                             const float* __restrict__ grads) {
  const float zp = *pzp;
  const float gzp = grads[1], cnt = grads[4];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    if (w[i] == zp) gw[i] = gw[i] + (gzp * 1.0f) / cnt;  // amin backward: (grad * mask) / count
  }
}

// =============================================================== PER_TENSOR weight layer of any size
// aux[7] = {s = 2^log_s, zp = min w, max w, lwq = log2((max - min) + s), -inf, +inf, hi of the backward}: the scalar chain of
// gdnsq_conv2d.py:72,82-83 and model_helper.py:36-37,44 in the finalize of the min / max sweep.  aux + 4 / aux + 5 serve
// as the (never clipping) clamp bounds of the forward launch.
__global__ __launch_bounds__(kBlock) void ptl_aux_kernel(const float* __restrict__ partials, int nparts,
                                                         const float* __restrict__ log_s, float* __restrict__ aux) {
  auto nmin = [](float a, float b) { return (a != a || b != b) ? NAN : fminf(a, b); };
  auto nmax = [](float a, float b) { return (a != a || b != b) ? NAN : fmaxf(a, b); };
  float mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < nparts; i += kBlock) {
    mn = nmin(mn, partials[i * 2]);
    mx = nmax(mx, partials[i * 2 + 1]);
  }
  __shared__ float smn[kBlock / 64], smx[kBlock / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = nmin(mn, __shfl_down(mn, o, 64));
    mx = nmax(mx, __shfl_down(mx, o, 64));
  }
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; ++w) { mn = nmin(mn, smn[w]); mx = nmax(mx, smx[w]); }
    const float sc = exp2f(*log_s);
    aux[0] = sc;
    aux[1] = mn;
    aux[2] = mx;
    aux[3] = log2f((mx - mn) + sc);
    aux[4] = -INFINITY;
    aux[5] = INFINITY;
    aux[6] = (mx != mx) ? INFINITY : mx;     // the backward's hi: the maximum (a NaN tensor keeps the unbounded path)
  }
}

// sums[5] = {dL/ds, dL/dzp, 0, count(w == max), count(w == min)} of the streaming backward (hi = max) ->
// g_log_s[1] and the two tie-split shares; the arithmetic (and its order) is pc_bwd_body's LAYER branch (fq_pc.hip).
#define MHAQ_PT_LN2F 0.69314718055994531f
__global__ void ptl_scalar_kernel(const float* __restrict__ sums, const float* __restrict__ aux,
                                  const float* __restrict__ g_lwq, float* __restrict__ g_log_s,
                                  float* __restrict__ ties) {
  const float sc = aux[0], z = aux[1], rmx = aux[2];
  float gzp_local = sums[1];
  float gs_local = sums[0];
  float t_local = 0.f;
  if (g_lwq) t_local = *g_lwq / (((rmx - z) + sc) * MHAQ_PT_LN2F);
  gzp_local = gzp_local - t_local;
  gs_local = gs_local + t_local;
  g_log_s[0] = (gs_local * sc) * MHAQ_PT_LN2F;               // exp2 backward
  ties[0] = (gzp_local * 1.0f) / sums[4];                    // amin backward: (grad * mask) / count
  ties[1] = (t_local * 1.0f) / sums[3];                      // amax backward
}

__global__ __launch_bounds__(kBlock) void tie2_scatter_kernel(const float* __restrict__ w, float* __restrict__ gw,
                                                              int64_t n, const float* __restrict__ aux,
                                                              const float* __restrict__ ties) {
  const float z = aux[1], rmx = aux[2];
  const float tie = ties[0], tie_max = ties[1];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const float x = w[i];
    if (x == z || x == rmx) {
      float o = gw[i];
      if (x == z) o = o + tie;
      if (x == rmx) o = o + tie_max;
      gw[i] = o;
    }
  }
}

// =============================================================== AEWGS column statistics
// stats[0][j] = mean_c sign(G*s)*e, stats[1][j] = mean_c e^2, stats[2][j] = mean_c e  (c over rows)
// Two phases so that tall tensors fill the chip: block (bx, by) sums rows [by*rpc, (by+1)*rpc) of columns
// [bx*256, (bx+1)*256) (thread = column: coalesced) into fp64 partials[by][3][row]; the finalize sums the
// chunks of a column in fixed order.  With one chunk the partial kernel writes the means itself.
template <bool DIRECT>
__global__ __launch_bounds__(kBlock) void pt_colstats_kernel(const float* __restrict__ w, const float* __restrict__ G,
                                                             int64_t co, int64_t row, int64_t rows_per_chunk,
                                                             const float* __restrict__ ps,
                                                             const float* __restrict__ pzp,
                                                             const float* __restrict__ plo,
                                                             const float* __restrict__ phi,
                                                             float* __restrict__ stats,
                                                             double* __restrict__ partials) {
  const float s = *ps, zp = *pzp;
  const float lo = plo ? *plo : -INFINITY, hi = phi ? *phi : INFINITY;
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= row) return;
  const int64_t c0 = (int64_t)blockIdx.y * rows_per_chunk;
  const int64_t c1 = (c0 + rows_per_chunk < co) ? c0 + rows_per_chunk : co;
  double num = 0, e2 = 0, me = 0;
#pragma unroll 4
  for (int64_t c = c0; c < c1; ++c) {
    const float x = w[c * row + j], g = G[c * row + j];
    QCore q = quant_core(x, s, zp, lo, hi);
    const float gq = g * s;
    num += (double)(sign_f(gq) * q.n);
    e2 += (double)(q.n * q.n);
    me += (double)q.n;
  }
  if (DIRECT) {
    const float inv = (float)co;
    stats[j] = (float)num / inv;
    stats[row + j] = (float)e2 / inv;
    stats[2 * row + j] = (float)me / inv;
  } else {
    double* p = partials + (int64_t)blockIdx.y * 3 * row;
    p[j] = num;
    p[row + j] = e2;
    p[2 * row + j] = me;
  }
}

__global__ __launch_bounds__(kBlock) void pt_colstats_finalize_kernel(const double* __restrict__ partials,
                                                                      int nchunks, int64_t co, int64_t row,
                                                                      float* __restrict__ stats) {
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;   // over 3*row outputs
  if (j >= 3 * row) return;
  double v = 0;
  for (int k = 0; k < nchunks; ++k) v += partials[(int64_t)k * 3 * row + j];
  stats[j] = (float)v / (float)co;
}

// chunking of the `co` rows: enough (column-block x chunk) workgroups to fill 256 CUs a few times over,
// at least 8 rows per chunk, at most 65535 chunks (grid.y)
static inline void colstats_plan(int64_t co, int64_t row, int64_t* rows_per_chunk, int64_t* nchunks) {
  const int64_t col_blocks = (row + kBlock - 1) / kBlock;
  int64_t want = (2048 + col_blocks - 1) / col_blocks;      // chunks wanted
  if (want < 1) want = 1;
  int64_t rpc = (co + want - 1) / want;
  if (rpc < 8) rpc = 8;
  int64_t nc = (co + rpc - 1) / rpc;
  if (nc > 65535) { rpc = (co + 65534) / 65535; nc = (co + rpc - 1) / rpc; }
  *rows_per_chunk = rpc;
  *nchunks = nc;
}

__global__ __launch_bounds__(kBlock) void fill_r_kernel(int8_t* __restrict__ r, int64_t n, uint64_t seed,
                                                        uint64_t offset) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    r[i] = philox_r(i, seed, offset) > 0.f ? 1 : -1;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }
static inline int simple_grid(int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}

}  // namespace mhaq

using namespace mhaq;

template <int METHOD>
static int launch_pt_bwd(const float* x, const float* g, float* gx, int64_t n, const float* s, const float* zp,
                         const float* lo, const float* hi, const float* col_stats, int64_t period,
                         const int8_t* r_sign, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                         float* parts, int grid, bool al, bool count_ties, hipStream_t st, bool act = false) {
  const bool big = al && n >= kBwdBigElems;      // the dword kernel of unaligned views has one form
#define MHAQ_LAUNCH_BWD_(RS, AL, CT, AC, BG)                                                                   \
  MHAQ_LAUNCH((pt_bwd_kernel<METHOD, RS, AL, CT, AC, BG>), dim3(grid), dim3(kBlock), 0, st, x, g, gx, n, \
                     s, zp, lo, hi, col_stats, period, r_sign, seed, offset, offset_dev, parts)
#define MHAQ_LAUNCH_BWD(RS, CT, AC)                                                                            \
  do {                                                                                                         \
    if (!al) MHAQ_LAUNCH_BWD_(RS, false, CT, AC, false);                                                       \
    else if (big) MHAQ_LAUNCH_BWD_(RS, true, CT, AC, true);                                                    \
    else MHAQ_LAUNCH_BWD_(RS, true, CT, AC, false);                                                            \
  } while (0)
  if (act) {
    if (r_sign) MHAQ_LAUNCH_BWD(true, false, true); else MHAQ_LAUNCH_BWD(false, false, true);
  } else if (count_ties) {
    if (r_sign) MHAQ_LAUNCH_BWD(true, true, false); else MHAQ_LAUNCH_BWD(false, true, false);
  } else {
    if (r_sign) MHAQ_LAUNCH_BWD(true, false, false); else MHAQ_LAUNCH_BWD(false, false, false);
  }
#undef MHAQ_LAUNCH_BWD
#undef MHAQ_LAUNCH_BWD_
  return launch_status();
}

extern "C" {

int mhaq_fq_abi_version(void) { return MHAQ_FQ_ABI_VERSION; }

const char* mhaq_fq_error_string(int code) {
  switch (code) {
    case 0: return "ok";
    case MHAQ_FQ_EINVAL: return "invalid argument";
    case MHAQ_FQ_EWORKSPACE: return "workspace too small";
    case MHAQ_FQ_EALIGN: return "pointer not 4-byte aligned";
    case MHAQ_FQ_EUNSUPPORTED: return "unsupported configuration";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

int mhaq_fq_fill_r(int8_t* r_sign, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
  if (n < 0 || (n > 0 && !r_sign)) return MHAQ_FQ_EINVAL;
  if (n == 0) return 0;
  MHAQ_LAUNCH(fill_r_kernel, dim3(simple_grid(n)), dim3(kBlock), 0, (hipStream_t)stream, r_sign, n, seed, offset);
  return launch_status();
}

size_t mhaq_fq_pt_fwd_workspace_bytes(int64_t n) {
  const int64_t a = blocks_for(n, kFwdStatsU), b = simple_grid(n);
  return (size_t)(a > b ? a : b) * 3 * sizeof(float);
}

static int pt_fwd_impl(const float* x, float* y, int64_t n, const float* s, const float* zp, const float* lo,
                       const float* hi, float* q_out, float* qstats, int32_t* flags, void* workspace,
                       size_t workspace_bytes, void* stream, bool logp, float* params_out) {
  if (n < 0 || !s || !zp || !lo || (!logp && !hi) || (logp && !params_out) || (n > 0 && (!x || !y)))
    return MHAQ_FQ_EINVAL;
  if (!aligned4(x) || !aligned4(y) || !aligned4(q_out)) return MHAQ_FQ_EALIGN;
  const bool stats = qstats || flags;
  if (stats && (!workspace || workspace_bytes < mhaq_fq_pt_fwd_workspace_bytes(n))) return MHAQ_FQ_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const bool al = aligned16(x) && aligned16(y) && (!q_out || aligned16(q_out));
  const int64_t grid64 = al ? blocks_for(n, stats ? kFwdStatsU : MHAQ_FWD_U) : simple_grid(n);
  if (grid64 > 0x7fffffff) return MHAQ_FQ_EUNSUPPORTED;
  const int grid = (int)grid64;
  float* parts = (float*)workspace;
  const bool ntld = MHAQ_FWD_NT_LD && n > kFwdPlainLoadElems;
#define MHAQ_LAUNCH_FWD_(WQ, ST, AL, LP, NL)                                                                 \
  MHAQ_LAUNCH((pt_fwd_kernel<WQ, ST, AL, LP, NL, (ST ? kFwdStatsU : MHAQ_FWD_U)>), dim3(grid), dim3(kBlock), 0, \
                     st, x, y, q_out, n, s, zp, lo, hi, parts, params_out)
#define MHAQ_LAUNCH_FWD(WQ, ST, AL)                                                                          \
  do {                                                                                                       \
    if (logp) { if (ntld) MHAQ_LAUNCH_FWD_(WQ, ST, AL, true, true); else MHAQ_LAUNCH_FWD_(WQ, ST, AL, true, false); }   \
    else      { if (ntld) MHAQ_LAUNCH_FWD_(WQ, ST, AL, false, true); else MHAQ_LAUNCH_FWD_(WQ, ST, AL, false, false); } \
  } while (0)
  if (n > 0 || stats || logp) {
    if (q_out) {
      if (stats) { if (al) MHAQ_LAUNCH_FWD(true, true, true); else MHAQ_LAUNCH_FWD(true, true, false); }
      else       { if (al) MHAQ_LAUNCH_FWD(true, false, true); else MHAQ_LAUNCH_FWD(true, false, false); }
    } else {
      if (stats) { if (al) MHAQ_LAUNCH_FWD(false, true, true); else MHAQ_LAUNCH_FWD(false, true, false); }
      else       { if (al) MHAQ_LAUNCH_FWD(false, false, true); else MHAQ_LAUNCH_FWD(false, false, false); }
    }
  }
#undef MHAQ_LAUNCH_FWD
#undef MHAQ_LAUNCH_FWD_
  int rc = launch_status();
  if (rc) return rc;
  if (stats) {
    MHAQ_LAUNCH(pt_fwd_finalize_kernel, dim3(3), dim3(kFwdFinalThreads), 0, st, parts, grid, qstats, flags);
    rc = launch_status();
  }
  return rc;
}

int mhaq_fq_pt_fwd(const float* x, float* y, int64_t n, const float* s, const float* zp, const float* lo,
                   const float* hi, float* q_out, float* qstats, int32_t* flags, void* workspace,
                   size_t workspace_bytes, void* stream) {
  return pt_fwd_impl(x, y, n, s, zp, lo, hi, q_out, qstats, flags, workspace, workspace_bytes, stream, false,
                     nullptr);
}

int mhaq_fq_act_fwd(const float* x, float* y, int64_t n, const float* log_s, const float* log_q, const float* b,
                    float* params_out, float* qstats, int32_t* flags, void* workspace, size_t workspace_bytes,
                    void* stream) {
  return pt_fwd_impl(x, y, n, log_s, log_q, b, nullptr, nullptr, qstats, flags, workspace, workspace_bytes,
                     stream, true, params_out);
}

// partial rows of a backward launch over n elements: one per WAVE of the aligned kernel below kBwdBigElems, one per block
// from there up and in the dword kernel of unaligned views
static inline int64_t bwd_partial_rows(int64_t n, bool aligned) {
  if (!aligned) return simple_grid(n);
  const int64_t blocks = blocks_for(n, MHAQ_BWD_U);
  return n >= kBwdBigElems ? blocks : blocks * (kBlock / 64);
}
size_t mhaq_fq_pt_bwd_workspace_bytes(int64_t n) {
  const int64_t a = bwd_partial_rows(n, true), b = simple_grid(n);
  return ((size_t)(a > b ? a : b) * kNAcc + 2) * sizeof(float);      // + {s, qr} behind the columns (publish_act_scales)
}


int mhaq_fq_pt_bwd_partials(const float* x, const float* g, float* gx, int64_t n, const float* s,
                            const float* zp, const float* lo, const float* hi, int method,
                            const float* col_stats, int64_t period, const int8_t* r_sign, uint64_t seed,
                            uint64_t offset, const uint64_t* offset_dev, int count_ties, void* workspace,
                            size_t workspace_bytes, int32_t* nparts_out, void* stream) {
  if (n < 0 || !s || !zp || !lo || !hi || (n > 0 && (!x || !g || !gx))) return MHAQ_FQ_EINVAL;
  if (method < 0 || method > 3) return MHAQ_FQ_EINVAL;
  if (method == MHAQ_FQ_AEWGS && (!col_stats || period <= 0)) return MHAQ_FQ_EINVAL;
  if (!aligned4(x) || !aligned4(g) || !aligned4(gx)) return MHAQ_FQ_EALIGN;
  if (!workspace || workspace_bytes < mhaq_fq_pt_bwd_workspace_bytes(n)) return MHAQ_FQ_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const bool al = n >= 4 && aligned16(x) && aligned16(g) && aligned16(gx) && (!r_sign || aligned4(r_sign));
  const int64_t grid64 = al ? blocks_for(n, MHAQ_BWD_U) : simple_grid(n);
  if (grid64 > 0x7fffffff) return MHAQ_FQ_EUNSUPPORTED;
  const int grid = (int)grid64;
  float* parts = (float*)workspace;
  const bool ct = count_ties != 0;
  const int64_t rows64 = bwd_partial_rows(n, al);
  if (rows64 > 0x7fffffff) return MHAQ_FQ_EUNSUPPORTED;
  if (nparts_out) *nparts_out = (int32_t)rows64;
  switch (method) {
    case MHAQ_FQ_STE: return launch_pt_bwd<MHAQ_FQ_STE>(x, g, gx, n, s, zp, lo, hi, col_stats, period, r_sign, seed, offset, offset_dev, parts, grid, al, ct, st);
    case MHAQ_FQ_EWGS: return launch_pt_bwd<MHAQ_FQ_EWGS>(x, g, gx, n, s, zp, lo, hi, col_stats, period, r_sign, seed, offset, offset_dev, parts, grid, al, ct, st);
    case MHAQ_FQ_AEWGS: return launch_pt_bwd<MHAQ_FQ_AEWGS>(x, g, gx, n, s, zp, lo, hi, col_stats, period, r_sign, seed, offset, offset_dev, parts, grid, al, ct, st);
    default: return launch_pt_bwd<MHAQ_FQ_LSQ>(x, g, gx, n, s, zp, lo, hi, col_stats, period, r_sign, seed, offset, offset_dev, parts, grid, al, ct, st);
  }
}

int mhaq_fq_pt_bwd_finalize(const void* workspace, int32_t nparts, float* grads, void* stream) {
  if (!workspace || !grads || nparts <= 0) return MHAQ_FQ_EINVAL;
  MHAQ_LAUNCH(sum_finalize_kernel, dim3(kNAcc), dim3(kFinalThreads), 0, (hipStream_t)stream,
                     (const float*)workspace, (int)nparts, grads);
  return launch_status();
}

int mhaq_fq_pt_bwd(const float* x, const float* g, float* gx, int64_t n, const float* s, const float* zp,
                   const float* lo, const float* hi, int method, const float* col_stats, int64_t period,
                   const int8_t* r_sign, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int count_ties,
                   float* grads, void* workspace, size_t workspace_bytes, void* stream) {
  if (!grads) return MHAQ_FQ_EINVAL;
  int32_t nparts = 0;
  int rc = mhaq_fq_pt_bwd_partials(x, g, gx, n, s, zp, lo, hi, method, col_stats, period, r_sign, seed, offset,
                                   offset_dev, count_ties, workspace, workspace_bytes, &nparts, stream);
  if (rc) return rc;
  return mhaq_fq_pt_bwd_finalize(workspace, nparts, grads, stream);
}

size_t mhaq_fq_act_bwd_workspace_bytes(int64_t n) { return mhaq_fq_pt_bwd_workspace_bytes(n); }

int mhaq_fq_act_bwd_partials(const float* x, const float* g, float* gx, int64_t n, const float* params, int method,
                             const int8_t* r_sign, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                             void* workspace, size_t workspace_bytes, int32_t* nparts_out, void* stream) {
  if (n < 0 || !params || (n > 0 && (!x || !g || !gx))) return MHAQ_FQ_EINVAL;
  if (method != MHAQ_FQ_STE && method != MHAQ_FQ_LSQ && method != MHAQ_FQ_EWGS)
    return (method == MHAQ_FQ_AEWGS) ? MHAQ_FQ_EUNSUPPORTED : MHAQ_FQ_EINVAL;
  if (!aligned4(x) || !aligned4(g) || !aligned4(gx)) return MHAQ_FQ_EALIGN;
  if (!workspace || workspace_bytes < mhaq_fq_pt_bwd_workspace_bytes(n)) return MHAQ_FQ_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const bool al = n >= 4 && aligned16(x) && aligned16(g) && aligned16(gx) && (!r_sign || aligned4(r_sign));
  const int64_t grid64 = al ? blocks_for(n, MHAQ_BWD_U) : simple_grid(n);
  if (grid64 > 0x7fffffff) return MHAQ_FQ_EUNSUPPORTED;
  const int grid = (int)grid64;
  float* parts = (float*)workspace;
  const float *s = params, *zp = params + 1, *lo = params + 2, *hi = params + 3;
  const int64_t rows64 = bwd_partial_rows(n, al);
  if (rows64 > 0x7fffffff) return MHAQ_FQ_EUNSUPPORTED;
  if (nparts_out) *nparts_out = (int32_t)rows64;
  switch (method) {
    case MHAQ_FQ_STE: return launch_pt_bwd<MHAQ_FQ_STE>(x, g, gx, n, s, zp, lo, hi, nullptr, 0, r_sign, seed, offset, offset_dev, parts, grid, al, false, st, true);
    case MHAQ_FQ_EWGS: return launch_pt_bwd<MHAQ_FQ_EWGS>(x, g, gx, n, s, zp, lo, hi, nullptr, 0, r_sign, seed, offset, offset_dev, parts, grid, al, false, st, true);
    default: return launch_pt_bwd<MHAQ_FQ_LSQ>(x, g, gx, n, s, zp, lo, hi, nullptr, 0, r_sign, seed, offset, offset_dev, parts, grid, al, false, st, true);
  }
}

int mhaq_fq_act_bwd(const float* x, const float* g, float* gx, int64_t n, const float* params, int method,
                    const int8_t* r_sign, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, float* grads,
                    void* workspace, size_t workspace_bytes, void* stream) {
  if (!grads) return MHAQ_FQ_EINVAL;
  int32_t nparts = 0;
  int rc = mhaq_fq_act_bwd_partials(x, g, gx, n, params, method, r_sign, seed, offset, offset_dev, workspace,
                                    workspace_bytes, &nparts, stream);
  if (rc) return rc;
  MHAQ_LAUNCH(act_finalize_kernel, dim3(3), dim3(kFinalThreads), 0, (hipStream_t)stream,
                     (const float*)workspace, (int)nparts, params, grads);
  return launch_status();
}

int mhaq_fq_act_bwd_finalize_multi(const mhaq_act_finalize_desc* descs_device, int nquant, float* grads_out,
                                   void* stream) {
  if (nquant < 0 || (nquant > 0 && (!descs_device || !grads_out))) return MHAQ_FQ_EINVAL;
  if (nquant == 0) return 0;
  MHAQ_LAUNCH(act_finalize_multi_kernel, dim3(3 * (unsigned)nquant), dim3(kFinalThreads), 0,
                     (hipStream_t)stream, (const ActFinalizeDesc*)descs_device, grads_out);
  return launch_status();
}

size_t mhaq_fq_minmax_workspace_bytes(int64_t) { return (size_t)kMaxBlocks * 2 * sizeof(float); }

int mhaq_fq_minmax(const float* x, int64_t n, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (n <= 0 || !x || !out) return MHAQ_FQ_EINVAL;
  if (!aligned4(x)) return MHAQ_FQ_EALIGN;
  if (!workspace || workspace_bytes < mhaq_fq_minmax_workspace_bytes(n)) return MHAQ_FQ_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const bool al = aligned16(x);
  int64_t work = al ? (n >> 2) : n;
  const int grid = simple_grid(work > 0 ? work : 1);
  float* parts = (float*)workspace;
  if (al) MHAQ_LAUNCH((minmax_kernel<true>), dim3(grid), dim3(kBlock), 0, st, x, n, parts);
  else MHAQ_LAUNCH((minmax_kernel<false>), dim3(grid), dim3(kBlock), 0, st, x, n, parts);
  int rc = launch_status();
  if (rc) return rc;
  MHAQ_LAUNCH(minmax_finalize_kernel, dim3(1), dim3(kBlock), 0, st, parts, grid, out);
  return launch_status();
}

int mhaq_fq_pt_tie_scatter(const float* w, float* gw, int64_t n, const float* zp, const float* grads,
                           void* stream) {
  if (n < 0 || !zp || !grads || (n > 0 && (!w || !gw))) return MHAQ_FQ_EINVAL;
  if (n == 0) return 0;
  MHAQ_LAUNCH(tie_scatter_kernel, dim3(simple_grid(n)), dim3(kBlock), 0, (hipStream_t)stream, w, gw, n, zp, grads);
  return launch_status();
}

size_t mhaq_fq_wlayer_ptl_workspace_bytes(int64_t n) {
  const size_t a = mhaq_fq_minmax_workspace_bytes(n), b = mhaq_fq_pt_bwd_workspace_bytes(n) + 8 * sizeof(float);
  return a > b ? a : b;
}

int mhaq_fq_wlayer_ptl_fwd(const float* w, float* wq, const float* log_s, int64_t n, float* aux, void* workspace,
                           size_t workspace_bytes, void* stream) {
  if (n <= 0 || !w || !wq || !log_s || !aux) return MHAQ_FQ_EINVAL;
  if (!aligned4(w) || !aligned4(wq)) return MHAQ_FQ_EALIGN;
  if (!workspace || workspace_bytes < mhaq_fq_wlayer_ptl_workspace_bytes(n)) return MHAQ_FQ_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const bool al = aligned16(w);
  const int64_t work = al ? (n >> 2) : n;
  const int grid = simple_grid(work > 0 ? work : 1);
  float* parts = (float*)workspace;
  if (al) MHAQ_LAUNCH((minmax_kernel<true>), dim3(grid), dim3(kBlock), 0, st, w, n, parts);
  else MHAQ_LAUNCH((minmax_kernel<false>), dim3(grid), dim3(kBlock), 0, st, w, n, parts);
  int rc = launch_status();
  if (rc) return rc;
  MHAQ_LAUNCH(ptl_aux_kernel, dim3(1), dim3(kBlock), 0, st, parts, grid, log_s, aux);
  rc = launch_status();
  if (rc) return rc;
  return pt_fwd_impl(w, wq, n, aux, aux + 1, aux + 4, aux + 5, nullptr, nullptr, nullptr, nullptr, 0, stream, false,
                     nullptr);
}

int mhaq_fq_wlayer_ptl_bwd(const float* w, const float* G, float* gw, float* g_log_s, const float* aux,
                           const float* g_lwq, int64_t n, int method, const float* col_stats, int64_t period,
                           const int8_t* r_sign, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                           void* workspace, size_t workspace_bytes, void* stream) {
  if (n <= 0 || !w || !G || !gw || !g_log_s || !aux) return MHAQ_FQ_EINVAL;
  if (!workspace || workspace_bytes < mhaq_fq_wlayer_ptl_workspace_bytes(n)) return MHAQ_FQ_EWORKSPACE;
  const size_t nb = mhaq_fq_pt_bwd_workspace_bytes(n);
  float* sums = (float*)((char*)workspace + nb);       // [5] sums + [2] tie shares behind the per-block partials
  float* ties = sums + 5;
  int32_t nparts = 0;
  // hi = the tensor's maximum: never clips, and makes the streaming kernel count the amax ties next to the amin ones
  int rc = mhaq_fq_pt_bwd_partials(w, G, gw, n, aux, aux + 1, aux + 4, aux + 6, method, col_stats, period, r_sign, seed,
                                   offset, offset_dev, 1, workspace, nb, &nparts, stream);
  if (rc) return rc;
  rc = mhaq_fq_pt_bwd_finalize(workspace, nparts, sums, stream);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  MHAQ_LAUNCH(ptl_scalar_kernel, dim3(1), dim3(1), 0, st, sums, aux, g_lwq, g_log_s, ties);
  rc = launch_status();
  if (rc) return rc;
  MHAQ_LAUNCH(tie2_scatter_kernel, dim3(simple_grid(n)), dim3(kBlock), 0, st, w, gw, n, aux, ties);
  return launch_status();
}

size_t mhaq_fq_pt_aewgs_colstats_workspace_bytes(int64_t co, int64_t row) {
  if (co <= 0 || row <= 0) return 0;
  int64_t rpc, nc;
  colstats_plan(co, row, &rpc, &nc);
  return nc > 1 ? (size_t)nc * 3 * (size_t)row * sizeof(double) : 0;
}

int mhaq_fq_pt_aewgs_colstats(const float* w, const float* G, int64_t co, int64_t row, const float* s,
                              const float* zp, const float* lo, const float* hi, float* stats, void* workspace,
                              size_t workspace_bytes, void* stream) {
  if (co <= 0 || row <= 0 || !w || !G || !s || !zp || !stats) return MHAQ_FQ_EINVAL;
  int64_t rpc, nc;
  colstats_plan(co, row, &rpc, &nc);
  const int64_t col_blocks = (row + kBlock - 1) / kBlock;
  if (col_blocks > 0x7fffffff) return MHAQ_FQ_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (nc == 1) {
    MHAQ_LAUNCH(pt_colstats_kernel<true>, dim3((unsigned)col_blocks), dim3(kBlock), 0, st, w, G, co, row, rpc,
                       s, zp, lo, hi, stats, nullptr);
    return launch_status();
  }
  if (!workspace || workspace_bytes < mhaq_fq_pt_aewgs_colstats_workspace_bytes(co, row)) return MHAQ_FQ_EWORKSPACE;
  if (reinterpret_cast<uintptr_t>(workspace) & 7u) return MHAQ_FQ_EALIGN;
  double* parts = (double*)workspace;
  MHAQ_LAUNCH(pt_colstats_kernel<false>, dim3((unsigned)col_blocks, (unsigned)nc), dim3(kBlock), 0, st, w, G,
                     co, row, rpc, s, zp, lo, hi, stats, parts);
  int rc = launch_status();
  if (rc) return rc;
  const int64_t fb = (3 * row + kBlock - 1) / kBlock;
  MHAQ_LAUNCH(pt_colstats_finalize_kernel, dim3((unsigned)fb), dim3(kBlock), 0, st, parts, (int)nc, co, row,
                     stats);
  return launch_status();
}

}  // extern "C"
