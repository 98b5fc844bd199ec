/*
 * mhaq_fq.h -- C ABI of the MI355X-native fake-quantization path for MHAQ.
 *
 * Drop-in boundary.  The reference has no native code:
 * its "operator seam" is the Python class `Quantizer` and the autograd
 * Functions `QN*` in
 *     /root/reference/src/quantization/gdnsq/gdnsq.py:11-241
 * driven by the layer wrappers
 *     /root/reference/src/quantization/gdnsq/layers/gdnsq_act.py:39-55
 *     /root/reference/src/quantization/gdnsq/layers/gdnsq_conv2d.py:71-100
 *     /root/reference/src/quantization/gdnsq/layers/gdnsq_linear.py:61-78
 * Each entry point below names the reference lines whose eager op chain it
 * replaces.  INTEGRATION.md shows the ctypes / autograd.Function binding a
 * maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - plain C, no torch / C++ types; every pointer is a DEVICE pointer to
 *     contiguous fp32 unless stated otherwise; sizes are element counts.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - Devices: a call launches on `stream` and holds no device state of its own.  As with any HIP library that takes a
 *     stream, the calling thread's CURRENT device must be the device `stream` and every pointer belong to (with the NULL
 *     stream: the current device's default stream).  The host side above this ABI provides that for tensors on any device:
 *     the compiled autograd nodes and the ctypes ops switch to their input's device for the duration of a call
 *     (mhaq_amd/csrc/torch_binding.cpp MHAQ_ON_DEVICE_OF, mhaq_amd/ops.py _on_device), as torch's own ops do.
 *     (Those host-side switches have run with ONE visible device only -- the training boxes are one process per GPU --;
 *     tests/test_gpu_two_devices.py exercises them and is skipped until a process sees two devices: a model on another
 *     device than the current one is untested on hardware.)
 *   - stream-ordered and asynchronous: no host synchronisation, no allocation,
 *     no state kept between calls (a launch's status travels to the return statement in a thread-local word; one
 *     device attribute is memoised per device) -> re-entrant, thread-safe, hipGraph-capture-safe.  `seed` / `offset` of the random
 *     sign stream are host arguments, which a captured launch freezes; every backward entry point therefore
 *     also takes `offset_dev`, a nullable DEVICE pointer to one uint64 that the kernel adds to `offset`
 *     (effective offset = offset + *offset_dev, mod 2^64).  A captured training step keeps that word in
 *     memory and advances it once per replay (one 8-byte add), so replay k of a launch captured with host
 *     offset c draws the stream (seed, c + k * stride): fresh signs per step at 0 extra bytes per element.
 *     NULL = host offset only (eager callers).
 *   - the caller owns every buffer, including `workspace` (query the size
 *     with the matching *_workspace_bytes(); contents need no initialisation).
 *   - scalar quantizer parameters (scale, zero point, clamp bounds) are
 *     DEVICE pointers: they are live autograd tensors in the caller and must
 *     not be read back to the host.
 *   - return value: 0 = ok; >0 = hipError_t of the failed launch;
 *     <0 = argument error (MHAQ_FQ_E*).  Nothing throws.
 *
 * Map to the export list sketched in SURVEY.md section 8b:
 *   mhaq_fq_act_fwd / mhaq_fq_act_bwd        -> same names (NoisyAct from its learnable parameters);
 *                                               mhaq_fq_pt_fwd / _bwd take (s, zp, lo, hi) tensors instead
 *   mhaq_fq_w_fwd / mhaq_fq_w_bwd            -> mhaq_fq_wlayer_fwd / _bwd (per-channel, from log_wght_s),
 *                                               mhaq_fq_pc_fwd / _bwd (given scales), mhaq_fq_wlayer_pt_*
 *                                               and mhaq_fq_minmax + mhaq_fq_pt_* + tie_scatter (per-tensor)
 *   mhaq_fq_w_aewgs_stats -> [3,Co], _apply  -> mhaq_fq_pc_aewgs_stats, then *_bwd with `stats` != NULL
 *   multi-tensor variants (pointer table)    -> mhaq_fq_wlayer_fwd_multi / _bwd_multi
 *
 * Arithmetic contract: fp32 throughout, IEEE-correct division, round half to
 * even, no FMA contraction -- every elementwise output (y, q, gx, wq) is
 * bit-identical to the reference's eager chain on the same inputs.  Reduced
 * gradients are accumulated in fp64 and rounded once (deterministic: fixed
 * partition + fixed-order final sum, no float atomics).
 */
#ifndef MHAQ_FQ_H
#define MHAQ_FQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MHAQ_FQ_ABI_VERSION 4   /* v2: every backward entry point takes `offset_dev`; v3: sign stream layout (128 elements per Philox call);
                                  v4 (additive): mhaq_fq_pc_quantize */

/* Estimator selector == QNMethod value (gdnsq_utils.py:9-13). */
enum { MHAQ_FQ_STE = 0, MHAQ_FQ_EWGS = 1, MHAQ_FQ_AEWGS = 2, MHAQ_FQ_LSQ = 3 };

/* Argument errors. */
enum {
  MHAQ_FQ_EINVAL = -1,     /* null pointer / negative size / unknown method */
  MHAQ_FQ_EWORKSPACE = -2, /* workspace smaller than *_workspace_bytes()     */
  MHAQ_FQ_EALIGN = -3,     /* pointer not 4-byte aligned                     */
  MHAQ_FQ_EUNSUPPORTED = -4
};

/* Flag bits written by the eval-mode integrity check (gdnsq.py:211-217). */
enum {
  MHAQ_FQ_FLAG_BELOW_MIN = 1, /* some q < floor((min_val-zp)/s) */
  MHAQ_FQ_FLAG_ABOVE_MAX = 2, /* some q > ceil((max_val-zp)/s)  */
  MHAQ_FQ_FLAG_NOT_INTEGER = 4
};

int mhaq_fq_abi_version(void);
const char* mhaq_fq_error_string(int code);

/* ------------------------------------------------------------------------
 * Random sign stream of the stochastic scale gradient (gdnsq.py:54,104,144:
 * r = randint_like(v, 2) - 0.5).  In-kernel Philox4x32-10, layout v3 (ABI v3): ONE Philox call
 * yields the signs of 128 CONSECUTIVE elements -- all 128 output bits are spent.  Element i of a
 * call with (seed, offset):
 *     c = i >> 7                                      (the Philox call)
 *     out[0..3] = Philox4x32-10(counter = {lo(c), hi(c), lo(offset), hi(offset)},
 *                               key     = {lo(seed), hi(seed)})
 *     j = i & 127;   r = ((out[j >> 5] >> (j & 31)) & 1) ? +0.5 : -0.5
 * A pure function of (seed, offset, i): independent of the launch geometry.  (Layouts v1 / v2 drew
 * one call per lane and used 8 of its 128 bits; a workgroup now computes the calls its elements need
 * once, into LDS, and every lane shifts its bits out of them.)  mhaq_fq_fill_r materialises the
 * stream as int8 signs (+1/-1) so a checker can replay a backward with an explicit `r`;
 * tests/philox_ref.py restates the layout in numpy and holds mhaq_fq_fill_r to it.
 * Every backward entry point takes `r_sign`: non-NULL = read signs from
 * memory (int8, 1 B/elem extra: a positive value is +0.5, zero or negative is -0.5, so both the
 * +-1 coding of mhaq_fq_fill_r and a 0/1 coding such as torch.randint(0, 2) work), NULL = generate in-kernel.
 * ---------------------------------------------------------------------- */
int mhaq_fq_fill_r(int8_t* r_sign, int64_t n, uint64_t seed, uint64_t offset, void* stream);

/* ------------------------------------------------------------------------
 * Per-tensor fake-quant (one scale / zero point / clamp range for the whole
 * tensor): the activation quantizer NoisyAct (gdnsq_act.py:39-55) and the
 * elementwise half of a PER_TENSOR weight quantizer.
 *
 * Forward: replaces Quantizer.quantize + dequantize (gdnsq.py:189-229):
 *   v0 = clamp(x, *lo, *hi); v = (v0 - *zp) / *s; q = v + (rne(v) - v);
 *   y = q * *s + *zp
 * y may alias x.  Optional outputs (NULL to skip):
 *   q_out     [n]  the rounding indices (integer-valued fp32)
 *   qstats    [2]  {min q, max q} for NoisyAct.bw = log2(max-min+1)
 *                  (gdnsq_act.py:51-54); needs workspace
 *   flags     [1]  int32 OR of MHAQ_FQ_FLAG_* (eval asserts, no host sync)
 * ---------------------------------------------------------------------- */
size_t mhaq_fq_pt_fwd_workspace_bytes(int64_t n);
int mhaq_fq_pt_fwd(const float* x, float* y, int64_t n,
                   const float* s, const float* zp, const float* lo, const float* hi,
                   float* q_out, float* qstats, int32_t* flags,
                   void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the same chain given g = dL/dy; replaces the autograd graph of
 * gdnsq.py:197-208,229 + QN*.backward (gdnsq.py:35-57,63-84,90-107):
 *   gq = g * s; gv = gq + estimator(gq, e); g1 = gv / s;
 *   gx = g1 * [lo <= x <= hi]                                (gx may alias g)
 *   grads[0] = dL/ds  = sum g*q - sum gv*(v/s) + noise_term
 *   grads[1] = dL/dzp = sum g - sum g1
 *   grads[2] = dL/dlo = sum g1*[x < lo]     (0 if lo > hi)
 *   grads[3] = dL/dhi = sum g1*[x > hi]     (all of sum g1 if lo > hi)
 *   grads[4] = number of elements with x == zp (tie count for amin backward;
 *              only when count_ties != 0)
 * count_ties != 0 is the weight-quantizer mode, whose bounds never clip (lo = -inf; hi = +inf or the
 * tensor's own maximum): dL/dhi is identically 0 there and grads[3] carries the number of elements with
 * x == hi instead (the amax tie count mhaq_fq_wlayer_ptl_bwd needs; 0 for hi = +inf).
 * noise_term = 3^-1/2 sum gq*r (STE, EWGS, AEWGS) or sum gq*(q-v) (LSQ).
 * method: MHAQ_FQ_STE, MHAQ_FQ_LSQ, MHAQ_FQ_EWGS (as intended; the reference
 * raises at gdnsq.py:102), or MHAQ_FQ_AEWGS with `col_stats` [3][period]
 * (mean sign(gq)*e, mean e^2, mean e per position i % period: the reference's
 * reduce_to_shape quirk for a [1]-shaped scale, gdnsq.py:150-152).
 * ---------------------------------------------------------------------- */
size_t mhaq_fq_pt_bwd_workspace_bytes(int64_t n);
int mhaq_fq_pt_bwd(const float* x, const float* g, float* gx, int64_t n,
                   const float* s, const float* zp, const float* lo, const float* hi,
                   int method, const float* col_stats, int64_t period,
                   const int8_t* r_sign, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                   int count_ties /* 0: grads[4] is left 0 (activations) */,
                   float* grads /* [5] */,
                   void* workspace, size_t workspace_bytes, void* stream);

/* The two launches of mhaq_fq_pt_bwd, separately (bench.py times the streaming kernel on its
 * own; a multi-tensor caller can batch the finalizes): *_partials runs the streaming kernel and
 * leaves `*nparts_out` partial rows in `workspace`; *_finalize reduces them to grads[5]. */
int mhaq_fq_pt_bwd_partials(const float* x, const float* g, float* gx, int64_t n,
                            const float* s, const float* zp, const float* lo, const float* hi,
                            int method, const float* col_stats, int64_t period,
                            const int8_t* r_sign, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int count_ties,
                            void* workspace, size_t workspace_bytes, int32_t* nparts_out, void* stream);
int mhaq_fq_pt_bwd_finalize(const void* workspace, int32_t nparts, float* grads /* [5] */, void* stream);

/* ------------------------------------------------------------------------
 * NoisyAct from its LEARNABLE parameters (gdnsq_act.py:39-55): the scalar chain
 * s = exp2(log_act_s), qr = exp2(log_act_q), zp = min_val = act_b, max_val = act_b + qr - s and
 * its backward are folded into the same two launches (the eager reference spends ~12 extra
 * launches per quantizer per step on them).
 *   act_fwd: as mhaq_fq_pt_fwd; writes params_out[5] = {s, zp, lo, hi, qr} for the backward and
 *            for side consumers of Quantizer.scale / zero_point / min_val / max_val.
 *   act_bwd: as mhaq_fq_pt_bwd with `params` from the forward; grads[3] =
 *            {dL/dlog_act_s, dL/dlog_act_q, dL/dact_b}.  method: STE, LSQ or EWGS.
 * ---------------------------------------------------------------------- */
int mhaq_fq_act_fwd(const float* x, float* y, int64_t n,
                    const float* log_act_s, const float* log_act_q, const float* act_b,
                    float* params_out /* [5] */, float* qstats, int32_t* flags,
                    void* workspace, size_t workspace_bytes, void* stream);
size_t mhaq_fq_act_bwd_workspace_bytes(int64_t n);
int mhaq_fq_act_bwd(const float* x, const float* g, float* gx, int64_t n, const float* params /* [5] */,
                    int method, const int8_t* r_sign, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                    float* grads /* [3] */, void* workspace, size_t workspace_bytes, void* stream);

/* The two launches of mhaq_fq_act_bwd, separately, so that ONE finalize serves every activation quantizer of
 * a backward pass (the scalar gradients are only needed by the optimizer): *_partials runs the streaming
 * kernel, leaves `*nparts_out` partial rows in `workspace` and {s, qr} behind them; *_finalize_multi reduces
 * the workspaces of `nquant` such calls (device-resident table; a caller that keeps its workspaces alive
 * across steps uploads the table once) into grads_out[nquant][3] = {dL/dlog_act_s, dL/dlog_act_q, dL/dact_b}
 * per quantizer -- bit-identical to the per-quantizer finalize of mhaq_fq_act_bwd. */
typedef struct {
  const float* partials; /* the `workspace` of one mhaq_fq_act_bwd_partials call */
  int64_t nparts;        /* its *nparts_out */
} mhaq_act_finalize_desc;
int mhaq_fq_act_bwd_partials(const float* x, const float* g, float* gx, int64_t n, const float* params /* [5] */,
                             int method, const int8_t* r_sign, uint64_t seed, uint64_t offset,
                             const uint64_t* offset_dev, void* workspace, size_t workspace_bytes,
                             int32_t* nparts_out, void* stream);
int mhaq_fq_act_bwd_finalize_multi(const mhaq_act_finalize_desc* descs_device, int nquant,
                                   float* grads_out /* [nquant][3] */, void* stream);

/* Whole-tensor min / max (zero point of a PER_TENSOR weight quantizer,
 * gdnsq_conv2d.py:82-83; min/max observer, calib/minmaxobserver.py:19-36).
 * out[0] = min, out[1] = max. */
size_t mhaq_fq_minmax_workspace_bytes(int64_t n);
int mhaq_fq_minmax(const float* x, int64_t n, float* out /* [2] */,
                   void* workspace, size_t workspace_bytes, void* stream);

/* Per-row min / max of W viewed as [co][row] (weight-scale calibration, calib/minmaxobserver.py:73-75:
 * weight.amax((1,2,3)) / amin((1,2,3))); NaN-propagating like torch.  Read-only, one workgroup per row. */
int mhaq_fq_row_minmax(const float* w, int64_t co, int64_t row, float* mn_out /* [co] */,
                       float* mx_out /* [co] */, void* stream);

/* amin backward for a PER_TENSOR weight (tie-split scatter):
 * gw[i] += [w[i] == *zp] * grads[1] / grads[4]   with grads from mhaq_fq_pt_bwd. */
int mhaq_fq_pt_tie_scatter(const float* w, float* gw, int64_t n, const float* zp,
                           const float* grads /* [5] */, void* stream);

/* AEWGS statistics for a PER_TENSOR scale: per position j in [0,row) the means
 * over the `co` rows of sign(G*s)*e, e^2, e  -> stats[3][row]  (gdnsq.py:118-124).
 * lo / hi: clamp bounds (NULL = unbounded, the weight case).  Tall tensors are summed in row chunks
 * (fp64 partials in `workspace`, fixed-order finalize); _workspace_bytes is 0 when one chunk suffices. */
size_t mhaq_fq_pt_aewgs_colstats_workspace_bytes(int64_t co, int64_t row);
int mhaq_fq_pt_aewgs_colstats(const float* w, const float* G, int64_t co, int64_t row,
                              const float* s, const float* zp, const float* lo, const float* hi,
                              float* stats, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Per-channel weight fake-quant: W viewed as [co][row], one scale per row,
 * zero point = row minimum (gdnsq_conv2d.py:71-98).  One workgroup per
 * channel; the row is staged in LDS so HBM is read once.
 *
 * Forward:  zp[c] = min_j W[c][j];  Wq = q*s[c] + zp[c]  (q as above, no clamp)
 * ---------------------------------------------------------------------- */
int mhaq_fq_pc_fwd(const float* w, float* wq, float* zp_out /* [co] */, float* q_out /* nullable */,
                   const float* s /* [co] */, int64_t co, int64_t row, void* stream);

/* Backward given G = dL/dWq:
 *   gW = gv/s + [W == zp] * g_zp / count(W == zp)     (amin backward, tie split)
 *   g_s[c] = sum_c G*q - sum_c gv*(v/s) + noise_term
 * `stats` [3][co]: AEWGS group statistics (after the cross-rank all-reduce of
 * gdnsq.py:126-129); NULL = compute them in-kernel from this rank's data
 * (single-process semantics).  Ignored for other methods.
 * `gzp_extra` [co]: gradient reaching zp from other consumers of the zero point
 * (the quantized bias, gdnsq_conv2d.py:87); added before the tie split.  NULL = none. */
int mhaq_fq_pc_bwd(const float* w, const float* G, float* gw, float* g_s /* [co] */,
                   const float* s /* [co] */, const float* zp /* [co] */,
                   int64_t co, int64_t row, int method, const float* stats,
                   const float* gzp_extra /* nullable [co] */,
                   const int8_t* r_sign, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, void* stream);

/* The whole per-channel NoisyConv2d weight path of one layer from its LEARNABLE parameter
 * (gdnsq_conv2d.py:71-98) plus the layer's regulariser input of ModelHelper.get_model_values
 * (utils/model_helper.py:21-25,44), which shares the row min/max already in LDS:
 *   fwd:  s = exp2(log_s[c]);  zp = row min;  mx = row max;  wq as mhaq_fq_pc_fwd;
 *         lwq[c] = log2((mx - zp) + s)
 *   bwd:  as mhaq_fq_pc_bwd, plus the gradient g_lwq of lwq: t = g_lwq / (((mx-zp)+s) * ln2) goes
 *         +t to the row maxima and -t to the row minima of gw (amax / amin backward, tie split)
 *         and +t to s;  g_log_s[c] = (dL/ds) * s * ln2   (exp2 backward).  g_lwq may be NULL. */
int mhaq_fq_wlayer_fwd(const float* w, float* wq, const float* log_s /* [co] */, int64_t co, int64_t row,
                       float* s_out, float* zp_out, float* mx_out, float* lwq_out /* [co] each */,
                       void* stream);
int mhaq_fq_wlayer_bwd(const float* w, const float* G, float* gw, float* g_log_s /* [co] */,
                       const float* s, const float* zp, const float* mx, const float* g_lwq /* nullable */,
                       int64_t co, int64_t row, int method, const float* stats,
                       const float* gzp_extra, const int8_t* r_sign, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                       void* stream);

/* The same for a PER_TENSOR layer small enough (n <= mhaq_fq_wlayer_pt_max_elements() = 64 K: every
 * CIFAR ResNet-20 / RFDN layer) for one workgroup: one launch per direction instead of
 * minmax + finalize + forward (+ torch's exp2 / amin / amax / log2 chain).  aux[4] = {s, zp, max, lwq}
 * (written by fwd, read by bwd); g_log_s[1]; g_lwq points at one float or is NULL.
 * method: STE, LSQ or EWGS (AEWGS' per-position statistics keep the general path). */
int64_t mhaq_fq_wlayer_pt_max_elements(void);
int mhaq_fq_wlayer_pt_fwd(const float* w, float* wq, const float* log_s /* [1] */, int64_t n,
                          float* aux /* [4] */, void* stream);
int mhaq_fq_wlayer_pt_bwd(const float* w, const float* G, float* gw, float* g_log_s /* [1] */,
                          const float* aux, const float* g_lwq, int64_t n, int method,
                          const int8_t* r_sign, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, void* stream);

/* The same for a PER_TENSOR layer of ANY size (e.g. ResNet-18 with `qscheme: 0`: up to 2.36 M weights per layer; and
 * every PER_TENSOR AEWGS layer -- the reference's default NoisyConv2d arguments, gdnsq_conv2d.py:27-32), as streaming
 * launches: fwd = min / max sweep -> scalar chain -> quantizer (3 launches); bwd = streaming backward with both tie
 * counts -> fixed-order sums -> scalar chain -> tie-split scatter to the minima AND maxima (4 launches).  Replaces
 * gdnsq_conv2d.py:72,82-83,96-98 and, for the regulariser input, model_helper.py:36-37,44 (torch amin / amax over the
 * weight again, and their autograd scatter).
 *   aux[7] = {s, zp = min, max, lwq = log2((max - min) + s), -inf, +inf, the backward's hi}   (written by fwd, read by bwd)
 *   bwd: g_log_s[1] = (dL/ds + t) * s * ln2 with t = g_lwq / (((max - min) + s) * ln2);
 *        gw = gv/s + [w == min] * (dL/dzp - t) / count(min) + [w == max] * t / count(max);  g_lwq nullable [1].
 *   AEWGS: `col_stats` [3][period] from mhaq_fq_pt_aewgs_colstats(w, G, co, period, aux, aux + 1, NULL, NULL, ...)
 *        (after the cross-rank all-reduce under data parallelism); NULL / 0 for the other estimators.
 *   workspace: mhaq_fq_wlayer_ptl_workspace_bytes(n) for both directions. */
size_t mhaq_fq_wlayer_ptl_workspace_bytes(int64_t n);
int mhaq_fq_wlayer_ptl_fwd(const float* w, float* wq, const float* log_s /* [1] */, int64_t n, float* aux /* [7] */,
                           void* workspace, size_t workspace_bytes, void* stream);
int mhaq_fq_wlayer_ptl_bwd(const float* w, const float* G, float* gw, float* g_log_s /* [1] */,
                           const float* aux /* [7] */, const float* g_lwq /* nullable [1] */, int64_t n, int method,
                           const float* col_stats, int64_t period,
                           const int8_t* r_sign, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Multi-tensor variants: every PER_CHANNEL weight layer of a model in ONE launch per direction, driven by
 * a device-resident pointer table.  Layer L owns channels [chan_offset, chan_offset + co) of the
 * per-channel slabs and elements [elem_offset, elem_offset + co*row) of the element slabs; the grid has
 * total_co workgroups.  aux_all is [4][total_co] = {s, zp, max, lwq} rows (written by fwd, read by bwd).
 * bwd reads G / g_lwq through the table (they arrive as separate autograd tensors) and writes gw_all
 * (element slab) and g_log_s_all [total_co].  stats_all: NULL or [3][total_co] AEWGS statistics.
 * Random signs: element e of layer L uses index elem_offset + e of the (seed, offset) stream.
 * The table's layers are listed in ascending chan_offset; every pointer in it (like every pointer of this header)
 * addresses device-visible GLOBAL memory -- the kernels read the rows through the global address space. */
typedef struct {
  const float* w;      /* [co][row] */
  const float* log_s;  /* [co] */
  const float* G;      /* bwd only */
  const float* g_lwq;  /* bwd only, nullable */
  int64_t co, row;
  int64_t elem_offset;
  int64_t chan_offset;
} mhaq_wlayer_desc;
int mhaq_fq_wlayer_fwd_multi(const mhaq_wlayer_desc* descs_device, int nlayers, int64_t total_co,
                             int64_t max_row, float* wq_all, float* aux_all, void* stream);
int mhaq_fq_wlayer_bwd_multi(const mhaq_wlayer_desc* descs_device, int nlayers, int64_t total_co,
                             int64_t max_row, const float* aux_all, float* gw_all, float* g_log_s_all,
                             int method, const float* stats_all, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                             void* stream);

/* The same backward for a GROUP of consecutive layers -- what a data-parallel trainer wants: the weight
 * gradients of a group leave together as soon as its last dL/dWq has arrived (gradient all-reduce overlap is
 * kept, unlike the model-wide launch), and AEWGS exchanges ONE packed [3][group_co] message per group instead of
 * one per layer (gdnsq.py:126-129 issues three per layer).  `descs_device` holds the group's layers with
 * chan_offset / elem_offset RELATIVE to the group (first layer: 0); `aux` points at the group's first channel in
 * row 0 of the forward's [4][aux_stride] slab (aux_stride = the model-wide total_co of mhaq_fq_wlayer_fwd_multi,
 * or group_co for a slab of the group's own); gw [group elements], g_log_s [group_co] and stats [3][group_co]
 * (nullable) are the group's own.  Sign stream: element e of the group uses index e of (seed, offset).
 * _aewgs_stats_group writes stats[3][group_co] = {mean sign(G*s)*e, mean e^2, mean e} of every channel of the
 * group in one launch (reads G through the table). */
int mhaq_fq_wlayer_bwd_group(const mhaq_wlayer_desc* descs_device, int nlayers, int64_t group_co, int64_t max_row,
                             const float* aux, int64_t aux_stride, float* gw, float* g_log_s, int method,
                             const float* stats, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                             void* stream);
int mhaq_fq_wlayer_aewgs_stats_group(const mhaq_wlayer_desc* descs_device, int nlayers, int64_t group_co,
                                     const float* aux, int64_t aux_stride, float* stats /* [3][group_co] */,
                                     void* stream);

/* ABI v4 -- Quantizer.quantize (gdnsq.py:189-219) of a [co][row] tensor with GIVEN per-row scale and zero point (bounds
 * -inf / +inf as for every weight quantizer, gdnsq_conv2d.py:76-77): q_out = rounding indices, y_out (nullable) =
 * q * s + zp.  The stand-alone facade on a weight -- utils/model_stats.py:118,123 quantizes the detached weights with the
 * zero point the layer's last forward left in Q, which mhaq_fq_pc_fwd (zero point = the row minimum it finds) cannot do.
 * flags: nullable int32[1] ZEROED by the caller; the kernel ORs MHAQ_FQ_FLAG_NOT_INTEGER into it (the only eval assert of
 * gdnsq.py:211-217 that can fire with infinite bounds).  Any row length / 4-byte alignment. */
int mhaq_fq_pc_quantize(const float* x, float* q_out, float* y_out, const float* s /* [co] */,
                        const float* zp /* [co] */, int64_t co, int64_t row, int32_t* flags, void* stream);

/* AEWGS per-channel statistics -> stats[3][co] = {mean sign(G*s)*e, mean e^2, mean e}. */
int mhaq_fq_pc_aewgs_stats(const float* w, const float* G, const float* s, const float* zp,
                           int64_t co, int64_t row, float* stats, void* stream);

/* ------------------------------------------------------------------------
 * Per-element parameters: x[i] quantized with s[i], zp[i] (no clamp) -- the
 * quant_bias=True branch of gdnsq_conv2d.py:86-94, where the bias reuses the
 * weight's per-channel scale and zero point.  n = C_out (small).
 *   fwd:  y = q*s + zp                      (q_out nullable)
 *   bwd:  gx = gv/s;  g_s[i] = g*q - gv*(v/s) + noise;  g_zp[i] = g - gv/s
 * AEWGS needs `stats` [3] = means over all n elements (reduce_to_shape with no
 * unit dimension reduces everything, gdnsq.py:150-152) from *_vec_aewgs_stats.
 * ---------------------------------------------------------------------- */
int mhaq_fq_vec_fwd(const float* x, float* y, float* q_out, const float* s, const float* zp, int64_t n,
                    void* stream);
int mhaq_fq_vec_aewgs_stats(const float* x, const float* g, const float* s, const float* zp, int64_t n,
                            float* stats /* [3] */, void* stream);
int mhaq_fq_vec_bwd(const float* x, const float* g, float* gx, float* g_s, float* g_zp,
                    const float* s, const float* zp, int64_t n, int method, const float* stats,
                    const int8_t* r_sign, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, void* stream);

/* ------------------------------------------------------------------------
 * The reference's custom autograd Functions themselves, for callers that use
 * Quantizer.quantize / dequantize / _get_rnoise separately (the unfused
 * facade; utils/model_stats.py:116-132):
 *   noise_fwd   QNoise.forward (gdnsq.py:14-16):  out = rne(v) - v
 *   noise_bwd   QN{STE,LSQ,EWGS,AEWGS}.backward (gdnsq.py:35-147) given g = dL/dnoise:
 *               gv = estimator term, gs[group] = sum over the group of the
 *               scale-gradient terms.  `groups` scales own `len` consecutive
 *               elements each (1 = per-tensor, C_out = per-channel).
 *               AEWGS `stats`: [3][groups] (period == 0) or per position
 *               [3][period] (period > 0: the [1]-shaped-scale quirk), e.g. from
 *               mhaq_fq_pc_aewgs_stats / mhaq_fq_pt_aewgs_colstats with s = 1, zp = 0.
 * ---------------------------------------------------------------------- */
int mhaq_fq_noise_fwd(const float* v, float* out, int64_t n, void* stream);
size_t mhaq_fq_noise_bwd_workspace_bytes(int64_t groups, int64_t len);
int mhaq_fq_noise_bwd(const float* v, const float* g, float* gv, float* gs /* [groups] */,
                      int64_t groups, int64_t len, int method, const float* stats, int64_t period,
                      const int8_t* r_sign, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * PotentialLoss / PotentialLossNoPred arithmetic (gdnsq_loss.py:47-71, 129-153) over the concatenated
 * regulariser vectors las/laq [na] and lws/lwq [nw], one workgroup:
 *   hinge_w = max(0, (lwq-lws) - (w_bits - 1e-3))^p, hinge_a likewise with a_bits;
 *   ploss = (loss_sum/cnt * l1) * (wmul*mean hinge_w + amul*mean hinge_a) + l2 * base^p,
 *   (l1, l2) = (t, 1) or, lossless, (1, t);  wmul/amul from the counts of active hinges.
 *   The module state lives on the device: state[3] = {loss_sum, cnt, t} (running sum of the task loss, its
 *   step count, the temperature), so that a captured hipGraph sees their current values at every replay.
 *   fwd: out[12] = {ploss, wloss, aloss, rloss, cw, ca, cb, -mean lws, mean lwq, -mean las, mean laq,
 *        max(lwq-lws)};  update_state != 0 also does loss_sum += base^p, cnt += 1 (training mode).
 *   bwd: given g = dL/dploss [1] and `out` from fwd: g_base [1], g_las/g_laq [na], g_lws/g_lwq [nw].
 * ---------------------------------------------------------------------- */
int mhaq_fq_potential_loss_fwd(const float* base, const float* las, const float* laq, int64_t na,
                               const float* lws, const float* lwq, int64_t nw,
                               float a_bits, float w_bits, float p, int lossless,
                               float* state /* [3] device, in/out */, int update_state,
                               float* out /* [12] */, void* stream);
int mhaq_fq_potential_loss_bwd(const float* g, const float* out, const float* las, const float* laq, int64_t na,
                               const float* lws, const float* lwq, int64_t nw,
                               float a_bits, float w_bits, float p,
                               float* g_base, float* g_las, float* g_laq, float* g_lws, float* g_lwq,
                               void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MHAQ_FQ_H */
