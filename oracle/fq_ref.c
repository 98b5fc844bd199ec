/* oracle/fq_ref.c -- TEST INFRASTRUCTURE ONLY: a plain-C, scalar restatement of the reference's fake-quant path,
 * independent of torch (the eager oracle in fq_eager.py leans on the same aten ops as the reference).  Only tests/ may
 * load it (oracle/fq_c.py); nothing under mhaq_amd/ does.  Built by oracle/Makefile (gcc -O2 -ffp-contract=off: fp32
 * operations stay separate, like the reference's eager kernels) into oracle/libfq_ref.so; pinned by the vectors
 * recorded from the real reference (tests/golden/, tests/test_oracle_c_golden.py).
 *
 * What it follows, line for line:
 *   Quantizer.quantize / dequantize        /root/reference/src/quantization/gdnsq/gdnsq.py:189-229
 *   QNoise.forward, QN{STE,LSQ,EWGS,AEWGS}.backward, reduce_to_shape      gdnsq.py:14-152
 *   NoisyAct.forward                       layers/gdnsq_act.py:39-55
 *   NoisyConv2d / NoisyLinear weight path  layers/gdnsq_conv2d.py:71-98, layers/gdnsq_linear.py:61-78
 * The backward is the autograd graph of those lines written out (closed forms K1 / K2):
 * elementwise results in fp32 in the reference's operation order, reductions in double (the reference sums in fp32).
 * The scales arrive exponentiated (s = 2^log_s from the caller's exp2: libm's exp2f may differ from torch's in the
 * last bit, and the fixtures pin the bits of y).  r: +1 / -1 per element, the sign of the reference's
 * randint_like(v, 2) - 0.5 (gdnsq.py:54).  method: 0 STE, 1 EWGS (as intended; the reference raises at gdnsq.py:102 --
 * pinned by tests/golden/ewgs_*.npz, recorded from the reference's own lines with the misspelled attribute supplied), 2 AEWGS, 3 LSQ. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define INV_SQRT3_F 0.57735026918962584f /* (float)(3 ** -0.5) */

static float clampf(float x, float lo, float hi) { /* torch.clamp(x, min=lo, max=hi) = min(max(x, lo), hi); NaN stays */
  if (x != x) return x;
  float t = x < lo ? lo : x;
  return t > hi ? hi : t;
}

static float signf_(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

/* QN*.backward grad_input for one element (gdnsq.py:52,81,101,136-141): the estimator's d(noise)/dv term. */
static float noise_grad_v(int method, float gq, float e, float delta) {
  if (method == 1) return -fabsf(gq) * e * 0.01f;
  if (method == 2) {
    float num_full = signf_(gq) * e;
    float gsc = 1.0f * delta * num_full;
    if (gsc == gsc && gsc > 0.99f) gsc = 0.99f; /* clamp_max(1 - 0.01) keeps NaN */
    return -gq * gsc;
  }
  return gq * 0.f;
}

/* AEWGS statistics (gdnsq.py:118-124, 131-134): delta = num / max(e2 - me^2, 1e-3) per group; the three means are
 * taken over `count` elements starting at `first` with stride `stride`. */
static float aewgs_delta(const float* gq, const float* e, int64_t first, int64_t stride, int64_t count) {
  double num = 0, e2 = 0, me = 0;
  for (int64_t k = 0; k < count; ++k) {
    const int64_t i = first + k * stride;
    num += (double)(signf_(gq[i]) * e[i]);
    e2 += (double)(e[i] * e[i]);
    me += (double)e[i];
  }
  const float fnum = (float)(num / (double)count), fe2 = (float)(e2 / (double)count), fme = (float)(me / (double)count);
  float den = fe2 - fme * fme;
  if (den < 1e-3f) den = 1e-3f;
  return fnum / den;
}

/* NoisyAct forward + backward (gdnsq_act.py:39-55).  x, g: [n]; d0 = size of dim 0 (the AEWGS statistics of a
 * [1]-shaped scale are per position over dim 0 only: reduce_to_shape, gdnsq.py:150-152).
 * out: y, q, gx [n]; grads[3] = {dL/dlog_act_s, dL/dlog_act_q, dL/dact_b}. */
int mhaq_ref_act(const float* x, const float* g, const int8_t* r, int64_t n, int64_t d0, float s, float qr, float b,
                 int method, float* y, float* q, float* gx, double* grads) {
  const float lo = b, hi = (b + qr) - s, zp = b;
  float* e = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  float* gq = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  float* v = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  if (!e || !gq || !v) return 1;
  for (int64_t i = 0; i < n; ++i) {
    const float v0 = clampf(x[i], lo, hi);
    const float v1 = v0 - zp;
    v[i] = v1 / s;
    e[i] = rintf(v[i]) - v[i]; /* QNoise.forward: round half to even */
    q[i] = v[i] + e[i];
    const float t = q[i] * s;
    y[i] = t + zp;
    gq[i] = g[i] * s; /* dequantize backward */
  }
  const int64_t row = d0 > 0 ? n / d0 : n;
  double t_s = 0, t_zp = 0, t_lo = 0, t_hi = 0;
  for (int64_t i = 0; i < n; ++i) {
    float delta = 0.f;
    if (method == 2) delta = aewgs_delta(gq, e, i % row, row, d0);
    const float gv = gq[i] + noise_grad_v(method, gq[i], e[i], delta);
    const float g1 = gv / s;
    const float noise_s = (method == 3) ? gq[i] * e[i] : (INV_SQRT3_F * gq[i]) * (r[i] > 0 ? 0.5f : -0.5f);
    t_s += (double)((g[i] * q[i] + (-gv) * (v[i] / s)) + noise_s);
    t_zp += (double)(g[i] - g1);
    const int lt = x[i] < lo, gt = x[i] > hi;
    if (lt && lo < hi) t_lo += (double)g1; /* clamp backward: the bound takes the gradient of what it clips ... */
    if (gt || hi < lo) t_hi += (double)g1; /* ... and max wins everything when the bounds are inverted */
    gx[i] = (x[i] >= lo && x[i] <= hi) ? g1 : 0.f;
  }
  const double ln2 = 0.69314718055994530942;
  grads[0] = (t_s - t_hi) * (double)s * ln2; /* s enters the noise path, the divide, the dequantize and hi = b + qr - s */
  grads[1] = t_hi * (double)qr * ln2;
  grads[2] = t_zp + t_lo + t_hi;
  free(e); free(gq); free(v);
  return 0;
}

/* NoisyConv2d / NoisyLinear weight path (gdnsq_conv2d.py:71-98): w viewed as [co][row]; per_channel: one scale and
 * one zero point (the row minimum, differentiable: amin backward splits among ties) per row; otherwise one for the
 * whole tensor, with AEWGS statistics per position over dim 0 (the [1]-shaped-scale quirk).
 * s: [co] or [1] (exponentiated).  out: wq, gw [co*row]; zp, g_log_s: [co] or [1]. */
int mhaq_ref_weight(const float* w, const float* G, const int8_t* r, int64_t co, int64_t row, const float* s,
                    int per_channel, int method, float* wq, float* zp, float* gw, double* g_log_s) {
  const int64_t n = co * row;
  const int64_t groups = per_channel ? co : 1, len = per_channel ? row : n;
  float* e = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  float* gq = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  float* v = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  float* q = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  if (!e || !gq || !v || !q) return 1;
  for (int64_t c = 0; c < groups; ++c) {
    const float* wr = w + c * len;
    float mn = wr[0];
    for (int64_t j = 1; j < len; ++j) mn = (wr[j] < mn || wr[j] != wr[j]) ? wr[j] : mn;
    zp[c] = mn;
    const float sc = s[c];
    for (int64_t j = 0; j < len; ++j) {
      const int64_t i = c * len + j;
      const float v1 = w[i] - mn; /* no clamp for weights: min_val / max_val are -inf / +inf */
      v[i] = v1 / sc;
      e[i] = rintf(v[i]) - v[i];
      q[i] = v[i] + e[i];
      const float t = q[i] * sc;
      wq[i] = t + mn;
      gq[i] = G[i] * sc;
    }
  }
  const double ln2 = 0.69314718055994530942;
  for (int64_t c = 0; c < groups; ++c) {
    const float sc = s[c];
    double t_s = 0, sum_g = 0, sum_gvs = 0;
    int64_t ties = 0;
    float delta_c = 0.f;
    if (method == 2 && per_channel) delta_c = aewgs_delta(gq, e, c * len, 1, len);
    for (int64_t j = 0; j < len; ++j) {
      const int64_t i = c * len + j;
      float delta = delta_c;
      if (method == 2 && !per_channel) delta = aewgs_delta(gq, e, i % row, row, co);
      const float gv = gq[i] + noise_grad_v(method, gq[i], e[i], delta);
      const float gvs = gv / sc;
      const float noise_s = (method == 3) ? gq[i] * e[i] : (INV_SQRT3_F * gq[i]) * (r[i] > 0 ? 0.5f : -0.5f);
      t_s += (double)((G[i] * q[i] + (-gv) * (v[i] / sc)) + noise_s);
      sum_g += (double)G[i];
      sum_gvs += (double)gvs;
      ties += (w[i] == zp[c]);
      gw[i] = gvs;
    }
    const float g_zp = (float)(sum_g - sum_gvs); /* +zp in dequantize, -zp before the divide */
    const float share = (g_zp * 1.0f) / (float)ties;
    for (int64_t j = 0; j < len; ++j) {
      const int64_t i = c * len + j;
      if (w[i] == zp[c]) gw[i] = gw[i] + share; /* amin backward */
    }
    g_log_s[c] = t_s * (double)sc * ln2;
  }
  free(e); free(gq); free(v); free(q);
  return 0;
}

/* PotentialLoss / PotentialLossNoPred value (gdnsq_loss.py:47-71, 129-153) over the regulariser vectors of
 * ModelHelper.get_model_values (utils/model_helper.py:13-76): hinge_w = max(0, (lwq - lws) - (w_bits - 1e-3))^p, hinge_a
 * likewise; ploss = (loss_sum / cnt) * l1 * (wmul * mean hinge_w + amul * mean hinge_a) + l2 * base^p with (l1, l2) =
 * (t, 1), or (1, t) when lossless; wmul / amul from the counts of active hinges.  Returns ploss; *rloss = base^p. */
double mhaq_ref_potential_loss(double base, const float* las, const float* laq, int64_t na, const float* lws,
                               const float* lwq, int64_t nw, double a_bits, double w_bits, double p, int lossless,
                               double t, double loss_sum, double cnt, double* rloss) {
  const double eps = 1e-3;
  double wsum = 0, asum = 0;
  int64_t wact = 0, aact = 0;
  for (int64_t i = 0; i < nw; ++i) {
    const double h = (double)(lwq[i] - lws[i]) - (w_bits - eps);
    const double v = h > 0 ? pow(h, p) : 0.0;
    wsum += v;
    wact += v > 0;
  }
  for (int64_t i = 0; i < na; ++i) {
    const double h = (double)(laq[i] - las[i]) - (a_bits - eps);
    const double v = h > 0 ? pow(h, p) : 0.0;
    asum += v;
    aact += v > 0;
  }
  const double wloss = nw ? wsum / (double)nw : 0.0, aloss = na ? asum / (double)na : 0.0;
  const double wmul = ((double)wact + eps) / ((double)(wact + aact) + eps);
  const double amul = ((double)aact + eps) / ((double)(wact + aact) + eps);
  const double r = pow(base, p);
  if (rloss) *rloss = r;
  const double l1 = lossless ? 1.0 : t, l2 = lossless ? t : 1.0;
  return (loss_sum / cnt) * l1 * (wmul * wloss + amul * aloss) + l2 * r;
}

/* The weight half of ModelHelper.get_model_values (model_helper.py:21-25, 44): lwq[c] = log2(max - min + 2^log_s[c])
 * per output channel (per_channel) or for the whole tensor. */
int mhaq_ref_regulariser_input(const float* w, int64_t co, int64_t row, const float* log_s, int per_channel,
                               float* lwq) {
  const int64_t groups = per_channel ? co : 1, len = per_channel ? row : co * row;
  for (int64_t c = 0; c < groups; ++c) {
    float mn = w[c * len], mx = w[c * len];
    for (int64_t j = 1; j < len; ++j) {
      const float v = w[c * len + j];
      mn = v < mn ? v : mn;
      mx = v > mx ? v : mx;
    }
    lwq[c] = log2f((mx - mn) + exp2f(log_s[c]));
  }
  return 0;
}
