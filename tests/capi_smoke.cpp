// Torch-free consumer of the C ABI (include/mhaq_fq.h): plain HIP runtime + libmhaq_fq.so.
// Runs the activation fake-quant forward and backward on a ragged tensor and checks them against a scalar
// host restatement of gdnsq.py:189-229 (built with -ffp-contract=off).  Exit code 0 = parity.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../include/mhaq_fq.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s (line %d)\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

int main() {
  const int64_t n = 100003;  // not a multiple of 4: exercises the scalar tail
  std::vector<float> x(n), g(n), y(n), gx(n);
  uint32_t st = 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)(st >> 8) / 16777216.0f; };
  for (int64_t i = 0; i < n; ++i) { x[i] = rnd() * 8.f - 4.f; g[i] = rnd() * 2.f - 1.f; }
  const float log_s = -3.7f, log_q = 2.3f, b = -2.6f;
  float *dx, *dg, *dy, *dgx, *dp, *dparams, *dgrads;
  void* ws;
  CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dg, n * 4)); CK(hipMalloc(&dy, n * 4)); CK(hipMalloc(&dgx, n * 4));
  CK(hipMalloc(&dp, 12)); CK(hipMalloc(&dparams, 20)); CK(hipMalloc(&dgrads, 12));
  const size_t wsb = mhaq_fq_act_bwd_workspace_bytes(n);
  CK(hipMalloc(&ws, wsb));
  const float hp[3] = {log_s, log_q, b};
  CK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dg, g.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dp, hp, 12, hipMemcpyHostToDevice));
  if (mhaq_fq_abi_version() != MHAQ_FQ_ABI_VERSION) { printf("ABI mismatch\n"); return 1; }
  int rc = mhaq_fq_act_fwd(dx, dy, n, dp, dp + 1, dp + 2, dparams, nullptr, nullptr, nullptr, 0, nullptr);
  if (rc) { printf("act_fwd: %s\n", mhaq_fq_error_string(rc)); return 1; }
  rc = mhaq_fq_act_bwd(dx, dg, dgx, n, dparams, MHAQ_FQ_LSQ, nullptr, 0, 0, nullptr, dgrads, ws, wsb, nullptr);
  if (rc) { printf("act_bwd: %s\n", mhaq_fq_error_string(rc)); return 1; }
  CK(hipDeviceSynchronize());
  float params[5], grads[3];
  CK(hipMemcpy(y.data(), dy, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(gx.data(), dgx, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(params, dparams, 20, hipMemcpyDeviceToHost));
  CK(hipMemcpy(grads, dgrads, 12, hipMemcpyDeviceToHost));
  // host restatement with the device's own scale bits (host exp2f may differ by 1 ulp)
  const float s = params[0], zp = params[1], lo = params[2], hi = params[3], qr = params[4];
  if (std::fabs(s - std::exp2(log_s)) > 2e-7f * s || std::fabs(qr - std::exp2(log_q)) > 2e-7f * qr || zp != b ||
      lo != b || hi != (b + qr) - s) { printf("parameter block wrong\n"); return 1; }
  int64_t bad = 0;
  double gs = 0, ghi = 0, glo = 0, gzp = 0, yard = 0;
  for (int64_t i = 0; i < n; ++i) {
    float v0 = std::fmin(std::fmax(x[i], lo), hi);
    volatile float v1 = v0 - zp;
    volatile float v = v1 / s;
    volatile float e = std::nearbyint(v) - v;
    volatile float q = v + e;
    volatile float t = q * s;
    const float yy = t + zp;
    volatile float gq = g[i] * s;
    const float g1 = gq / s;
    const bool in = x[i] >= lo && x[i] <= hi;
    const float gxx = in ? g1 : 0.f;
    bad += (std::memcmp(&yy, &y[i], 4) != 0) + (gxx != gx[i]);
    gs += (double)g[i] * e + (double)gq * e;       // LSQ: g*(q - v) + gq*e
    yard += std::fabs((double)g[i] * q) * 2;
    gzp += (double)g[i] - g1;
    if (x[i] < lo) glo += g1;
    if (x[i] > hi) ghi += g1;
  }
  const double ln2 = 0.6931471805599453;
  const double want_ls = (gs - ghi) * s * ln2, want_lq = ghi * qr * ln2, want_b = gzp + glo + ghi;
  const bool ok = bad == 0 && std::fabs(grads[0] - want_ls) <= 1e-6 * yard * s * ln2 &&
                  std::fabs(grads[1] - want_lq) <= 1e-6 * yard * qr * ln2 && std::fabs(grads[2] - want_b) <= 1e-6 * yard;
  printf("capi_smoke: n=%lld mismatching elements=%lld  dlog_s %.6g (want %.6g)  dlog_q %.6g (want %.6g)  db %.6g (want %.6g)  -> %s\n",
         (long long)n, (long long)bad, grads[0], want_ls, grads[1], want_lq, grads[2], want_b, ok ? "OK" : "FAIL");
  if (!ok) return 1;

  // ---- ABI v2: (a) the split backward -- partials + ONE finalize for several quantizers -- gives the bits of
  // mhaq_fq_act_bwd; (b) STE with host offset o and *offset_dev = d draws the stream of offset o + d.
  void *ws2, *ws3;
  float *dgx2, *dgrads2, *dmulti;
  uint64_t* doff;
  CK(hipMalloc(&ws2, wsb)); CK(hipMalloc(&ws3, wsb)); CK(hipMalloc(&dgx2, n * 4)); CK(hipMalloc(&dgrads2, 12));
  CK(hipMalloc(&dmulti, 2 * 12)); CK(hipMalloc(&doff, 8));
  const uint64_t seed = 0x1234abcdu, d_off = 5;
  CK(hipMemcpy(doff, &d_off, 8, hipMemcpyHostToDevice));
  rc = mhaq_fq_act_bwd(dx, dg, dgx, n, dparams, MHAQ_FQ_STE, nullptr, seed, 12, nullptr, dgrads, ws, wsb, nullptr);
  if (rc) { printf("act_bwd STE: %s\n", mhaq_fq_error_string(rc)); return 1; }
  int32_t np2 = 0, np3 = 0;
  rc = mhaq_fq_act_bwd_partials(dx, dg, dgx2, n, dparams, MHAQ_FQ_STE, nullptr, seed, 7, doff, ws2, wsb, &np2, nullptr);
  if (rc) { printf("act_bwd_partials: %s\n", mhaq_fq_error_string(rc)); return 1; }
  rc = mhaq_fq_act_bwd_partials(dx, dg, dgx2, n, dparams, MHAQ_FQ_LSQ, nullptr, 0, 0, nullptr, ws3, wsb, &np3, nullptr);
  if (rc) { printf("act_bwd_partials: %s\n", mhaq_fq_error_string(rc)); return 1; }
  mhaq_act_finalize_desc hd[2] = {{(const float*)ws2, np2}, {(const float*)ws3, np3}};
  mhaq_act_finalize_desc* dd;
  CK(hipMalloc(&dd, sizeof(hd)));
  CK(hipMemcpy(dd, hd, sizeof(hd), hipMemcpyHostToDevice));
  rc = mhaq_fq_act_bwd_finalize_multi(dd, 2, dmulti, nullptr);
  if (rc) { printf("finalize_multi: %s\n", mhaq_fq_error_string(rc)); return 1; }
  CK(hipDeviceSynchronize());
  float ste[3], multi[6];
  CK(hipMemcpy(ste, dgrads, 12, hipMemcpyDeviceToHost));
  CK(hipMemcpy(multi, dmulti, 24, hipMemcpyDeviceToHost));
  const bool same_stream = std::memcmp(ste, multi, 12) == 0;          // offset 12 == 7 + *offset_dev (5)
  const bool same_lsq = std::memcmp(grads, multi + 3, 12) == 0;       // the LSQ run of the first part
  // ---- (c) per-row min / max
  const int64_t co = 37, row = 2700;                                 // 99900 <= n elements of x as a [37][2700] weight
  float *dmn, *dmx;
  CK(hipMalloc(&dmn, co * 4)); CK(hipMalloc(&dmx, co * 4));
  rc = mhaq_fq_row_minmax(dx, co, row, dmn, dmx, nullptr);
  if (rc) { printf("row_minmax: %s\n", mhaq_fq_error_string(rc)); return 1; }
  std::vector<float> mn(co), mx(co);
  CK(hipMemcpy(mn.data(), dmn, co * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(mx.data(), dmx, co * 4, hipMemcpyDeviceToHost));
  int64_t badrow = 0;
  for (int64_t c = 0; c < co; ++c) {
    float a = x[c * row], b2 = x[c * row];
    for (int64_t j = 1; j < row; ++j) { a = std::fmin(a, x[c * row + j]); b2 = std::fmax(b2, x[c * row + j]); }
    badrow += (a != mn[c]) + (b2 != mx[c]);
  }
  const bool ok2 = same_stream && same_lsq && badrow == 0;
  printf("capi_smoke v2: split backward == fused %d, offset + *offset_dev selects the stream %d, row_minmax mismatches %lld -> %s\n",
         (int)same_lsq, (int)same_stream, (long long)badrow, ok2 ? "OK" : "FAIL");
  if (!ok2) return 1;

  // ---- (d) the weight path a data-parallel trainer runs: ONE model-wide forward launch over a device pointer table,
  // then the backward of layers 1..2 as a GROUP (window of the forward's aux slab, group-relative offsets): the bits of
  // the per-layer entry points, the sign stream of the group taken at the layers' element offsets in the group.
  const int64_t cos[3] = {5, 8, 3}, rows[3] = {36, 100, 27};            // x viewed as three [co][row] weights
  int64_t eoff[3], coff[3], te = 0, tc = 0;
  for (int l = 0; l < 3; ++l) { eoff[l] = te; coff[l] = tc; te += cos[l] * rows[l]; tc += cos[l]; }
  float *dls, *dwq, *daux, *dgw_g, *dgls_g, *dgw_l, *dgls_l;
  CK(hipMalloc(&dls, tc * 4)); CK(hipMalloc(&dwq, te * 4)); CK(hipMalloc(&daux, 4 * tc * 4));
  CK(hipMalloc(&dgw_g, te * 4)); CK(hipMalloc(&dgls_g, tc * 4)); CK(hipMalloc(&dgw_l, te * 4)); CK(hipMalloc(&dgls_l, tc * 4));
  std::vector<float> hls(tc);
  for (int64_t c = 0; c < tc; ++c) hls[c] = -4.f - 0.1f * (float)(c % 7);
  CK(hipMemcpy(dls, hls.data(), tc * 4, hipMemcpyHostToDevice));
  mhaq_wlayer_desc fd[3], gd[2];
  for (int l = 0; l < 3; ++l) fd[l] = {dx + eoff[l], dls + coff[l], nullptr, nullptr, cos[l], rows[l], eoff[l], coff[l]};
  for (int l = 1; l < 3; ++l)
    gd[l - 1] = {dx + eoff[l], nullptr, dg + eoff[l], nullptr, cos[l], rows[l], eoff[l] - eoff[1], coff[l] - coff[1]};
  mhaq_wlayer_desc *dfd, *dgd;
  CK(hipMalloc(&dfd, sizeof(fd))); CK(hipMalloc(&dgd, sizeof(gd)));
  CK(hipMemcpy(dfd, fd, sizeof(fd), hipMemcpyHostToDevice));
  CK(hipMemcpy(dgd, gd, sizeof(gd), hipMemcpyHostToDevice));
  rc = mhaq_fq_wlayer_fwd_multi(dfd, 3, tc, 100, dwq, daux, nullptr);
  if (rc) { printf("wlayer_fwd_multi: %s\n", mhaq_fq_error_string(rc)); return 1; }
  const int64_t gco = cos[1] + cos[2], gel = cos[1] * rows[1] + cos[2] * rows[2];
  rc = mhaq_fq_wlayer_bwd_group(dgd, 2, gco, 100, daux + coff[1], tc, dgw_g, dgls_g, MHAQ_FQ_STE, nullptr, seed, 3,
                                nullptr, nullptr);
  if (rc) { printf("wlayer_bwd_group: %s\n", mhaq_fq_error_string(rc)); return 1; }
  int8_t* dr;
  CK(hipMalloc(&dr, gel));
  rc = mhaq_fq_fill_r(dr, gel, seed, 3, nullptr);
  if (rc) { printf("fill_r: %s\n", mhaq_fq_error_string(rc)); return 1; }
  for (int l = 1; l < 3; ++l) {
    const int64_t e0 = eoff[l] - eoff[1], c0 = coff[l] - coff[1];
    rc = mhaq_fq_wlayer_bwd(dx + eoff[l], dg + eoff[l], dgw_l + e0, dgls_l + c0, daux + coff[l], daux + tc + coff[l],
                            daux + 2 * tc + coff[l], nullptr, cos[l], rows[l], MHAQ_FQ_STE, nullptr, nullptr, dr + e0,
                            0, 0, nullptr, nullptr);
    if (rc) { printf("wlayer_bwd: %s\n", mhaq_fq_error_string(rc)); return 1; }
  }
  CK(hipDeviceSynchronize());
  std::vector<float> a(gel), b3(gel), c3(gco), d3(gco);
  CK(hipMemcpy(a.data(), dgw_g, gel * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b3.data(), dgw_l, gel * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(c3.data(), dgls_g, gco * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(d3.data(), dgls_l, gco * 4, hipMemcpyDeviceToHost));
  const bool ok3 = std::memcmp(a.data(), b3.data(), gel * 4) == 0 && std::memcmp(c3.data(), d3.data(), gco * 4) == 0;
  printf("capi_smoke groups: grouped weight backward == per-layer backward (gW %lld floats, dlog_s %lld) -> %s\n",
         (long long)gel, (long long)gco, ok3 ? "OK" : "FAIL");
  if (!ok3) return 1;

  // ---- (e) a PER_TENSOR weight layer of any size (mhaq_fq_wlayer_ptl_fwd / _bwd): x as one weight tensor of n elements,
  // log_s = -3 (exp2 exact), LSQ, with a regulariser gradient; against a scalar host restatement of gdnsq_conv2d.py:71-98
  // + model_helper.py:36-37,44.  Two elements are tied at the minimum and two at the maximum.
  std::vector<float> xw(x.begin(), x.begin() + n);
  float wmn = xw[0], wmx = xw[0];
  for (int64_t i = 1; i < n; ++i) { wmn = std::fmin(wmn, xw[i]); wmx = std::fmax(wmx, xw[i]); }
  xw[3] = xw[n - 2] = wmn - 0.25f;
  xw[5] = xw[n / 2] = wmx + 0.5f;
  wmn -= 0.25f; wmx += 0.5f;
  float *dw, *dwq2, *dgw2, *daux7, *dlogs, *dgls1, *dglwq;
  CK(hipMalloc(&dw, n * 4)); CK(hipMalloc(&dwq2, n * 4)); CK(hipMalloc(&dgw2, n * 4)); CK(hipMalloc(&daux7, 7 * 4));
  CK(hipMalloc(&dlogs, 4)); CK(hipMalloc(&dgls1, 4)); CK(hipMalloc(&dglwq, 4));
  CK(hipMemcpy(dw, xw.data(), n * 4, hipMemcpyHostToDevice));
  const float logs = -3.f, glwq = 0.75f;
  CK(hipMemcpy(dlogs, &logs, 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dglwq, &glwq, 4, hipMemcpyHostToDevice));
  const size_t pb = mhaq_fq_wlayer_ptl_workspace_bytes(n);
  void* pws; CK(hipMalloc(&pws, pb));
  rc = mhaq_fq_wlayer_ptl_fwd(dw, dwq2, dlogs, n, daux7, pws, pb, nullptr);
  if (rc) { printf("wlayer_ptl_fwd: %s\n", mhaq_fq_error_string(rc)); return 1; }
  rc = mhaq_fq_wlayer_ptl_bwd(dw, dg, dgw2, dgls1, daux7, dglwq, n, MHAQ_FQ_LSQ, nullptr, 0, nullptr, 0, 0, nullptr, pws, pb,
                              nullptr);
  if (rc) { printf("wlayer_ptl_bwd: %s\n", mhaq_fq_error_string(rc)); return 1; }
  CK(hipDeviceSynchronize());
  std::vector<float> wq2(n), gw2(n);
  float aux7[7], gls1;
  CK(hipMemcpy(wq2.data(), dwq2, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(gw2.data(), dgw2, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(aux7, daux7, 28, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&gls1, dgls1, 4, hipMemcpyDeviceToHost));
  const float sw = 0.125f;
  int64_t bad_wq = 0, bad_gw = 0;
  double ws_gs = 0, ws_gzp = 0, abs_s = 0, abs_g = 0;
  int cmin = 0, cmax = 0;
  for (int64_t i = 0; i < n; ++i) {
    const float v = (xw[i] - wmn) / sw, q = v + (std::nearbyintf(v) - v);
    bad_wq += (q * sw + wmn) != wq2[i];
    const float gq = g[i] * sw, gvs = gq / sw;
    ws_gs += (double)g[i] * (double)(q - v) + (double)gq * (double)(q - v);       // LSQ: g*(q - v) + noise term gq*(q - v)
    ws_gzp += (double)g[i] - (double)gvs;
    abs_s += std::fabs((double)g[i] * q) * 2 + std::fabs((double)gq);
    abs_g += std::fabs((double)g[i]) * 2;
    cmin += xw[i] == wmn;
    cmax += xw[i] == wmx;
    if (xw[i] != wmn && xw[i] != wmx) bad_gw += gvs != gw2[i];
  }
  const double u = ((double)wmx - (double)wmn) + sw, t = glwq / (u * 0.6931471805599453);
  const double want_gls = (ws_gs + t) * sw * 0.6931471805599453;
  const double tie_min = (ws_gzp - t) / cmin, tie_max = t / cmax;
  const float lwq_host = std::log2f((wmx - wmn) + sw);      // the device's log2f may differ from the host's by an ulp
  const bool aux_ok = aux7[0] == sw && aux7[1] == wmn && aux7[2] == wmx && std::fabs(aux7[3] - lwq_host) <= 4e-7f * std::fabs(lwq_host);
  const double e_min = std::fabs((double)gw2[3] - ((double)((g[3] * sw) / sw) + tie_min));
  const double e_max = std::fabs((double)gw2[5] - ((double)((g[5] * sw) / sw) + tie_max));
  const double e_gls = std::fabs((double)gls1 - want_gls);
  const bool ok4 = bad_wq == 0 && bad_gw == 0 && aux_ok && cmin == 2 && cmax == 2 && e_min <= 1e-6 * (abs_g + 1) &&
                   e_max <= 1e-6 * (abs_g + 1) && e_gls <= 1e-6 * (abs_s + 1) * sw;
  printf("capi_smoke per-tensor layer: wq mismatches %lld, gW mismatches off the extremes %lld, aux %d, ties %d/%d, "
         "|d tie_min| %.2e |d tie_max| %.2e |d dlog_s| %.2e -> %s\n", (long long)bad_wq, (long long)bad_gw, (int)aux_ok, cmin, cmax,
         e_min, e_max, e_gls, ok4 ? "OK" : "FAIL");
  if (!ok4) return 1;

  // ---- (f) ABI v3: the sign stream as include/mhaq_fq.h documents it, restated here from the header's text alone: one
  // Philox4x32-10 call per 128 consecutive elements, counter = {lo(c), hi(c), lo(offset), hi(offset)}, key = {lo(seed),
  // hi(seed)}, element i = bit (i & 31) of output word ((i & 127) >> 5) of call c = i >> 7.  mhaq_fq_fill_r must write it.
  {
    const int64_t nr = 4 * 2048 + 1027 + 3;
    const uint64_t sd = 0x9E3779B97F4A7C15ull ^ 2024ull, of = (1ull << 40) + 7;
    int8_t* drr;
    CK(hipMalloc(&drr, nr));
    rc = mhaq_fq_fill_r(drr, nr, sd, of, nullptr);
    if (rc) { printf("fill_r: %s\n", mhaq_fq_error_string(rc)); return 1; }
    std::vector<int8_t> got(nr);
    CK(hipMemcpy(got.data(), drr, nr, hipMemcpyDeviceToHost));
    int64_t badr = 0;
    for (int64_t c = 0; c * 128 < nr; ++c) {
      uint32_t c0 = (uint32_t)c, c1 = (uint32_t)((uint64_t)c >> 32), c2 = (uint32_t)of, c3 = (uint32_t)(of >> 32);
      uint32_t k0 = (uint32_t)sd, k1 = (uint32_t)(sd >> 32);
      for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
      }
      const uint32_t w[4] = {c0, c1, c2, c3};
      for (int j = 0; j < 128 && c * 128 + j < nr; ++j) {
        const int8_t want = ((w[j >> 5] >> (j & 31)) & 1u) ? 1 : -1;
        badr += got[c * 128 + j] != want;
      }
    }
    printf("capi_smoke signs: mhaq_fq_fill_r == the header's Philox layout restated on the host (%lld elements, %lld differ) -> %s\n",
           (long long)nr, (long long)badr, badr == 0 ? "OK" : "FAIL");
    if (badr) return 1;
  }
  return 0;
}
