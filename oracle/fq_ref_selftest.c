/* TEST INFRASTRUCTURE: drives oracle/fq_ref.c under AddressSanitizer + UBSan (tests/test_oracle_c_golden.py builds
 * it with -fsanitize=address,undefined and runs it on the CPU): every estimator, both schemes, ragged sizes, a batch of
 * one, NaN / inf inputs -- no out-of-bounds access, no undefined arithmetic, finite outputs for finite inputs. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int mhaq_ref_act(const float*, const float*, const int8_t*, int64_t, int64_t, float, float, float, int, float*, float*,
                 float*, double*);
int mhaq_ref_weight(const float*, const float*, const int8_t*, int64_t, int64_t, const float*, int, int, float*, float*,
                    float*, double*);
double mhaq_ref_potential_loss(double, const float*, const float*, int64_t, const float*, const float*, int64_t, double,
                               double, double, int, double, double, double, double*);
int mhaq_ref_regulariser_input(const float*, int64_t, int64_t, const float*, int, float*);

static uint32_t st = 12345u;
static float rnd(void) { st = st * 1664525u + 1013904223u; return (float)(st >> 8) / 16777216.0f; }

int main(void) {
  int bad = 0;
  const int64_t shapes[][2] = {{1, 1}, {1, 7}, {3, 5}, {4, 36}, {8, 101}, {2, 1024}};
  for (int sh = 0; sh < 6; ++sh) {
    const int64_t d0 = shapes[sh][0], row = shapes[sh][1], n = d0 * row;
    float *x = malloc(n * 4), *g = malloc(n * 4), *y = malloc(n * 4), *q = malloc(n * 4), *gx = malloc(n * 4);
    float *wq = malloc(n * 4), *gw = malloc(n * 4), *zp = malloc(d0 * 4), *s = malloc(d0 * 4), *ls = malloc(d0 * 4),
          *lwq = malloc(d0 * 4);
    int8_t* r = malloc(n);
    double grads[3], *gls = malloc(d0 * 8);
    for (int64_t i = 0; i < n; ++i) { x[i] = rnd() * 6.f - 3.f; g[i] = rnd() * 2.f - 1.f; r[i] = rnd() > 0.5f ? 1 : -1; }
    for (int64_t c = 0; c < d0; ++c) { ls[c] = -5.f + rnd(); s[c] = exp2f(ls[c]); }
    for (int method = 0; method < 4; ++method) {
      for (int inverted = 0; inverted < 2; ++inverted) {
        const float qr = inverted ? 0.01f : 4.f;      /* qr < s: lo > hi */
        if (mhaq_ref_act(x, g, r, n, d0, 0.25f, qr, -1.5f, method, y, q, gx, grads)) bad++;
        for (int64_t i = 0; i < n; ++i) bad += !(isfinite(y[i]) && isfinite(gx[i]));
        bad += !(isfinite(grads[0]) && isfinite(grads[1]) && isfinite(grads[2]));
      }
      for (int pc = 0; pc < 2; ++pc) {
        if (mhaq_ref_weight(x, g, r, d0, row, s, pc, method, wq, zp, gw, gls)) bad++;
        for (int64_t i = 0; i < n; ++i) bad += !(isfinite(wq[i]) && isfinite(gw[i]));
        for (int64_t c = 0; c < (pc ? d0 : 1); ++c) bad += !isfinite(gls[c]);
      }
    }
    x[0] = NAN;                                        /* NaN / inf propagate without tripping the sanitizers */
    if (n > 1) x[n - 1] = INFINITY;
    mhaq_ref_act(x, g, r, n, d0, 0.25f, 4.f, -1.5f, 2, y, q, gx, grads);
    mhaq_ref_weight(x, g, r, d0, row, s, 1, 2, wq, zp, gw, gls);
    x[0] = 0.f; if (n > 1) x[n - 1] = 1.f;
    mhaq_ref_regulariser_input(x, d0, row, ls, 1, lwq);
    double rl;
    const double pl = mhaq_ref_potential_loss(0.7, ls, lwq, d0, ls, lwq, d0, 4, 4, 1, 0, 0.5, 3.0, 4.0, &rl);
    bad += !(isfinite(pl) && isfinite(rl));
    free(x); free(g); free(y); free(q); free(gx); free(wq); free(gw); free(zp); free(s); free(ls); free(lwq); free(r); free(gls);
  }
  printf("fq_ref selftest: %d problems\n", bad);
  return bad ? 1 : 0;
}
