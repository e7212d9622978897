/* The CPU oracle (oracle/gprc_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer: this driver #includes the
 * oracle's translation unit, runs the reference's closed-form case (tests/testthat/test-gpr.R:23-27), a blocked fit + predict
 * with ragged sizes (n = 531, ns = 77: blocks of the blocked tier do not divide them), the jitter loop on an indefinite input, a
 * GPC fit + latent predict and the eigen / sampling helpers, and exits non-zero on a wrong answer; the sanitizers abort on any
 * out-of-bounds access, leak, or undefined operation.  GPU code cannot be sanitised on this pool (no GPU ASan): this covers the
 * checker the GPU is compared against.  Built and run by tests/test_oracle_cpu.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../oracle/gprc_oracle.c"

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { printf("FAIL line %d: ", __LINE__); printf(__VA_ARGS__); printf("\n"); ++fails; } } while (0)

int main(void) {
  /* closed form */
  const double X[2] = {1.0, 2.0}, y[2] = {0.0, 1.0}, l = 1.0, xs = 0.0;
  double L[4], alpha[2], logp, nu, mean, var;
  int att, info1;
  CHECK(oracle_gpr_fit(3, &l, 1, X, 1, 2, y, 1.0, L, alpha, &logp, &nu, &att, &info1) == 0 && att == 1, "closed-form fit");
  CHECK(oracle_gpr_predict(3, &l, 1, X, 1, 2, L, alpha, &xs, 1, 1, &mean, &var) == 0, "closed-form predict");
  CHECK(fabs(mean - (2 * exp(-2.0) - exp(-1.0)) / (4 - exp(-1.0))) < 1e-15, "mean %.17g", mean);
  CHECK(fabs(var - (1 - (2 * exp(-1.0) - 2 * exp(-3.0) + 2 * exp(-4.0)) / (4 - exp(-1.0)))) < 1e-15, "var %.17g", var);
  /* ragged blocked fit + predict, compared with the unblocked tier */
  const int64_t n = 531, d = 3, ns = 77;
  double* Xr = malloc(sizeof(double) * d * n), *yr = malloc(sizeof(double) * n), *Xq = malloc(sizeof(double) * d * ns);
  double* Lb = malloc(sizeof(double) * n * n), *work = malloc(sizeof(double) * ns * n), *ab = malloc(sizeof(double) * n);
  double* Lu = malloc(sizeof(double) * n * n), *au = malloc(sizeof(double) * n);
  double *mb = malloc(sizeof(double) * ns), *vb = malloc(sizeof(double) * ns), *mu = malloc(sizeof(double) * ns), *vu = malloc(sizeof(double) * ns);
  unsigned long long st = 88172645463325252ULL;
#define U01() (st ^= st << 13, st ^= st >> 7, st ^= st << 17, (double)(st >> 11) / 9007199254740992.0)
  for (int64_t i = 0; i < d * n; ++i) Xr[i] = 2 * U01() - 1;
  for (int64_t i = 0; i < n; ++i) yr[i] = U01() - 0.5;
  for (int64_t i = 0; i < d * ns; ++i) Xq[i] = 2 * U01() - 1;
  const double rq[2] = {0.8, 1.5};
  double lpb, lpu;
  CHECK(oracle_gpr_fit_predict_blocked(5, rq, 2, Xr, d, n, yr, 0.1, Xq, ns, Lb, work, ab, &lpb, mb, vb) == 0, "blocked tier");
  CHECK(oracle_gpr_fit(5, rq, 2, Xr, d, n, yr, 0.1, Lu, au, &lpu, &nu, &att, &info1) == 0 && att == 1, "unblocked fit");
  CHECK(oracle_gpr_predict(5, rq, 2, Xr, d, n, Lu, au, Xq, ns, 1, mu, vu) == 0, "unblocked predict");
  double e = 0;
  for (int64_t i = 0; i < ns; ++i) { e = fmax(e, fabs(mb[i] - mu[i])); e = fmax(e, fabs(vb[i] - vu[i])); }
  CHECK(e < 1e-11 && fabs(lpb - lpu) < 1e-9 * fabs(lpu), "blocked vs unblocked %g", e);
  /* full covariance */
  double* cov = malloc(sizeof(double) * ns * ns);
  CHECK(oracle_gpr_predict(5, rq, 2, Xr, d, n, Lu, au, Xq, ns, 0, mu, cov) == 0 && fabs(cov[0] - vu[0]) < 1e-12, "full covariance");
  /* jitter loop on an indefinite matrix: fourth attempt */
  const double Xn[2] = {0.15, 0.05}, yn[2] = {1.0, -1.0}, sigma = -1.0;
  CHECK(oracle_gpr_fit(1, &sigma, 1, Xn, 1, 2, yn, 0.0, L, alpha, &logp, &nu, &att, &info1) == 0 && att == 4 && info1 == 1, "jitter att=%d info=%d", att, info1);
  /* GPC step problem + latent predict */
  double Xc[21], yc[21], fh[21], Lc[21 * 21], logq, probe[2] = {-0.2, 0.2}, fs[2], vf[2], lc = sqrt(1.0 / 6.0);
  int iters = 0;
  for (int i = 0; i < 21; ++i) { Xc[i] = -1.0 + 0.1 * i; yc[i] = Xc[i] > 1e-12 ? 1.0 : -1.0; }
  CHECK(oracle_gpc_fit(3, &lc, 1, Xc, 1, 21, yc, 1e-5, 1000, 1, fh, Lc, &logq, &iters) == 0 && iters >= 2, "gpc fit iters=%d", iters);
  CHECK(oracle_gpc_predict_latent(3, &lc, 1, Xc, 1, 21, yc, fh, Lc, probe, 2, fs, vf) == 0 && fs[0] < 0 && fs[1] > 0 && vf[0] > 0, "gpc latent");
  /* eigen + sampling helpers on a rank-deficient covariance */
  const double c2[4] = {1.0, 1.0, 1.0, 1.0};
  double vals[2], vecs[4], Lf[4];
  CHECK(oracle_sym_eigen(c2, 2, 2, vals, vecs) >= 0 && fabs(vals[0] - 2.0) < 1e-14, "eigen %g", vals[0]);
  CHECK(oracle_mvn_factor(c2, 2, 2, 1e-6, Lf) >= 0, "mvn_factor");
  free(Xr); free(yr); free(Xq); free(Lb); free(work); free(ab); free(Lu); free(au); free(mb); free(vb); free(mu); free(vu); free(cov);
  printf(fails ? "oracle_sanitize_driver: %d FAILED\n" : "oracle_sanitize_driver: ok\n", fails);
  return fails ? 1 : 0;
}
