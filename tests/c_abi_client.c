/* Plain-C consumer of include/gprc_native.h -- what the R `.Call` shim does, minus R: host arrays in, host arrays
 * out, status codes, opaque handles.  Runs the reference's closed-form case (tests/testthat/test-gpr.R:23-27), the
 * jitter loop on a robustly indefinite input, a GPC fit + class probabilities, and the error paths.
 * Built and run by tests/test_gpu_c_abi.py (gcc, links libgprc_native.so only).  Exit code 0 = all checks passed. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gprc_native.h"

static int fails = 0;
#define CHECK(cond, ...)                                  \
  do {                                                    \
    if (!(cond)) { printf("FAIL %s:%d ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); ++fails; } \
  } while (0)

int main(void) {
  gprc_ctx* ctx = NULL;
  int ndev = 0;
  CHECK(gprc_abi_version() == GPRC_ABI_VERSION, "abi version");
  CHECK(gprc_device_count(&ndev) == 0 && ndev >= 1, "device count %d", ndev);
  int rc = gprc_ctx_create(0, NULL, &ctx);
  CHECK(rc == 0, "ctx_create rc=%d (%s)", rc, gprc_last_error());
  if (rc != 0) return 2;

  /* GPR.sqrexp$new(X = (1,2), y = (0,1), noise = 1, l = 1)$predict(0) */
  const double X[2] = {1.0, 2.0}, y[2] = {0.0, 1.0}, l = 1.0, xs = 0.0;
  gprc_model* m = NULL;
  double noise_used = -1.0, mean = 0.0, var = 0.0, alpha[2], logp = 0.0, L[4];
  int attempts = 0;
  rc = gprc_gpr_fit_retry(ctx, GPRC_SQREXP, &l, 1, X, 1, 2, y, 1.0, &m, &noise_used, &attempts);
  CHECK(rc == 0 && attempts == 1 && noise_used == 1.0, "fit_retry rc=%d attempts=%d noise=%g", rc, attempts, noise_used);
  CHECK(gprc_gpr_predict(m, &xs, 1, 1, &mean, &var) == 0, "predict");
  const double em = (2 * exp(-2.0) - exp(-1.0)) / (4 - exp(-1.0));
  const double ev = 1 - (2 * exp(-1.0) - 2 * exp(-3.0) + 2 * exp(-4.0)) / (4 - exp(-1.0));
  CHECK(fabs(mean - em) < 1e-15 && fabs(var - ev) < 1e-15, "known answer: mean %.17g var %.17g", mean, var);
  CHECK(gprc_gpr_get_alpha(m, alpha) == 0 && gprc_gpr_get_logp(m, &logp) == 0 && gprc_model_get_L(m, L, 2) == 0, "getters");
  CHECK(fabs(L[0] - sqrt(2.0)) < 1e-15 && L[2] == 0.0 && fabs(L[1] - exp(-0.5) / sqrt(2.0)) < 1e-15, "L = t(chol(K + I))");
  int64_t n = 0, d = 0;
  CHECK(gprc_model_dims(m, &n, &d) == 0 && n == 2 && d == 1, "dims");
  gprc_model_free(m);

  /* one attempt on an indefinite matrix -> LAPACK info; the jitter loop recovers at the 4th attempt */
  const double Xn[2] = {0.15, 0.05}, yn[2] = {1.0, -1.0}, sigma = -1.0;
  m = NULL;
  rc = gprc_gpr_fit(ctx, GPRC_LINEAR, &sigma, 1, Xn, 1, 2, yn, 0.0, &m);
  CHECK(rc == 1 && m == NULL && strstr(gprc_last_error(), "leading minor of order 1") != NULL, "info rc=%d msg=%s", rc, gprc_last_error());
  rc = gprc_gpr_fit_retry(ctx, GPRC_LINEAR, &sigma, 1, Xn, 1, 2, yn, 0.0, &m, &noise_used, &attempts);
  CHECK(rc == 0 && attempts == 4 && noise_used == 0.03, "jitter rc=%d attempts=%d noise=%.17g", rc, attempts, noise_used);
  gprc_model_free(m);
  const double Xp[3] = {2.0, 0.1, 0.5}, yp[3] = {1, 2, 3}, pp[2] = {-1.0, 1.0};
  m = NULL;
  rc = gprc_gpr_fit_retry(ctx, GPRC_POLYNOMIAL, pp, 2, Xp, 1, 3, yp, 0.0, &m, &noise_used, &attempts);
  CHECK(rc == GPRC_ERR_NOT_PD && m == NULL, "all ten attempts fail rc=%d", rc);

  /* argument errors never abort */
  CHECK(gprc_gpr_fit(ctx, 99, &l, 1, X, 1, 2, y, 1.0, &m) == GPRC_ERR_ARG, "unknown kernel id");
  CHECK(gprc_gpr_fit(ctx, GPRC_SQREXP, &l, 2, X, 1, 2, y, 1.0, &m) == GPRC_ERR_ARG, "wrong parameter count");
  CHECK(gprc_gpr_fit(ctx, GPRC_SQREXP, &l, 1, X, 1, 2, y, -1.0, &m) == GPRC_ERR_ARG, "negative noise");

  /* GPC: 1-D step problem of tests/testthat/test-gpc.R:5-10 (kappa = exp(-3 (x-y)^2) -> sqrexp, l = sqrt(1/6)) */
  double Xc[21], yc[21], probe[2] = {-0.2, 0.2}, prob[2], lc = sqrt(1.0 / 6.0);
  for (int i = 0; i < 21; ++i) { Xc[i] = -1.0 + 0.1 * i; yc[i] = Xc[i] > 1e-12 ? 1.0 : -1.0; }
  int iters = 0;
  m = NULL;
  rc = gprc_gpc_fit(ctx, GPRC_SQREXP, &lc, 1, Xc, 1, 21, yc, 1e-5, 0, GPRC_GPC_REFERENCE_STOP, &m, &iters);
  CHECK(rc == 0 && iters >= 2 && iters < 30, "gpc_fit rc=%d iters=%d", rc, iters);
  CHECK(gprc_gpc_predict_class(m, probe, 2, prob) == 0 && prob[0] < 0.5 && prob[1] > 0.5, "gpc probs %g %g", prob[0], prob[1]);
  gprc_model_free(m);

  /* covariance_matrix with an odd leading dimension */
  const double A[6] = {0, 0, 1, 0, 0, 1};  /* three 2-d points */
  double K[3 * 5];
  for (int i = 0; i < 15; ++i) K[i] = -7.0;
  CHECK(gprc_kernel_matrix(ctx, GPRC_SQREXP, &l, 1, A, 2, 3, A, 3, K, 5) == 0, "kernel_matrix");
  CHECK(K[0] == 1.0 && fabs(K[1] - exp(-0.5)) < 1e-16 && fabs(K[5 + 2] - exp(-1.0)) < 1e-16 && K[3] == -7.0, "kernel_matrix values / ld");


  /* fit(): objective and gradient (R/fit.R:117-139) on three well separated points */
  const double Xf[3] = {0.0, 1.5, 3.5}, yf[3] = {0.3, -0.2, 0.9}, lf = 0.4;
  double lml = 0.0, grad[2] = {0.0, 0.0};
  CHECK(gprc_gpr_log_marginal(ctx, GPRC_SQREXP, &lf, 1, Xf, 1, 3, yf, 0.1, &lml) == 0 && lml < 0.0 && lml > -10.0, "log_marginal %g", lml);
  CHECK(gprc_fit_gradient(ctx, GPRC_SQREXP, &lf, 1, Xf, 1, 3, yf, grad) == 0 && isfinite(grad[0]), "fit_gradient %g", grad[0]);
  const double same[3] = {1.0, 1.0, 2.0};
  rc = gprc_fit_gradient(ctx, GPRC_SQREXP, &lf, 1, same, 1, 3, yf, grad);
  CHECK(rc > 0, "fit_gradient on a singular K returns info, rc=%d", rc);

  /* multivariate_normal (R/GPRclass.R:360-370): Cholesky branch on diag(4, 9), eigen branch on a rank-1 matrix */
  const double mu[2] = {1.0, -1.0}, cov1[4] = {4.0, 0.0, 0.0, 9.0}, cov2[4] = {1.0, 1.0, 1.0, 1.0}, Z[4] = {1.0, 0.5, -1.0, 2.0};
  double draws[4], vals[2], vecs[4];
  int method = 0, sweeps = 0;
  CHECK(gprc_mvn_sample(ctx, cov1, 2, 2, mu, 1e-6, Z, 2, draws, &method) == 0 && method == 1, "mvn chol method=%d", method);
  CHECK(fabs(draws[0] - 3.0) < 1e-14 && fabs(draws[1] - 0.5) < 1e-14 && fabs(draws[2] + 1.0) < 1e-14 && fabs(draws[3] - 5.0) < 1e-14, "mvn chol draws");
  CHECK(gprc_mvn_sample(ctx, cov2, 2, 2, mu, 1e-6, Z, 2, draws, &method) == 0 && method == 2, "mvn eigen method=%d", method);
  CHECK(fabs((draws[0] - mu[0]) - (draws[1] - mu[1])) < 1e-14, "rank-1 covariance: both coordinates move together");
  CHECK(gprc_sym_eigen(ctx, cov2, 2, 2, vals, vecs, &sweeps) == 0 && fabs(vals[0] - 2.0) < 1e-14 && fabs(vals[1]) < 1e-14, "sym_eigen %g %g", vals[0], vals[1]);
  const double neg[4] = {1.0, 0.0, 0.0, -1.0};
  CHECK(gprc_mvn_sample(ctx, neg, 2, 2, mu, 1e-6, Z, 2, draws, &method) == GPRC_ERR_NOT_PD, "indefinite covariance is refused");

  /* combine_all (R/simulation.R:338-349): last axis fastest */
  const double axes[5] = {0.0, 1.0, 10.0, 20.0, 30.0};
  const int64_t lens[2] = {2, 3};
  double grid[12];
  CHECK(gprc_combine_all(ctx, axes, lens, 2, grid) == 0, "combine_all");
  CHECK(grid[0] == 0.0 && grid[1] == 10.0 && grid[2] == 0.0 && grid[3] == 20.0 && grid[6] == 1.0 && grid[7] == 10.0 && grid[11] == 30.0, "combine_all order");
  CHECK(gprc_ctx_trim(ctx) == 0, "ctx_trim");

  /* ---- multi-GPU from ONE process (gprc_mgpu_*): G = 1, 2, 3 ranks -- VIRTUAL ranks, all on device 0, exchanging
   * panels by device copies -- must reproduce gprc_gpr_fit / gprc_gpr_predict bit for bit.  n = 2900: 6 panels, so every
   * rank owns several, look-ahead and the batched far updates are exercised; then no look-ahead; then the RCCL exchange
   * with the one rank a one-GPU box allows (ncclCommInitAll + grouped ncclBroadcast). */
  {
    const int64_t nn = 2900, dd = 3, nst = 777;
    double* Xm = (double*)malloc(sizeof(double) * dd * nn);
    double* ym = (double*)malloc(sizeof(double) * nn);
    double* Xsm = (double*)malloc(sizeof(double) * dd * nst);
    double *a_ref = (double*)malloc(sizeof(double) * nn), *a_got = (double*)malloc(sizeof(double) * nn);
    double *mean_ref = (double*)malloc(sizeof(double) * nst), *var_ref = (double*)malloc(sizeof(double) * nst);
    double *mean_got = (double*)malloc(sizeof(double) * nst), *var_got = (double*)malloc(sizeof(double) * nst);
    unsigned long long st = 88172645463325252ULL;   /* xorshift64: the client's own generator */
    #define NEXT_U01() (st ^= st << 13, st ^= st >> 7, st ^= st << 17, (double)(st >> 11) / 9007199254740992.0)
    for (int64_t i = 0; i < nn; ++i) {
      double acc = 0.0;
      for (int64_t r = 0; r < dd; ++r) { const double v = 2.0 * NEXT_U01() - 1.0; Xm[i * dd + r] = v; acc += v * v * v; }
      ym[i] = 0.1 * acc + 0.05 * (NEXT_U01() - 0.5);
    }
    for (int64_t i = 0; i < dd * nst; ++i) Xsm[i] = 2.0 * NEXT_U01() - 1.0;
    const double lm = 0.6;
    double logp_ref = 0.0, logp_got = 0.0;
    gprc_model* ref = NULL;
    rc = gprc_gpr_fit(ctx, GPRC_SQREXP, &lm, 1, Xm, dd, nn, ym, 0.1, &ref);
    CHECK(rc == 0, "reference fit rc=%d (%s)", rc, gprc_last_error());
    CHECK(gprc_gpr_get_alpha(ref, a_ref) == 0 && gprc_gpr_get_logp(ref, &logp_ref) == 0, "reference getters");
    CHECK(gprc_gpr_predict(ref, Xsm, nst, 1, mean_ref, var_ref) == 0, "reference predict");
    const int devs[3] = {0, 0, 0};
    for (int variant = 0; variant < 5; ++variant) {   /* G = 1, 2, 3 with look-ahead; G = 2 without; RCCL with one rank */
      const int G = variant < 3 ? variant + 1 : (variant == 3 ? 2 : 1);
      const int flags = variant == 3 ? GPRC_MGPU_NO_LOOKAHEAD : (variant == 4 ? GPRC_MGPU_RCCL : 0);
      gprc_mgpu* mg = NULL;
      gprc_mgpu_model* mm = NULL;
      rc = gprc_mgpu_create(devs, G, flags, &mg);
      CHECK(rc == 0, "mgpu_create variant %d rc=%d (%s)", variant, rc, gprc_last_error());
      if (rc != 0) continue;
      int gr = 0;
      CHECK(gprc_mgpu_ranks(mg, &gr) == 0 && gr == G, "mgpu_ranks");
      for (int rep = 0; rep < 2 && rc == 0; ++rep) {   /* twice: streams, events and flags are reused across fits */
        rc = gprc_mgpu_gpr_fit_retry(mg, GPRC_SQREXP, &lm, 1, Xm, dd, nn, ym, 0.1, &mm, &noise_used, &attempts);
        CHECK(rc == 0 && attempts == 1 && noise_used == 0.1, "mgpu fit variant %d rc=%d (%s)", variant, rc, gprc_last_error());
        if (rc != 0) break;
        CHECK(gprc_mgpu_gpr_get_alpha(mm, a_got) == 0 && gprc_mgpu_gpr_get_logp(mm, &logp_got) == 0, "mgpu getters");
        CHECK(memcmp(a_got, a_ref, sizeof(double) * nn) == 0, "variant %d: alpha differs from gprc_gpr_fit", variant);
        CHECK(logp_got == logp_ref, "variant %d: logp %.17g vs %.17g", variant, logp_got, logp_ref);
        CHECK(gprc_mgpu_gpr_predict(mm, Xsm, nst, mean_got, var_got) == 0, "mgpu predict (%s)", gprc_last_error());
        CHECK(memcmp(mean_got, mean_ref, sizeof(double) * nst) == 0 && memcmp(var_got, var_ref, sizeof(double) * nst) == 0,
              "variant %d: sliced predict differs from gprc_gpr_predict", variant);
        gprc_model* r_last = NULL;                     /* every rank holds the whole factor: the LAST rank's replica predicts alone */
        CHECK(gprc_mgpu_model_rank(mm, G - 1, &r_last) == 0 && r_last != NULL, "model_rank");
        CHECK(gprc_gpr_predict(r_last, Xsm, 64, 1, mean_got, var_got) == 0 && memcmp(mean_got, mean_ref, sizeof(double) * 64) == 0 &&
              memcmp(var_got, var_ref, sizeof(double) * 64) == 0, "variant %d: rank %d's replica of L", variant, G - 1);
        gprc_mgpu_model_free(mm);
        mm = NULL;
      }
      /* LAPACK info and the jitter loop travel through the multi-rank sweep too */
      rc = gprc_mgpu_gpr_fit(mg, GPRC_LINEAR, &sigma, 1, Xn, 1, 2, yn, 0.0, &mm);
      CHECK(rc == 1 && mm == NULL, "mgpu info rc=%d", rc);
      rc = gprc_mgpu_gpr_fit_retry(mg, GPRC_LINEAR, &sigma, 1, Xn, 1, 2, yn, 0.0, &mm, &noise_used, &attempts);
      CHECK(rc == 0 && attempts == 4 && noise_used == 0.03, "mgpu jitter rc=%d attempts=%d", rc, attempts);
      gprc_mgpu_model_free(mm);
      gprc_mgpu_destroy(mg);
    }
    /* the same at 18 panels with three ranks on the one GPU: nine look-ahead chains (one fused, flag-synchronised launch
     * each) run beside three ranks' trailing updates -- the hand-offs under uneven load; every word of alpha compared */
    {
      const int64_t n2 = 9100;
      double* X2 = (double*)malloc(sizeof(double) * dd * n2);
      double* y2 = (double*)malloc(sizeof(double) * n2);
      double *a2r = (double*)malloc(sizeof(double) * n2), *a2g = (double*)malloc(sizeof(double) * n2);
      for (int64_t i = 0; i < n2; ++i) {
        double acc = 0.0;
        for (int64_t r = 0; r < dd; ++r) { const double v = 2.0 * NEXT_U01() - 1.0; X2[i * dd + r] = v; acc += v * v * v; }
        y2[i] = 0.1 * acc + 0.05 * (NEXT_U01() - 0.5);
      }
      gprc_model* r2 = NULL;
      gprc_mgpu* mg3 = NULL;
      gprc_mgpu_model* m3 = NULL;
      double lp_r = 0.0, lp_g = 0.0;
      CHECK(gprc_gpr_fit(ctx, GPRC_SQREXP, &lm, 1, X2, dd, n2, y2, 0.1, &r2) == 0, "reference fit n=9100 (%s)", gprc_last_error());
      CHECK(gprc_gpr_get_alpha(r2, a2r) == 0 && gprc_gpr_get_logp(r2, &lp_r) == 0, "reference getters n=9100");
      CHECK(gprc_mgpu_create(devs, 3, 0, &mg3) == 0, "mgpu_create G=3 (%s)", gprc_last_error());
      for (int rep = 0; rep < 3 && mg3; ++rep) {
        rc = gprc_mgpu_gpr_fit(mg3, GPRC_SQREXP, &lm, 1, X2, dd, n2, y2, 0.1, &m3);
        CHECK(rc == 0, "mgpu fit n=9100 rc=%d (%s)", rc, gprc_last_error());
        if (rc != 0) break;
        CHECK(gprc_mgpu_gpr_get_alpha(m3, a2g) == 0 && gprc_mgpu_gpr_get_logp(m3, &lp_g) == 0, "mgpu getters n=9100");
        CHECK(memcmp(a2g, a2r, sizeof(double) * n2) == 0 && lp_g == lp_r, "n=9100, G=3, rep %d: differs from gprc_gpr_fit", rep);
        gprc_mgpu_model_free(m3);
        m3 = NULL;
      }
      gprc_mgpu_destroy(mg3);
      gprc_model_free(r2);
      free(X2); free(y2); free(a2r); free(a2g);
    }
    const int twice[2] = {0, 0};
    gprc_mgpu* bad = NULL;
    CHECK(gprc_mgpu_create(twice, 2, GPRC_MGPU_RCCL, &bad) == GPRC_ERR_ARG && bad == NULL, "RCCL refuses virtual ranks");
    gprc_model_free(ref);
    free(Xm); free(ym); free(Xsm); free(a_ref); free(a_got); free(mean_ref); free(var_ref); free(mean_got); free(var_got);
  }

  gprc_ctx_destroy(ctx);
  printf(fails ? "c_abi_client: %d FAILED\n" : "c_abi_client: all checks passed\n", fails);
  return fails ? 1 : 0;
}
