/*
 * gprc_oracle.c -- CPU restatement of the GP predict hot path of the R package `gprc`
 * (MoHawastaken/Gaussian-Process-Regression).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gaussian-process-regression_amd/ may include, link,
 * import or execute this file.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
 * leg use it -- as the checker / the timed CPU baseline, never as the product.
 *
 * Parity status: PINNED for GPR by the reference's four closed-form known answers
 * (tests/testthat/test-gpr.R:6-27, restated in tests/test_oracle_cpu.py) and by an independent
 * numpy/scipy(LAPACK) restatement (tests/golden/make_golden.py).  The reference itself (R) cannot be
 * executed in the build container (no R toolchain) and has no C sources, so there is no oracle/_ref.
 * gammaexp / rationalquadratic / linear / polynomial p!=1 / full covariance / logp / all GPC numerics
 * are unpinned BY THE REFERENCE (its GPC tests are sign-only); they are pinned here by the
 * independent restatement only.
 *
 * The arithmetic of the reference lives in base R (third-party, version unpinned by the package:
 * DESCRIPTION:1-32 has no Depends): outer/colSums/chol(dpotrf)/solve(dgesv)/%*%.  This file restates
 * the published semantics of those primitives:
 *   - colSums accumulates in long double (base R src/main/array.c do_colsum, LDOUBLE);
 *   - x^2 is x*x, other powers go through libm pow (base R arithmetic.c R_POW / R_pow);
 *   - chol() is LAPACK dpotrf('U') of the upper triangle, transposed by the caller (R/GPRclass.R:142);
 *   - solve(L, b) is a general dgesv; restated as the mathematically identical triangular solve.
 *
 * Layout convention (R/GPRclass.R:29,132,137): X is d x n, one observation per COLUMN, column-major,
 * i.e. point i is the contiguous d doubles at X + i*d.  All matrices are column-major fp64.
 *
 * Two tiers:
 *   oracle_*            textbook, unblocked, single thread -- the restatement proper
 *   oracle_*_blocked    cache-blocked + OpenMP, same algorithm -- the timed CPU "port" baseline
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* kernel ids -- must equal include/gprc_native.h gprc_kernel_id */
enum { K_CONSTANT = 0, K_LINEAR = 1, K_POLYNOMIAL = 2, K_SQREXP = 3, K_GAMMAEXP = 4, K_RATQUAD = 5 };

typedef long double ldbl;

ORACLE_API int oracle_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
ORACLE_API void oracle_set_threads(int t) {
#ifdef _OPENMP
  if (t > 0) omp_set_num_threads(t);
#else
  (void)t;
#endif
}

/* base R `^` on doubles: arithmetic.c R_POW -> (y == 2) ? x*x : R_pow(x, y) */
static double r_pow(double x, double y) {
  if (y == 2.0) return x * x;
  if (x == 1.0 || y == 0.0) return 1.0;
  if (x == 0.0) {
    if (y > 0.0) return 0.0;
    if (y < 0.0) return INFINITY;
    return y;
  }
  return pow(x, y);
}

/* number of parameters each kernel takes; linear takes 1 (recycled) or d */
static int check_params(int id, int npar, int64_t d) {
  switch (id) {
    case K_CONSTANT: return npar == 1;
    case K_LINEAR: return npar == 1 || npar == d;
    case K_POLYNOMIAL: return npar == 2;
    case K_SQREXP: return npar == 1;
    case K_GAMMAEXP: return npar == 2;
    case K_RATQUAD: return npar == 2;
    default: return 0;
  }
}

/* One kernel value k(x, y) for two d-vectors.
 * Formulas: R/GPRclass.R:382 constant, :386 linear, :390 polynomial, :394 sqrexp, :398 gammaexp,
 * :402 rationalquadratic (the .matrix methods; colSums -> long double accumulation). */
static double kernel_pair(int id, const double* par, int npar, const double* x, const double* y, int64_t d) {
  ldbl s = 0.0L;
  switch (id) {
    case K_CONSTANT: /* rep(c, ncol(x)) */
      return par[0];
    case K_LINEAR: /* colSums(sigma * x * y): (sigma*x)*y, sigma recycled down the rows */
      for (int64_t r = 0; r < d; ++r) {
        double sg = (npar == 1) ? par[0] : par[r];
        double t = sg * x[r];
        t = t * y[r];
        s += (ldbl)t;
      }
      return (double)s;
    case K_POLYNOMIAL: /* (colSums(x * y) + sigma)^p ; par = (sigma, p) */
      for (int64_t r = 0; r < d; ++r) s += (ldbl)(x[r] * y[r]);
      return r_pow((double)s + par[0], par[1]);
    default: break;
  }
  /* stationary kernels share colSums((x - y)^2) */
  for (int64_t r = 0; r < d; ++r) {
    double df = x[r] - y[r];
    s += (ldbl)(df * df);
  }
  double ss = (double)s;
  switch (id) {
    case K_SQREXP: { /* exp(-colSums((x-y)^2) / (2*l^2)) ; par = (l) */
      double l = par[0];
      return exp(-ss / (2.0 * (l * l)));
    }
    case K_GAMMAEXP: { /* exp(-(sqrt(colSums((x-y)^2))/l)^gamma) ; par = (l, gamma) */
      double l = par[0], g = par[1];
      return exp(-r_pow(sqrt(ss) / l, g));
    }
    case K_RATQUAD: { /* (1 + colSums((x-y)^2)/(2*alpha*l^2))^(-alpha) ; par = (l, alpha) */
      double l = par[0], a = par[1];
      return r_pow(1.0 + ss / (2.0 * a * (l * l)), -a);
    }
    default: return NAN;
  }
}

/* a2: column-wise kernel on two d x m matrices (the cov_func closure contract, R/GPRclass.R:353-354) */
ORACLE_API int oracle_kernel_colwise(int id, const double* par, int npar, const double* x, const double* y,
                                     int64_t d, int64_t m, double* out) {
  if (!check_params(id, npar, d)) return -1;
  for (int64_t c = 0; c < m; ++c) out[c] = kernel_pair(id, par, npar, x + c * d, y + c * d, d);
  return 0;
}

/* a3: covariance_matrix(A, B, k) -> nA x nB, [i,j] = k(A[,i], B[,j])  (R/GPRclass.R:355-357) */
ORACLE_API int oracle_kernel_matrix(int id, const double* par, int npar, const double* A, int64_t d, int64_t nA,
                                    const double* B, int64_t nB, double* out) {
  if (!check_params(id, npar, d)) return -1;
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < nB; ++j)
    for (int64_t i = 0; i < nA; ++i) out[i + j * nA] = kernel_pair(id, par, npar, A + i * d, B + j * d, d);
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Textbook tier
 * ---------------------------------------------------------------------------------------------- */

/* Lower Cholesky in place (only the lower triangle is read/written; upper left untouched), the
 * unblocked LAPACK dpotf2 recurrence.  Returns LAPACK info: 0, or j+1 = order of the first leading
 * minor that is not positive definite (pivot <= 0 or NaN), as chol() reports it (R/GPRclass.R:142). */
ORACLE_API int oracle_potrf_lower(double* A, int64_t n, int64_t lda) {
  for (int64_t j = 0; j < n; ++j) {
    double ajj = A[j + j * lda];
    for (int64_t k = 0; k < j; ++k) ajj -= A[j + k * lda] * A[j + k * lda];
    if (!(ajj > 0.0)) return (int)(j + 1);
    ajj = sqrt(ajj);
    A[j + j * lda] = ajj;
    for (int64_t i = j + 1; i < n; ++i) {
      double s = A[i + j * lda];
      for (int64_t k = 0; k < j; ++k) s -= A[i + k * lda] * A[j + k * lda];
      A[i + j * lda] = s / ajj;
    }
  }
  return 0;
}

/* b := L^{-1} b */
ORACLE_API void oracle_trsv_lower(const double* L, int64_t n, int64_t ld, double* b) {
  for (int64_t i = 0; i < n; ++i) {
    double s = b[i];
    for (int64_t k = 0; k < i; ++k) s -= L[i + k * ld] * b[k];
    b[i] = s / L[i + i * ld];
  }
}
/* b := L^{-T} b */
ORACLE_API void oracle_trsv_lower_t(const double* L, int64_t n, int64_t ld, double* b) {
  for (int64_t i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int64_t k = i + 1; k < n; ++k) s -= L[k + i * ld] * b[k];
    b[i] = s / L[i + i * ld];
  }
}

static void zero_upper(double* L, int64_t n) {
  for (int64_t j = 1; j < n; ++j)
    for (int64_t i = 0; i < j; ++i) L[i + j * n] = 0.0;
}

/* a4: GPR$initialize (R/GPRclass.R:127-154), one Cholesky attempt per jitter step.
 * Outputs: L (n x n, lower, upper zero, as t(chol(.)) gives), alpha, logp, noise actually used,
 * attempts (1..10).  Returns 0, or 1 when all ten attempts fail (the reference stop()s, :149).
 * info_first receives the LAPACK info of the first attempt (0 when it succeeded). */
ORACLE_API int oracle_gpr_fit(int id, const double* par, int npar, const double* X, int64_t d, int64_t n,
                              const double* y, double noise, double* L, double* alpha, double* logp,
                              double* noise_used, int* attempts, int* info_first) {
  if (!check_params(id, npar, d)) return -1;
  double* K = (double*)malloc(sizeof(double) * n * n);
  if (!K) return -2;
  oracle_kernel_matrix(id, par, npar, X, d, n, X, n, K);
  double new_noise = noise;
  int ok = 0;
  *info_first = 0;
  for (int i = 1; i <= 10; ++i) { /* R/GPRclass.R:141-148 */
    memcpy(L, K, sizeof(double) * n * n);
    for (int64_t j = 0; j < n; ++j) L[j + j * n] += new_noise; /* K + new_noise * diag(n): noise is a variance */
    int info = oracle_potrf_lower(L, n, n);
    if (i == 1) *info_first = info;
    if (info == 0) {
      ok = 1;
      *attempts = i;
      break;
    }
    new_noise = 0.01 * i + noise;
  }
  free(K);
  if (!ok) {
    *attempts = 10;
    return 1;
  }
  zero_upper(L, n);
  *noise_used = new_noise;
  memcpy(alpha, y, sizeof(double) * n); /* alpha = solve(t(L), solve(L, y))  :152 */
  oracle_trsv_lower(L, n, n, alpha);
  oracle_trsv_lower_t(L, n, n, alpha);
  ldbl ya = 0.0L, sl = 0.0L; /* logp = -0.5 y.alpha - sum(log(diag(L))) - n/2 log(2 pi)  :153 */
  for (int64_t i = 0; i < n; ++i) {
    ya += (ldbl)(y[i] * alpha[i]);
    sl += (ldbl)log(L[i + i * n]);
  }
  *logp = -0.5 * (double)ya - (double)sl - (double)n / 2.0 * log(2.0 * M_PI);
  return 0;
}

/* a5: GPR$predict (R/GPRclass.R:155-170).
 * pointwise != 0: mean[ns], var[ns] = k(x*,x*) - colSums(v*v).
 * pointwise == 0: mean[ns], var = ns x ns posterior covariance K(X*,X*) - t(v) %*% v. */
ORACLE_API int oracle_gpr_predict(int id, const double* par, int npar, const double* X, int64_t d, int64_t n,
                                  const double* L, const double* alpha, const double* Xs, int64_t ns, int pointwise,
                                  double* mean, double* var) {
  if (!check_params(id, npar, d)) return -1;
  double* Ks = (double*)malloc(sizeof(double) * n * ns); /* K_star: n x ns  :160 */
  if (!Ks) return -2;
  oracle_kernel_matrix(id, par, npar, X, d, n, Xs, ns, Ks);
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < ns; ++j) {
    double* col = Ks + j * n;
    double m = 0.0; /* t(K_star) %*% alpha  :161 (BLAS dgemv: plain double accumulation) */
    for (int64_t i = 0; i < n; ++i) m += col[i] * alpha[i];
    mean[j] = m;
    oracle_trsv_lower(L, n, n, col); /* v = solve(L, K_star)  :162 */
  }
  if (pointwise) {
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < ns; ++j) { /* k(X*,X*) - colSums(v*v)  :164 */
      const double* col = Ks + j * n;
      ldbl s = 0.0L;
      for (int64_t i = 0; i < n; ++i) s += (ldbl)(col[i] * col[i]);
      double kss = kernel_pair(id, par, npar, Xs + j * d, Xs + j * d, d);
      var[j] = kss - (double)s;
    }
  } else {
    oracle_kernel_matrix(id, par, npar, Xs, d, ns, Xs, ns, var); /* :167 */
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < ns; ++j)
      for (int64_t i = 0; i < ns; ++i) {
        double s = 0.0;
        for (int64_t k = 0; k < n; ++k) s += Ks[k + i * n] * Ks[k + j * n];
        var[i + j * ns] -= s;
      }
  }
  free(Ks);
  return 0;
}

static double sigmoid(double x) { return 1.0 / (1.0 + exp(-x)); } /* R/GPCclass.R:63 */

/* B = I + (sqrt(W) %o% sqrt(W)) * K, then lower Cholesky  (R/GPCclass.R:80,102) */
static int gpc_chol_B(const double* K, const double* W, int64_t n, double* L) {
  for (int64_t j = 0; j < n; ++j) {
    double sj = sqrt(W[j]);
    for (int64_t i = 0; i < n; ++i) {
      double o = sqrt(W[i]) * sj;
      L[i + j * n] = (i == j ? 1.0 : 0.0) + o * K[i + j * n];
    }
  }
  int info = oracle_potrf_lower(L, n, n);
  if (info == 0) zero_upper(L, n);
  return info;
}

/* a8: GPC$initialize (R/GPCclass.R:66-107): Laplace mode by Newton/IRLS, labels y in {-1,+1}.
 * Returns 0 ok; 2 = "Apparently does not converge." (:90-91); 3 = chol failed; 4 = max_iter hit
 * (the reference loops forever; max_iter is a safety net of this restatement).
 * Reference quirks kept: logq = objective - sum(diag(L)) (NOT log), :103; and the stop rule of :90 --
 * `least_objective + 10 < objective`, where least_objective is the objective of iteration 1 and the
 * objective is being MAXIMISED -- fires whenever Newton improves the objective by more than 10, i.e. for
 * any sizeable n.  divergence_stop = 1 reproduces it (the reference's behaviour); 0 switches it off. */
ORACLE_API int oracle_gpc_fit(int id, const double* par, int npar, const double* X, int64_t d, int64_t n,
                              const double* y, double epsilon, int max_iter, int divergence_stop, double* f_hat, double* L,
                              double* logq, int* iters) {
  if (!check_params(id, npar, d)) return -1;
  double* K = (double*)malloc(sizeof(double) * n * n);
  double* w = (double*)malloc(sizeof(double) * n * 6);
  if (!K || !w) return -2;
  double *P = w, *W = w + n, *b = w + 2 * n, *t = w + 3 * n, *a = w + 4 * n, *f = w + 5 * n;
  oracle_kernel_matrix(id, par, npar, X, d, n, X, n, K);
  for (int64_t i = 0; i < n; ++i) f[i] = 0.0;
  int it = 0, rc = 0;
  double objective = 0.0, last_objective = 0.0, least_objective = 0.0;
  for (;;) {
    ++it;
    for (int64_t i = 0; i < n; ++i) {
      P[i] = sigmoid(f[i]);
      W[i] = (1.0 - P[i]) * P[i];
    }
    if (gpc_chol_B(K, W, n, L) != 0) { rc = 3; break; }
    for (int64_t i = 0; i < n; ++i) b[i] = W[i] * f[i] + (y[i] + 1.0) / 2.0 - P[i]; /* :81 */
    for (int64_t i = 0; i < n; ++i) { /* sqrt(W) * (K %*% b)  :82 */
      double s = 0.0;
      for (int64_t k = 0; k < n; ++k) s += K[i + k * n] * b[k];
      t[i] = sqrt(W[i]) * s;
    }
    oracle_trsv_lower(L, n, n, t);   /* :82 */
    oracle_trsv_lower_t(L, n, n, t); /* :83 */
    for (int64_t i = 0; i < n; ++i) a[i] = b[i] - sqrt(W[i]) * t[i]; /* :84 */
    for (int64_t i = 0; i < n; ++i) { /* f = K %*% a  :85 */
      double s = 0.0;
      for (int64_t k = 0; k < n; ++k) s += K[i + k * n] * a[k];
      f[i] = s;
    }
    double saf = 0.0, sll = 0.0; /* :86 (R sum(): long double) */
    {
      ldbl s1 = 0.0L, s2 = 0.0L;
      for (int64_t i = 0; i < n; ++i) {
        s1 += (ldbl)(a[i] * f[i]);
        s2 += (ldbl)log(1.0 + exp(-y[i] * f[i]));
      }
      saf = (double)s1;
      sll = (double)s2;
    }
    objective = -saf / 2.0 - sll;
    if (it > 1) {
      if (fabs(objective - last_objective) < epsilon) break;          /* :88 */
      else if (divergence_stop && least_objective + 10.0 < objective) { rc = 2; break; } /* :90 */
    } else {
      least_objective = objective;
    }
    last_objective = objective;
    if (it >= max_iter) { rc = 4; break; }
  }
  *iters = it;
  if (rc == 0) {
    for (int64_t i = 0; i < n; ++i) {
      P[i] = sigmoid(f[i]);
      W[i] = (1.0 - P[i]) * P[i];
    }
    if (gpc_chol_B(K, W, n, L) != 0) rc = 3; /* :102 */
    ldbl sd = 0.0L;
    for (int64_t i = 0; i < n; ++i) sd += (ldbl)L[i + i * n];
    *logq = objective - (double)sd; /* :103 sic */
    memcpy(f_hat, f, sizeof(double) * n);
  }
  free(K);
  free(w);
  return rc;
}

/* a9: the hot part of GPC$predict_class (R/GPCclass.R:109-115): fs_bar and Vfs.
 * (The per-point integrate() of :116-117 is SURVEY 8f "next"; tests restate it with scipy QUADPACK.) */
ORACLE_API int oracle_gpc_predict_latent(int id, const double* par, int npar, const double* X, int64_t d, int64_t n,
                                         const double* y, const double* f_hat, const double* L, const double* Xs,
                                         int64_t ns, double* fs_bar, double* Vfs) {
  if (!check_params(id, npar, d)) return -1;
  double* Ks = (double*)malloc(sizeof(double) * n * ns);
  double* w = (double*)malloc(sizeof(double) * n * 2);
  if (!Ks || !w) return -2;
  double *g = w, *sw = w + n;
  for (int64_t i = 0; i < n; ++i) {
    double P = sigmoid(f_hat[i]);
    g[i] = (y[i] + 1.0) / 2.0 - P;
    sw[i] = sqrt(P * (1.0 - P));
  }
  oracle_kernel_matrix(id, par, npar, X, d, n, Xs, ns, Ks); /* :112 */
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < ns; ++j) {
    double* col = Ks + j * n;
    double m = 0.0;
    for (int64_t i = 0; i < n; ++i) m += col[i] * g[i]; /* :113 */
    fs_bar[j] = m;
    for (int64_t i = 0; i < n; ++i) col[i] = sw[i] * col[i]; /* sqrt(W) * K_star (row scaling)  :114 */
    oracle_trsv_lower(L, n, n, col);
    ldbl s = 0.0L;
    for (int64_t i = 0; i < n; ++i) s += (ldbl)(col[i] * col[i]);
    Vfs[j] = kernel_pair(id, par, npar, Xs + j * d, Xs + j * d, d) - (double)s; /* :115 */
  }
  free(Ks);
  free(w);
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Blocked / OpenMP tier: same algorithm, used for the timed CPU baseline and for parity at sizes
 * the textbook tier would take minutes on.  Plain double accumulation throughout.
 * ---------------------------------------------------------------------------------------------- */
#define OB 64 /* block size */

/* C[M x N] -= A[M x K] * B[N x K]^T, all column-major.  lower != 0: C is square-aligned with A rows ==
 * B rows and only the lower triangle (row >= col) has to be right (we still compute whole 4-column
 * strips from the diagonal down). */
static void gemm_sub_nt(int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, const double* B, int64_t ldb,
                        double* C, int64_t ldc, int lower) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t j0 = 0; j0 < N; j0 += 4) {
    int64_t jb = N - j0 < 4 ? N - j0 : 4;
    int64_t istart = lower ? j0 : 0;
    for (int64_t i0 = istart; i0 < M; i0 += 256) {
      int64_t ib = M - i0 < 256 ? M - i0 : 256;
      for (int64_t k = 0; k < K; ++k) {
        const double* a = A + i0 + k * lda;
        for (int64_t jj = 0; jj < jb; ++jj) {
          double bjk = B[(j0 + jj) + k * ldb];
          double* c = C + i0 + (j0 + jj) * ldc;
          for (int64_t i = 0; i < ib; ++i) c[i] -= a[i] * bjk;
        }
      }
    }
  }
}

/* X[M x nb] := X * L11^{-T} for a lower-triangular nb x nb L11 (row-wise forward substitution). */
static void trsm_right_lt(int64_t M, int64_t nb, const double* L11, int64_t ldl, double* X, int64_t ldx) {
#pragma omp parallel for schedule(static)
  for (int64_t i0 = 0; i0 < M; i0 += 64) {
    int64_t ib = M - i0 < 64 ? M - i0 : 64;
    for (int64_t j = 0; j < nb; ++j) {
      double* xj = X + i0 + j * ldx;
      for (int64_t k = 0; k < j; ++k) {
        double ljk = L11[j + k * ldl];
        const double* xk = X + i0 + k * ldx;
        for (int64_t i = 0; i < ib; ++i) xj[i] -= xk[i] * ljk;
      }
      double ljj = L11[j + j * ldl];
      for (int64_t i = 0; i < ib; ++i) xj[i] /= ljj;
    }
  }
}

ORACLE_API int oracle_potrf_lower_blocked(double* A, int64_t n, int64_t lda) {
  for (int64_t k0 = 0; k0 < n; k0 += OB) {
    int64_t kb = n - k0 < OB ? n - k0 : OB;
    int info = oracle_potrf_lower(A + k0 + k0 * lda, kb, lda);
    if (info) return (int)(k0 + info);
    int64_t m = n - k0 - kb;
    if (m > 0) {
      trsm_right_lt(m, kb, A + k0 + k0 * lda, lda, A + (k0 + kb) + k0 * lda, lda);
      gemm_sub_nt(m, m, kb, A + (k0 + kb) + k0 * lda, lda, A + (k0 + kb) + k0 * lda, lda,
                  A + (k0 + kb) + (k0 + kb) * lda, lda, 1);
    }
  }
  return 0;
}

/* Vt[ns x n] := Vt * L^{-T}  (i.e. row j of the result is (L^{-1} K_star[, j])^T), blocked. */
ORACLE_API void oracle_trsm_right_lt_blocked(const double* L, int64_t n, int64_t ldl, double* Vt, int64_t ns, int64_t ldv) {
  for (int64_t k0 = 0; k0 < n; k0 += OB) {
    int64_t kb = n - k0 < OB ? n - k0 : OB;
    trsm_right_lt(ns, kb, L + k0 + k0 * ldl, ldl, Vt + k0 * ldv, ldv);
    int64_t m = n - k0 - kb;
    if (m > 0) gemm_sub_nt(ns, m, kb, Vt + k0 * ldv, ldv, L + (k0 + kb) + k0 * ldl, ldl, Vt + (k0 + kb) * ldv, ldv, 0);
  }
}

/* Single-attempt blocked fit + pointwise predict, the shape bench.py times as the CPU baseline.
 * Returns LAPACK info of the Cholesky (0 ok).  L must hold n*n doubles, work must hold ns*n doubles. */
ORACLE_API int oracle_gpr_fit_predict_blocked(int id, const double* par, int npar, const double* X, int64_t d,
                                              int64_t n, const double* y, double noise, const double* Xs, int64_t ns,
                                              double* L, double* work, double* alpha, double* logp, double* mean,
                                              double* var) {
  if (!check_params(id, npar, d)) return -1;
  oracle_kernel_matrix(id, par, npar, X, d, n, X, n, L);
  for (int64_t j = 0; j < n; ++j) L[j + j * n] += noise;
  int info = oracle_potrf_lower_blocked(L, n, n);
  if (info) return info;
  memcpy(alpha, y, sizeof(double) * n);
  oracle_trsv_lower(L, n, n, alpha);
  double zz = 0.0, sl = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    zz += alpha[i] * alpha[i]; /* y.alpha == |L^{-1}y|^2 up to rounding; recomputed below as y.alpha */
    sl += log(L[i + i * n]);
  }
  oracle_trsv_lower_t(L, n, n, alpha);
  double ya = 0.0;
  for (int64_t i = 0; i < n; ++i) ya += y[i] * alpha[i];
  (void)zz;
  *logp = -0.5 * ya - sl - (double)n / 2.0 * log(2.0 * M_PI);
  /* work = K_star^T (ns x n) */
  oracle_kernel_matrix(id, par, npar, Xs, d, ns, X, n, work);
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < ns; ++j) {
    double m = 0.0;
    for (int64_t i = 0; i < n; ++i) m += work[j + i * ns] * alpha[i];
    mean[j] = m;
  }
  oracle_trsm_right_lt_blocked(L, n, n, work, ns, ns);
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < ns; ++j) {
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += work[j + i * ns] * work[j + i * ns];
    var[j] = kernel_pair(id, par, npar, Xs + j * d, Xs + j * d, d) - s;
  }
  return 0;
}

/* ================================================================================================
 * fit(): gradient of the log marginal likelihood as the reference computes it (R/fit.R:126-139).
 * ============================================================================================== */

/* cov_dict$<kernel>$deriv (R/fit.R:4-31); v is bound positionally in deriv's own argument order:
 * sqrexp (l) :4-7, gammaexp (gamma, l) :10-13, polynomial (sigma, p) :21-23, rationalquadratic (alpha, l) :26-31.
 * Note rationalquadratic's r is the SQUARED distance (:27), the others' r the distance.  out has npar entries. */
static void deriv_pair(int id, const double* v, const double* x, const double* y, int64_t d, double* out) {
  if (id == K_POLYNOMIAL) {
    double xy = 0.0; /* x %*% y */
    for (int64_t k = 0; k < d; ++k) xy += x[k] * y[k];
    double t = xy + v[0];
    out[0] = v[1] * r_pow(t, v[1] - 1.0);
    out[1] = r_pow(t, v[1]) * log(t);
    return;
  }
  ldbl s = 0.0L; /* sum((x - y)^2) */
  for (int64_t k = 0; k < d; ++k) { double df = x[k] - y[k]; s += (ldbl)(df * df); }
  double ss = (double)s;
  if (id == K_SQREXP) {
    double r = sqrt(ss), l = v[0];
    out[0] = r_pow(r, 2.0) / r_pow(l, 3.0) * exp(-r_pow(r, 2.0) / (r_pow(l, 2.0) * 2.0));
  } else if (id == K_GAMMAEXP) {
    double r = sqrt(ss), g = v[0], l = v[1];
    out[0] = -exp(-r_pow(r / l, g)) * r_pow(r / l, g) * log(r / l);
    out[1] = exp(-r_pow(r / l, g)) * g * r_pow(r, g) / r_pow(l, g + 1.0);
  } else { /* K_RATQUAD */
    double r = ss, a = v[0], l = v[1];
    double c = 2.0 * r_pow(l, 2.0) * a;
    out[0] = (r_pow(r / c + 1.0, -a) * (r - (c + r) * log(r / c + 1.0))) / (c + r);
    out[1] = (r * r_pow(r / c + 1.0, -a - 1.0)) / r_pow(l, 3.0);
  }
}

/* solve(K): LU with partial pivoting applied to the identity (what dgesv does).  Returns 1 on an exactly zero
 * pivot.  R additionally refuses when the estimated rcond < .Machine$double.eps; that estimate is not restated. */
static int lu_inverse(double* A, int64_t n, double* inv) {
  int64_t* piv = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
  if (!piv) return -1;
  for (int64_t k = 0; k < n; ++k) {
    int64_t pr = k;
    double best = fabs(A[k + k * n]);
    for (int64_t i = k + 1; i < n; ++i)
      if (fabs(A[i + k * n]) > best) { best = fabs(A[i + k * n]); pr = i; }
    piv[k] = pr;
    if (best == 0.0 || best != best) { free(piv); return 1; }
    if (pr != k)
      for (int64_t j = 0; j < n; ++j) { double t = A[k + j * n]; A[k + j * n] = A[pr + j * n]; A[pr + j * n] = t; }
    double pv = 1.0 / A[k + k * n];
    for (int64_t i = k + 1; i < n; ++i) A[i + k * n] *= pv;
    for (int64_t j = k + 1; j < n; ++j) {
      double akj = A[k + j * n];
      for (int64_t i = k + 1; i < n; ++i) A[i + j * n] -= A[i + k * n] * akj;
    }
  }
  for (int64_t c = 0; c < n; ++c) {
    double* b = inv + c * n;
    for (int64_t i = 0; i < n; ++i) b[i] = (i == c) ? 1.0 : 0.0;
    for (int64_t k = 0; k < n; ++k)
      if (piv[k] != k) { double t = b[k]; b[k] = b[piv[k]]; b[piv[k]] = t; }
    for (int64_t k = 0; k < n; ++k) /* L y = P b, unit lower */
      for (int64_t i = k + 1; i < n; ++i) b[i] -= A[i + k * n] * b[k];
    for (int64_t k = n - 1; k >= 0; --k) { /* U x = y */
      b[k] /= A[k + k * n];
      for (int64_t i = 0; i < k; ++i) b[i] -= A[i + k * n] * b[k];
    }
  }
  free(piv);
  return 0;
}

/* dens_deriv(v)  --  R/fit.R:126-139.  `par` is v (the kernel is evaluated with it in the kernel's argument order
 * :132, deriv with it in deriv's order :133).  grad has npar entries.  Returns 0, 1 (singular K), <0 (argument). */
ORACLE_API int oracle_fit_gradient(int id, const double* par, int npar, const double* X, int64_t d, int64_t n,
                                   const double* y, double* grad) {
  if (!(id == K_SQREXP || id == K_GAMMAEXP || id == K_POLYNOMIAL || id == K_RATQUAD) || !check_params(id, npar, d)) return -1;
  double* K = (double*)malloc(sizeof(double) * (size_t)(n * n));
  double* Kd = (double*)malloc(sizeof(double) * (size_t)(n * n * npar));
  double* Kinv = (double*)malloc(sizeof(double) * (size_t)(n * n));
  double* alpha = (double*)malloc(sizeof(double) * (size_t)n);
  if (!K || !Kd || !Kinv || !alpha) { free(K); free(Kd); free(Kinv); free(alpha); return -2; }
  for (int64_t i = 0; i < n; ++i)     /* :130-135 */
    for (int64_t j = 0; j < n; ++j) {
      double g[2] = {0.0, 0.0};
      K[i + j * n] = kernel_pair(id, par, npar, X + i * d, X + j * d, d);
      deriv_pair(id, par, X + i * d, X + j * d, d, g);
      for (int k = 0; k < npar; ++k) Kd[i + j * n + (int64_t)k * n * n] = g[k];
    }
  int rc = lu_inverse(K, n, Kinv);    /* :136  K_inv <- solve(K) */
  if (rc == 0) {
    for (int64_t i = 0; i < n; ++i) { /* :137  alpha <- K_inv %*% y */
      double s = 0.0;
      for (int64_t j = 0; j < n; ++j) s += Kinv[i + j * n] * y[j];
      alpha[i] = s;
    }
    for (int k = 0; k < npar; ++k) {  /* :138  0.5 * sum(diag(alpha %*% t(alpha) - K_inv) %*% K_deriv[,,i]) */
      ldbl total = 0.0L;
      for (int64_t c = 0; c < n; ++c) {
        double s = 0.0;
        for (int64_t r = 0; r < n; ++r) s += (alpha[r] * alpha[r] - Kinv[r + r * n]) * Kd[r + c * n + (int64_t)k * n * n];
        total += (ldbl)s;
      }
      grad[k] = (double)(0.5L * total);
    }
  }
  free(K); free(Kd); free(Kinv); free(alpha);
  return rc;
}

/* ================================================================================================
 * multivariate_normal (R/GPRclass.R:360-370) and the eigen() it falls back to.
 * ============================================================================================== */

/* eigen(A, symmetric = TRUE): classical row-cyclic Jacobi (Golub & Van Loan, Alg. 8.4.3); the lower triangle of A is
 * read.  values: decreasing; vectors: m x m, columns orthonormal (signs arbitrary, as LAPACK's are).  Returns sweeps. */
ORACLE_API int oracle_sym_eigen(const double* A, int64_t m, int64_t lda, double* values, double* vectors) {
  double* W = (double*)malloc(sizeof(double) * (size_t)(m * m));
  double* V = (double*)malloc(sizeof(double) * (size_t)(m * m));
  int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)m);
  if (!W || !V || !idx) { free(W); free(V); free(idx); return -1; }
  for (int64_t j = 0; j < m; ++j)
    for (int64_t i = 0; i < m; ++i) {
      W[i + j * m] = (i >= j) ? A[i + j * lda] : A[j + i * lda];
      V[i + j * m] = (i == j) ? 1.0 : 0.0;
    }
  int sweeps = 0;
  for (; sweeps < 60; ++sweeps) {
    ldbl off = 0.0L, tot = 0.0L;
    for (int64_t j = 0; j < m; ++j)
      for (int64_t i = 0; i < m; ++i) {
        ldbl v = (ldbl)W[i + j * m] * W[i + j * m];
        tot += v;
        if (i != j) off += v;
      }
    ldbl rel = (ldbl)m * 2.220446049250313e-16L; /* ||off||_F <= max(1e-15, m eps) ||A||_F: the rounding floor */
    if (rel < 1e-15L) rel = 1e-15L;
    if (off <= rel * rel * tot) break;
    for (int64_t p = 0; p < m - 1; ++p)
      for (int64_t q = p + 1; q < m; ++q) {
        double apq = W[p + q * m];
        if (apq == 0.0) continue;
        double tau = (W[q + q * m] - W[p + p * m]) / (2.0 * apq);
        double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        double c = 1.0 / sqrt(1.0 + t * t), sn = t * c;
        for (int64_t i = 0; i < m; ++i) { /* W <- W J, V <- V J */
          double a = W[i + p * m], b = W[i + q * m];
          W[i + p * m] = c * a - sn * b;
          W[i + q * m] = sn * a + c * b;
          a = V[i + p * m]; b = V[i + q * m];
          V[i + p * m] = c * a - sn * b;
          V[i + q * m] = sn * a + c * b;
        }
        for (int64_t j = 0; j < m; ++j) { /* W <- J^T W */
          double a = W[p + j * m], b = W[q + j * m];
          W[p + j * m] = c * a - sn * b;
          W[q + j * m] = sn * a + c * b;
        }
      }
  }
  for (int64_t i = 0; i < m; ++i) idx[i] = i;
  for (int64_t i = 1; i < m; ++i) { /* stable insertion sort, decreasing */
    int64_t k = idx[i], j = i;
    while (j > 0 && W[idx[j - 1] + idx[j - 1] * m] < W[k + k * m]) { idx[j] = idx[j - 1]; --j; }
    idx[j] = k;
  }
  for (int64_t k = 0; k < m; ++k) {
    values[k] = W[idx[k] + idx[k] * m];
    if (vectors)
      for (int64_t i = 0; i < m; ++i) vectors[i + k * m] = V[i + idx[k] * m];
  }
  free(W); free(V); free(idx);
  return sweeps;
}

/* L of multivariate_normal: t(chol(covariance)) (:362) or the eigen fallback (:363-368).  L: m x m.
 * Returns 1 (Cholesky), 2 (eigen), -4 (stopifnot(all(eigval > -tol * abs(eigval[1]))) failed), -1 (memory). */
ORACLE_API int oracle_mvn_factor(const double* cov, int64_t m, int64_t ld, double tol, double* L) {
  for (int64_t j = 0; j < m; ++j)
    for (int64_t i = 0; i < m; ++i) L[i + j * m] = (i >= j) ? cov[i + j * ld] : 0.0;
  if (oracle_potrf_lower(L, m, m) == 0) {
    zero_upper(L, m);
    return 1;
  }
  double* val = (double*)malloc(sizeof(double) * (size_t)m);
  double* vec = (double*)malloc(sizeof(double) * (size_t)(m * m));
  if (!val || !vec) { free(val); free(vec); return -1; }
  int rc = 2;
  if (oracle_sym_eigen(cov, m, ld, val, vec) < 0) rc = -1;
  for (int64_t k = 0; rc == 2 && k < m; ++k)
    if (!(val[k] > -tol * fabs(val[0]))) rc = -4;
  if (rc == 2)
    for (int64_t k = 0; k < m; ++k) {
      double sc = sqrt(val[k] > 0.0 ? val[k] : 0.0); /* sqrt(pmax(eigval, 0)) */
      for (int64_t i = 0; i < m; ++i) L[i + k * m] = vec[i + k * m] * sc;
    }
  free(val); free(vec);
  return rc;
}

/* drop(mean) + L %*% Z   (:369); Z, out: m x n */
ORACLE_API void oracle_affine_lz(const double* L, int64_t m, const double* mean, const double* Z, int64_t n, double* out) {
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < m; ++i) {
      double s = 0.0;
      for (int64_t k = 0; k < m; ++k) s += L[i + k * m] * Z[k + j * m];
      out[i + j * m] = mean[i] + s;
    }
}
