/*
 * gprc_call_shim.c -- the `.Call` layer between the R package `gprc` and libgprc_native.so.
 *
 * NOT COMPILED IN THIS REPOSITORY'S BUILD: the build image has no R toolchain (no Rinternals.h).  It is
 * the file a maintainer drops into the reference package's src/ (with `useDynLib(gprc, .registration =
 * TRUE)` in NAMESPACE and PKG_LIBS = -lgprc_native).  It only marshals: every pointer handed to the C ABI
 * is REAL(x) of a caller-owned SEXP, borrowed for the call; results are allocated here and filled by the
 * library; model handles live behind external pointers with a finalizer.  Status codes are turned into R
 * conditions AFTER all temporaries are released (Rf_error longjmps).
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>

#include "gprc_native.h"

static gprc_ctx* g_ctx = NULL;

static gprc_ctx* ctx(void) {
  if (!g_ctx && gprc_ctx_create(0, NULL, &g_ctx) != 0) Rf_error("gprc: %s", gprc_last_error());
  return g_ctx;
}

static void model_finalizer(SEXP ptr) {
  gprc_model* m = (gprc_model*)R_ExternalPtrAddr(ptr);
  if (m) { gprc_model_free(m); R_ClearExternalPtr(ptr); }
}

/* An EMPTY external pointer with the finalizer already registered.  Every fit entry point creates it (and every other R
 * object it returns) BEFORE the native fit runs and stores the model in it with R_SetExternalPtrAddr -- which does not
 * allocate -- straight after: an R allocation that longjmps can then never strand a fitted model (device memory). */
static SEXP new_model_ptr(void) {
  SEXP ptr = PROTECT(R_MakeExternalPtr(NULL, Rf_install("gprc_model"), R_NilValue));
  R_RegisterCFinalizerEx(ptr, model_finalizer, TRUE);
  UNPROTECT(1);
  return ptr;
}

static gprc_model* model_of(SEXP ptr) {
  gprc_model* m = (gprc_model*)R_ExternalPtrAddr(ptr);
  if (!m) Rf_error("gprc: model handle is NULL (object restored from a saved workspace?)");
  return m;
}

/* covariance_matrix(A, B, k) for a tagged kernel  --  R/GPRclass.R:355-357 */
SEXP gprc_R_kernel_matrix(SEXP kernel, SEXP params, SEXP A, SEXP B) {
  const int64_t d = Rf_nrows(A), nA = Rf_ncols(A), nB = Rf_ncols(B);
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, (int)nA, (int)nB));
  int rc = gprc_kernel_matrix(ctx(), Rf_asInteger(kernel), REAL(params), LENGTH(params), REAL(A), d, nA, REAL(B), nB, REAL(out), nA);
  UNPROTECT(1);
  if (rc != 0) Rf_error("gprc: %s", gprc_last_error());
  return out;
}

/* GPR$initialize  --  R/GPRclass.R:138-153.  Returns list(handle, noise, attempts, alpha, logp). */
SEXP gprc_R_gpr_fit(SEXP kernel, SEXP params, SEXP X, SEXP y, SEXP noise) {
  const int64_t d = Rf_nrows(X), n = Rf_ncols(X);
  gprc_model* m = NULL;
  double noise_used = 0.0, logp = 0.0;
  int attempts = 0;
  /* everything R allocates for the result exists before the model does (see new_model_ptr) */
  SEXP res = PROTECT(Rf_allocVector(VECSXP, 5));
  SEXP ptr = new_model_ptr();
  SET_VECTOR_ELT(res, 0, ptr);
  SEXP alpha = Rf_allocVector(REALSXP, (R_xlen_t)n);
  SET_VECTOR_ELT(res, 3, alpha);
  SEXP s_noise = Rf_allocVector(REALSXP, 1), s_logp, s_att;
  SET_VECTOR_ELT(res, 1, s_noise);
  s_att = Rf_allocVector(INTSXP, 1);
  SET_VECTOR_ELT(res, 2, s_att);
  s_logp = Rf_allocVector(REALSXP, 1);
  SET_VECTOR_ELT(res, 4, s_logp);
  int rc = gprc_gpr_fit_retry(ctx(), Rf_asInteger(kernel), REAL(params), LENGTH(params), REAL(X), d, n, REAL(y), Rf_asReal(noise), &m,
                              &noise_used, &attempts);
  if (rc != 0) {
    UNPROTECT(1);
    if (rc == GPRC_ERR_NOT_PD)
      Rf_error("Inputs lead to non positive definite covariance matrix. Try using a larger noise or a smaller lengthscale.");
    Rf_error("gprc: %s", gprc_last_error());
  }
  R_SetExternalPtrAddr(ptr, m);   /* owned by the finalizer from here on; no allocation happened in between */
  gprc_gpr_get_alpha(m, REAL(alpha));
  gprc_gpr_get_logp(m, &logp);
  REAL(s_noise)[0] = noise_used;
  INTEGER(s_att)[0] = attempts;
  REAL(s_logp)[0] = logp;
  UNPROTECT(1);
  return res;
}

/* GPR$predict  --  R/GPRclass.R:155-170.  pointwise: n* x 2 matrix; else list(mean n* x 1, cov n* x n*). */
SEXP gprc_R_gpr_predict(SEXP handle, SEXP X_star, SEXP pointwise) {
  gprc_model* m = model_of(handle);
  const int64_t ns = Rf_ncols(X_star);
  const int pw = Rf_asLogical(pointwise);
  SEXP res;
  int rc;
  if (pw) {
    res = PROTECT(Rf_allocMatrix(REALSXP, (int)ns, 2));
    rc = gprc_gpr_predict(m, REAL(X_star), ns, 1, REAL(res), REAL(res) + ns);
    UNPROTECT(1);
  } else {
    res = PROTECT(Rf_allocVector(VECSXP, 2));
    SEXP mean = PROTECT(Rf_allocMatrix(REALSXP, (int)ns, 1));
    SEXP cov = PROTECT(Rf_allocMatrix(REALSXP, (int)ns, (int)ns));
    rc = gprc_gpr_predict(m, REAL(X_star), ns, 0, REAL(mean), REAL(cov));
    SET_VECTOR_ELT(res, 0, mean);
    SET_VECTOR_ELT(res, 1, cov);
    UNPROTECT(3);
  }
  if (rc != 0) Rf_error("gprc: %s", gprc_last_error());
  return res;
}

/* ---- multi-GPU from R's one process: options(gprc.devices = c(0, 1, ..., 7)) selects it (native.R) ------------------
 * One gprc_mgpu per distinct (devices, flags) request, ALL kept until the package is unloaded: creating the streams / RCCL
 * communicators is not free, and GPR objects fitted earlier hold gprc_mgpu_model handles into theirs -- destroying a
 * gprc_mgpu when options(gprc.devices / gprc.rccl) change would leave those objects' $predict and their finalizers on freed
 * ranks (the library additionally refuses a model whose gprc_mgpu is gone: include/gprc_native.h, "Lifetime").  The model
 * handle is a gprc_mgpu_model behind its own external-pointer class. */
#define GPRC_MGPU_SLOTS 16
static struct { gprc_mgpu* mg; int devs[64], n, flags; } g_mgpus[GPRC_MGPU_SLOTS];
static int g_mgpu_count = 0;

static gprc_mgpu* mgpu_for(SEXP devices, SEXP flags) {
  const int n = LENGTH(devices), fl = Rf_asInteger(flags);
  if (n < 1 || n > 64) Rf_error("gprc: 1..64 devices");
  for (int k = 0; k < g_mgpu_count; ++k) {
    int same = n == g_mgpus[k].n && fl == g_mgpus[k].flags;
    for (int i = 0; same && i < n; ++i) same = INTEGER(devices)[i] == g_mgpus[k].devs[i];
    if (same) return g_mgpus[k].mg;
  }
  if (g_mgpu_count == GPRC_MGPU_SLOTS) Rf_error("gprc: more than %d distinct (gprc.devices, flags) settings in one session", GPRC_MGPU_SLOTS);
  gprc_mgpu* mg = NULL;
  if (gprc_mgpu_create(INTEGER(devices), n, fl, &mg) != 0) Rf_error("gprc: %s", gprc_last_error());
  g_mgpus[g_mgpu_count].mg = mg;
  for (int i = 0; i < n; ++i) g_mgpus[g_mgpu_count].devs[i] = INTEGER(devices)[i];
  g_mgpus[g_mgpu_count].n = n;
  g_mgpus[g_mgpu_count].flags = fl;
  ++g_mgpu_count;
  return mg;
}

static void mgpu_model_finalizer(SEXP ptr) {
  gprc_mgpu_model* m = (gprc_mgpu_model*)R_ExternalPtrAddr(ptr);
  if (m) { gprc_mgpu_model_free(m); R_ClearExternalPtr(ptr); }
}

/* GPR$initialize over several GPUs  --  R/GPRclass.R:138-153.  Returns list(handle, noise, attempts, alpha, logp). */
SEXP gprc_R_mgpu_gpr_fit(SEXP devices, SEXP flags, SEXP kernel, SEXP params, SEXP X, SEXP y, SEXP noise) {
  const int64_t d = Rf_nrows(X), n = Rf_ncols(X);
  gprc_mgpu* mg = mgpu_for(devices, flags);
  gprc_mgpu_model* m = NULL;
  double noise_used = 0.0, logp = 0.0;
  int attempts = 0;
  SEXP res = PROTECT(Rf_allocVector(VECSXP, 5));   /* all R allocations before the model exists (see new_model_ptr) */
  SEXP ptr = R_MakeExternalPtr(NULL, Rf_install("gprc_mgpu_model"), R_NilValue);
  SET_VECTOR_ELT(res, 0, ptr);
  R_RegisterCFinalizerEx(ptr, mgpu_model_finalizer, TRUE);
  SEXP alpha = Rf_allocVector(REALSXP, (R_xlen_t)n);
  SET_VECTOR_ELT(res, 3, alpha);
  SEXP s_noise = Rf_allocVector(REALSXP, 1);
  SET_VECTOR_ELT(res, 1, s_noise);
  SEXP s_att = Rf_allocVector(INTSXP, 1);
  SET_VECTOR_ELT(res, 2, s_att);
  SEXP s_logp = Rf_allocVector(REALSXP, 1);
  SET_VECTOR_ELT(res, 4, s_logp);
  int rc = gprc_mgpu_gpr_fit_retry(mg, Rf_asInteger(kernel), REAL(params), LENGTH(params), REAL(X), d, n, REAL(y), Rf_asReal(noise), &m,
                                   &noise_used, &attempts);
  if (rc != 0) {
    UNPROTECT(1);
    if (rc == GPRC_ERR_NOT_PD)
      Rf_error("Inputs lead to non positive definite covariance matrix. Try using a larger noise or a smaller lengthscale.");
    Rf_error("gprc: %s", gprc_last_error());
  }
  R_SetExternalPtrAddr(ptr, m);
  gprc_mgpu_gpr_get_alpha(m, REAL(alpha));
  gprc_mgpu_gpr_get_logp(m, &logp);
  REAL(s_noise)[0] = noise_used;
  INTEGER(s_att)[0] = attempts;
  REAL(s_logp)[0] = logp;
  UNPROTECT(1);
  return res;
}

/* GPR$predict(X_star, pointwise_var = TRUE) over several GPUs: the test points are sliced over the ranks */
SEXP gprc_R_mgpu_gpr_predict(SEXP handle, SEXP X_star) {
  gprc_mgpu_model* m = (gprc_mgpu_model*)R_ExternalPtrAddr(handle);
  if (!m) Rf_error("gprc: model handle is NULL (object restored from a saved workspace?)");
  const int64_t ns = Rf_ncols(X_star);
  SEXP res = PROTECT(Rf_allocMatrix(REALSXP, (int)ns, 2));
  int rc = gprc_mgpu_gpr_predict(m, REAL(X_star), ns, REAL(res), REAL(res) + ns);
  UNPROTECT(1);
  if (rc != 0) Rf_error("gprc: %s", gprc_last_error());
  return res;
}

/* `$L` active binding: materialised lazily (n x n doubles over PCIe) */
SEXP gprc_R_model_L(SEXP handle) {
  gprc_model* m = model_of(handle);
  int64_t n = 0, d = 0;
  gprc_model_dims(m, &n, &d);
  SEXP L = PROTECT(Rf_allocMatrix(REALSXP, (int)n, (int)n));
  int rc = gprc_model_get_L(m, REAL(L), n);
  UNPROTECT(1);
  if (rc != 0) Rf_error("gprc: %s", gprc_last_error());
  return L;
}

/* GPC$initialize  --  R/GPCclass.R:66-107.  Returns list(handle, f_hat, logq, iterations). */
SEXP gprc_R_gpc_fit(SEXP kernel, SEXP params, SEXP X, SEXP y, SEXP epsilon) {
  const int64_t d = Rf_nrows(X), n = Rf_ncols(X);
  gprc_model* m = NULL;
  int iters = 0;
  double logq = 0.0;
  SEXP res = PROTECT(Rf_allocVector(VECSXP, 4));   /* all R allocations first (see new_model_ptr) */
  SEXP ptr = new_model_ptr();
  SET_VECTOR_ELT(res, 0, ptr);
  SEXP f = Rf_allocVector(REALSXP, (R_xlen_t)n);
  SET_VECTOR_ELT(res, 1, f);
  SEXP s_logq = Rf_allocVector(REALSXP, 1);
  SET_VECTOR_ELT(res, 2, s_logq);
  SEXP s_it = Rf_allocVector(INTSXP, 1);
  SET_VECTOR_ELT(res, 3, s_it);
  int rc = gprc_gpc_fit(ctx(), Rf_asInteger(kernel), REAL(params), LENGTH(params), REAL(X), d, n, REAL(y), Rf_asReal(epsilon), 0,
                        GPRC_GPC_REFERENCE_STOP, &m, &iters);
  if (rc != 0) {
    UNPROTECT(1);
    if (rc == GPRC_ERR_DIVERGED) Rf_error("Apparently does not converge.");
    Rf_error("gprc: %s", gprc_last_error());
  }
  R_SetExternalPtrAddr(ptr, m);
  gprc_gpc_get_f_hat(m, REAL(f));
  gprc_gpc_get_logq(m, &logq);
  REAL(s_logq)[0] = logq;
  INTEGER(s_it)[0] = iters;
  UNPROTECT(1);
  return res;
}

/* whole GPC$predict_class on the device (latent stage + batched class-probability quadrature)  --  R/GPCclass.R:108-118 */
SEXP gprc_R_gpc_predict_class(SEXP handle, SEXP X_star) {
  gprc_model* m = model_of(handle);
  const int64_t ns = Rf_ncols(X_star);
  SEXP res = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)ns));
  int rc = gprc_gpc_predict_class(m, REAL(X_star), ns, REAL(res));
  UNPROTECT(1);
  if (rc != 0) Rf_error("gprc: %s", gprc_last_error());
  return res;
}

/* fs_bar, Vfs of GPC$predict_class  --  R/GPCclass.R:109-115; for callers that keep the integrate() loop in R */
SEXP gprc_R_gpc_predict_latent(SEXP handle, SEXP X_star) {
  gprc_model* m = model_of(handle);
  const int64_t ns = Rf_ncols(X_star);
  SEXP res = PROTECT(Rf_allocMatrix(REALSXP, (int)ns, 2));
  int rc = gprc_gpc_predict_latent(m, REAL(X_star), ns, REAL(res), REAL(res) + ns);
  UNPROTECT(1);
  if (rc != 0) Rf_error("gprc: %s", gprc_last_error());
  return res;
}

/* dens(v) of fit()  --  R/fit.R:117-124.  Returns the log marginal likelihood; not positive definite -> R error,
 * which optim_until_error's tryCatch turns into the -10000 sentinel exactly as it does for chol()'s error. */
SEXP gprc_R_log_marginal(SEXP kernel, SEXP params, SEXP X, SEXP y, SEXP noise) {
  const int64_t d = Rf_nrows(X), n = Rf_ncols(X);
  double logp = 0.0;
  int rc = gprc_gpr_log_marginal(ctx(), Rf_asInteger(kernel), REAL(params), LENGTH(params), REAL(X), d, n, REAL(y), Rf_asReal(noise), &logp);
  if (rc > 0) Rf_error("the leading minor of order %d is not positive definite", rc);
  if (rc != 0) Rf_error("gprc: %s", gprc_last_error());
  return Rf_ScalarReal(logp);
}

/* dens_deriv(v) of fit()  --  R/fit.R:126-139 (as written; see gprc_native.h) */
SEXP gprc_R_fit_gradient(SEXP kernel, SEXP params, SEXP X, SEXP y) {
  const int64_t d = Rf_nrows(X), n = Rf_ncols(X);
  SEXP g = PROTECT(Rf_allocVector(REALSXP, LENGTH(params)));
  int rc = gprc_fit_gradient(ctx(), Rf_asInteger(kernel), REAL(params), LENGTH(params), REAL(X), d, n, REAL(y), REAL(g));
  UNPROTECT(1);
  if (rc > 0) Rf_error("system is computationally singular (leading minor of order %d)", rc);   /* solve(K)'s condition */
  if (rc != 0) Rf_error("gprc: %s", gprc_last_error());
  return g;
}

/* multivariate_normal(n, mean, covariance, tol)  --  R/GPRclass.R:360-370.  Z = matrix(rnorm(n * length(mean)), nrow =
 * length(mean)) is drawn by the R caller, so set.seed() keeps governing the draws. */
SEXP gprc_R_mvn_sample(SEXP mean, SEXP covariance, SEXP tol, SEXP Z) {
  const int64_t m = Rf_nrows(covariance), nd = Rf_ncols(Z);
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, (int)m, (int)nd));
  int method = 0;
  int rc = gprc_mvn_sample(ctx(), REAL(covariance), m, m, REAL(mean), Rf_asReal(tol), REAL(Z), nd, REAL(out), &method);
  UNPROTECT(1);
  if (rc == GPRC_ERR_NOT_PD) Rf_error("all(eigval > -tol * abs(eigval[1])) is not TRUE");
  if (rc != 0) Rf_error("gprc: %s", gprc_last_error());
  return out;
}

/* combine_all(lst)  --  R/simulation.R:338-349; `values` = unlist(lst), `lengths` = lengths(lst) as doubles */
SEXP gprc_R_combine_all(SEXP values, SEXP lengths) {
  const int d = LENGTH(lengths);
  int64_t len[64], total = 1;
  if (d < 1 || d > 64) Rf_error("gprc: combine_all supports 1..64 axes");
  for (int k = 0; k < d; ++k) { len[k] = (int64_t)REAL(lengths)[k]; total *= len[k]; }
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, d, (int)total));
  int rc = gprc_combine_all(ctx(), REAL(values), len, d, REAL(out));
  UNPROTECT(1);
  if (rc != 0) Rf_error("gprc: %s", gprc_last_error());
  return out;
}

SEXP gprc_R_device_count(void) {
  int c = 0;
  gprc_device_count(&c);
  return Rf_ScalarInteger(c);
}

static const R_CallMethodDef call_methods[] = {
    {"gprc_R_kernel_matrix", (DL_FUNC)&gprc_R_kernel_matrix, 4},
    {"gprc_R_gpr_fit", (DL_FUNC)&gprc_R_gpr_fit, 5},
    {"gprc_R_gpr_predict", (DL_FUNC)&gprc_R_gpr_predict, 3},
    {"gprc_R_model_L", (DL_FUNC)&gprc_R_model_L, 1},
    {"gprc_R_gpc_fit", (DL_FUNC)&gprc_R_gpc_fit, 5},
    {"gprc_R_gpc_predict_latent", (DL_FUNC)&gprc_R_gpc_predict_latent, 2},
    {"gprc_R_gpc_predict_class", (DL_FUNC)&gprc_R_gpc_predict_class, 2},
    {"gprc_R_log_marginal", (DL_FUNC)&gprc_R_log_marginal, 5},
    {"gprc_R_fit_gradient", (DL_FUNC)&gprc_R_fit_gradient, 4},
    {"gprc_R_mvn_sample", (DL_FUNC)&gprc_R_mvn_sample, 4},
    {"gprc_R_combine_all", (DL_FUNC)&gprc_R_combine_all, 2},
    {"gprc_R_device_count", (DL_FUNC)&gprc_R_device_count, 0},
    {"gprc_R_mgpu_gpr_fit", (DL_FUNC)&gprc_R_mgpu_gpr_fit, 7},
    {"gprc_R_mgpu_gpr_predict", (DL_FUNC)&gprc_R_mgpu_gpr_predict, 2},
    {NULL, NULL, 0}};

void R_init_gprc(DllInfo* dll) {
  R_registerRoutines(dll, NULL, call_methods, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}

void R_unload_gprc(DllInfo* dll) {
  (void)dll;
  /* Finalizers of GPR objects may still be pending when the package is unloaded: the library keeps registries of live contexts /
   * gprc_mgpu objects, so a model freed afterwards hands its memory back without touching what is destroyed here. */
  for (int k = 0; k < g_mgpu_count; ++k) gprc_mgpu_destroy(g_mgpus[k].mg);
  g_mgpu_count = 0;
  if (g_ctx) { gprc_ctx_destroy(g_ctx); g_ctx = NULL; }
}
