/*
 * gprc_native.h -- C ABI of the MI355X-native GP predict hot path (libgprc_native.so).
 *
 * Drop-in boundary for the R package `gprc` (MoHawastaken/Gaussian-Process-Regression).  The
 * reference has NO FFI today (pure R closures / R6 methods); each entry point below names the
 * reference interface (file:line under the reference root) whose arithmetic it replaces.  The `.Call`
 * shim a maintainer adds on the R side is shown in INTEGRATION.md; the Python host mirror used by the
 * tests binds the same symbols through ctypes.
 *
 * Conventions
 *  - plain C, no R / torch / C++ types; sizes are int64_t; every matrix is column-major fp64;
 *  - X is d x n with one observation per COLUMN (R/GPRclass.R:29,132,137): point i = X[i*d .. i*d+d);
 *  - every data pointer may be HOST memory (the `.Call` case: REAL(x)) or DEVICE memory of the
 *    context's GPU (the resident-data case: bench.py, the multi-GPU driver).  The library asks the HIP
 *    runtime which it is; host data is staged through the context's stream, device data is used in
 *    place.  Inputs are borrowed for the duration of the call only;
 *  - ORDERING of device data: every kernel of a call runs on the CONTEXT'S stream (and on side streams the
 *    library orders behind it), never on the null stream, and a context that creates its own stream creates it
 *    non-blocking -- it is NOT ordered with the caller's streams.  Device inputs must therefore be complete on,
 *    or ordered before, the context's stream when the call is made, and device outputs are ready on that stream
 *    when it returns (the fit / predict entry points also synchronise it; the gprc_dev_* building blocks are
 *    asynchronous).  Two ways to satisfy it: (a) pass YOUR stream to gprc_ctx_create -- data produced and
 *    consumed on that stream needs no synchronisation at all; (b) keep the context's own stream and finish the
 *    producers first (hipStreamSynchronize / hipDeviceSynchronize, or an event the context's stream cannot see
 *    is not enough).  A copy still in flight on another stream when the first kernel starts is a data race
 *    (seen once in this repository's own tests: a 1.8 GB clone overwrote the first panel's update);
 *  - return value: 0 = ok; > 0 = LAPACK-style info (order of the first leading minor that is not
 *    positive definite -- what R's chol() reports, R/GPRclass.R:142); < 0 = gprc_status error, with
 *    text in gprc_last_error().  The library never aborts, exits or throws across this boundary;
 *  - a gprc_ctx is bound to one GPU and one HIP stream; calls on one context are serialised by the
 *    caller (R's single thread).  Different contexts may be used from different threads/processes.
 */
#ifndef GPRC_NATIVE_H
#define GPRC_NATIVE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPRC_ABI_VERSION 1

#if defined(__GNUC__)
#define GPRC_API __attribute__((visibility("default")))
#else
#define GPRC_API
#endif

/* Kernel ids and parameter vectors (the `...` of cov_func, R/GPRclass.R:424-427):
 *   GPRC_CONSTANT     params = {c}               R/GPRclass.R:382
 *   GPRC_LINEAR       params = {sigma} or sigma[d] R/GPRclass.R:386
 *   GPRC_POLYNOMIAL   params = {sigma, p}        R/GPRclass.R:390
 *   GPRC_SQREXP       params = {l}               R/GPRclass.R:394
 *   GPRC_GAMMAEXP     params = {l, gamma}        R/GPRclass.R:398
 *   GPRC_RATQUAD      params = {l, alpha}        R/GPRclass.R:402 */
typedef enum {
  GPRC_CONSTANT = 0,
  GPRC_LINEAR = 1,
  GPRC_POLYNOMIAL = 2,
  GPRC_SQREXP = 3,
  GPRC_GAMMAEXP = 4,
  GPRC_RATQUAD = 5
} gprc_kernel_id;

typedef enum {
  GPRC_OK = 0,
  GPRC_ERR_ARG = -1,      /* bad argument (the reference's stopifnot() failures) */
  GPRC_ERR_HIP = -2,      /* HIP runtime error */
  GPRC_ERR_NOMEM = -3,    /* allocation failed */
  GPRC_ERR_NOT_PD = -4,   /* all ten jitter attempts failed: "Inputs lead to non positive definite
                             covariance matrix..." (R/GPRclass.R:149) */
  GPRC_ERR_DIVERGED = -5, /* GPC: "Apparently does not converge." (R/GPCclass.R:90-91) */
  GPRC_ERR_MAXITER = -6,  /* GPC: safety cap on IRLS iterations hit (the reference loops forever) */
  GPRC_ERR_NO_DEVICE = -7 /* no usable gfx950 device */
} gprc_status;

typedef struct gprc_ctx gprc_ctx;     /* one GPU + one stream + scratch */
typedef struct gprc_model gprc_model; /* fitted GPR or GPC: device-resident X, y, L, alpha / f_hat */

/* ---- library / device --------------------------------------------------------------------- */
GPRC_API int gprc_abi_version(void);
/* text of the last error on the calling thread ("" if none); valid until the next failing call */
GPRC_API const char* gprc_last_error(void);
GPRC_API int gprc_device_count(int* count_out);
/* stream: a hipStream_t owned by the caller (e.g. a torch.cuda.Stream's handle), or NULL to let the
 * context create and own a non-blocking one (see "ORDERING of device data" above; note that the legacy
 * default stream's handle IS NULL, so it cannot be handed over this way). */
GPRC_API int gprc_ctx_create(int device, void* stream, gprc_ctx** ctx_out);
GPRC_API int gprc_ctx_destroy(gprc_ctx* ctx);
/* GPR$predict / GPC$predict_class work through the test points in chunks of rows of K*^T.  The chunk is sized from
 * GPRC_CHUNK_BYTES (default 40 GiB) but never beyond what hipMemGetInfo reports free; if an allocation still fails
 * the chunk is halved and tried again (down to 256 rows) before the call returns GPRC_ERR_NOMEM.  Results do not
 * depend on the chunking, bit for bit.
 * Gives the device memory a context keeps for reuse back to the driver: the free-list of blocks released by earlier
 * calls (exact-size reuse; fit() evaluates the same n over and over; capped by GPRC_POOL_BYTES, default 16 GiB, 0 = off)
 * and the predict workspaces.  Never needed for correctness. */
GPRC_API int gprc_ctx_trim(gprc_ctx* ctx);
GPRC_API int gprc_ctx_synchronize(gprc_ctx* ctx);

/* ---- L1: covariance-function layer --------------------------------------------------------- */
/* covariance_matrix(A, B, k) (R/GPRclass.R:355-357): out[i + j*ld_out] = k(A[,i], B[,j]),
 * nA x nB.  ld_out >= nA. */
GPRC_API int gprc_kernel_matrix(gprc_ctx* ctx, int kernel, const double* params, int n_params, const double* A, int64_t d,
                       int64_t nA, const double* B, int64_t nB, double* out, int64_t ld_out);
/* the closure contract itself (R/GPRclass.R:353-354, the .matrix methods :382-402): for two d x m
 * matrices, out[c] = k(x[,c], y[,c]).  GPR$predict uses it for k(X*,X*) (R/GPRclass.R:164). */
GPRC_API int gprc_kernel_colwise(gprc_ctx* ctx, int kernel, const double* params, int n_params, const double* x,
                        const double* y, int64_t d, int64_t m, double* out);

/* ---- L2: GPR ------------------------------------------------------------------------------ */
/* One Cholesky attempt of GPR$initialize (R/GPRclass.R:138,142,152-153): K = k(X,X),
 * L = chol(K + noise*I)^T, alpha = L^-T L^-1 y, logp.  Returns 0 and a model, or info > 0 (no model)
 * when K + noise*I is not positive definite. */
GPRC_API int gprc_gpr_fit(gprc_ctx* ctx, int kernel, const double* params, int n_params, const double* X, int64_t d,
                 int64_t n, const double* y, double noise, gprc_model** model_out);
/* The whole of R/GPRclass.R:139-151: attempts noise, noise+0.01, ..., noise+0.09; *noise_used is
 * the `$noise` the reference stores, *attempts > 1 means the reference would warn (:144).
 * Returns GPRC_ERR_NOT_PD when all ten fail (:149). */
GPRC_API int gprc_gpr_fit_retry(gprc_ctx* ctx, int kernel, const double* params, int n_params, const double* X, int64_t d,
                       int64_t n, const double* y, double noise, gprc_model** model_out, double* noise_used,
                       int* attempts);
/* dens(v), the objective of fit() (R/fit.R:117-124): the log marginal likelihood of (kernel, params) on (X, y, noise),
 * i.e. kernel fill + Cholesky + alpha + logp without keeping a model.  Returns 0 and *logp_out, or info > 0 when
 * K + noise*I is not positive definite.  The reference guards with min(det(leading minors)) > 0 (:119, O(n^4));
 * the Cholesky's own info is the same test in exact arithmetic. */
GPRC_API int gprc_gpr_log_marginal(gprc_ctx* ctx, int kernel, const double* params, int n_params, const double* X, int64_t d,
                          int64_t n, const double* y, double noise, double* logp_out);
/* dens_deriv(v), the gradient fit() hands to optim(method = "BFGS") (R/fit.R:126-139), reproduced with its quirks:
 * K is the NOISE-FREE kernel matrix, alpha = K^-1 y, and component i is
 *   0.5 * sum( diag(alpha alpha^T - K^-1) %*% dK/dv_i )      -- a vector %*% matrix, not a trace (:138);
 * dK/dv comes from cov_dict$<kernel>$deriv (R/fit.R:4-31) with v bound positionally in deriv's OWN argument order
 * (gammaexp: (gamma, l), rationalquadratic: (alpha, l) -- the opposite of the kernels' own order, :10,25).
 * Defined for sqrexp, gammaexp, polynomial, rationalquadratic (:125).  gammaexp's first component is NaN by
 * construction (0 * log 0 on the diagonal).  grad_out: n_params doubles in HOST memory.  Returns 0, or info > 0 when
 * K is not (numerically) positive definite -- the reference's solve(K) fails there with "computationally singular". */
GPRC_API int gprc_fit_gradient(gprc_ctx* ctx, int kernel, const double* params, int n_params, const double* X, int64_t d,
                      int64_t n, const double* y, double* grad_out);
/* GPR$predict (R/GPRclass.R:155-170).  X_star is d x n_star.
 * pointwise != 0: mean_out[n_star], var_out[n_star] = k(x*,x*) - colSums(v*v)      (:164-165)
 * pointwise == 0: mean_out[n_star], var_out = n_star x n_star K(X*,X*) - t(v) %*% v (:167-168) */
GPRC_API int gprc_gpr_predict(gprc_model* model, const double* X_star, int64_t n_star, int pointwise, double* mean_out,
                     double* var_out);
/* active bindings (R/GPRclass.R:230-280).  get_L materialises the n x n lower factor (upper = 0,
 * as t(chol(.)) has it) -- lazily, only when asked: it is 32 GiB at n = 65536. */
GPRC_API int gprc_model_dims(const gprc_model* model, int64_t* n_out, int64_t* d_out);
GPRC_API int gprc_model_get_L(gprc_model* model, double* L_out, int64_t ld_out);
GPRC_API int gprc_gpr_get_alpha(gprc_model* model, double* alpha_out);
GPRC_API int gprc_gpr_get_logp(gprc_model* model, double* logp_out);
GPRC_API int gprc_gpr_get_noise(gprc_model* model, double* noise_out);
GPRC_API int gprc_model_free(gprc_model* model);

/* ---- L2: GPC ------------------------------------------------------------------------------ */
/* GPC$initialize (R/GPCclass.R:66-107): Laplace mode by Newton/IRLS; y in {-1,+1}.
 * max_iter <= 0 selects 1000.  *iters_out = the "Convergence after %s iterations" count (:98).
 * flags: GPRC_GPC_REFERENCE_STOP reproduces the reference's stop rule of :90-91 verbatim
 * (`least_objective + 10 < objective` -> GPRC_ERR_DIVERGED; note it fires whenever the maximised objective
 * IMPROVES by more than 10 over iteration 1, i.e. for most problems beyond a few hundred points);
 * 0 switches that rule off and iterates to |delta objective| < epsilon. */
#define GPRC_GPC_REFERENCE_STOP 1
GPRC_API int gprc_gpc_fit(gprc_ctx* ctx, int kernel, const double* params, int n_params, const double* X, int64_t d,
                 int64_t n, const double* y, double epsilon, int max_iter, int flags, gprc_model** model_out,
                 int* iters_out);
/* the hot part of GPC$predict_class (R/GPCclass.R:109-115): fs_bar and Vfs for X_star. */
GPRC_API int gprc_gpc_predict_latent(gprc_model* model, const double* X_star, int64_t n_star, double* fs_bar_out,
                            double* Vfs_out);
/* GPC$predict_class (R/GPCclass.R:108-118): prob_out[i] = integral of sigmoid(z) dnorm(z, fs_bar[i], sd = Vfs[i]) dz.
 * The latent stage is gprc_gpc_predict_latent(); the 1-D integrals (stats::integrate / QUADPACK in the reference,
 * rel.tol 1.2e-4) run as one batched device kernel (composite Gauss-Legendre, ~1e-14).  The reference's use of the
 * VARIANCE as dnorm's sd (:117) is reproduced.  Vfs <= 0 gives NaN. */
GPRC_API int gprc_gpc_predict_class(gprc_model* model, const double* X_star, int64_t n_star, double* prob_out);
/* the integral alone, for callers that already hold fs_bar / Vfs */
GPRC_API int gprc_class_probability(gprc_ctx* ctx, const double* fs_bar, const double* Vfs, int64_t n, double* prob_out);
GPRC_API int gprc_gpc_get_f_hat(gprc_model* model, double* f_hat_out);
GPRC_API int gprc_gpc_get_logq(gprc_model* model, double* logq_out);

/* ---- device-level building blocks (multi-GPU driver, bench) ------------------------------- *
 * All pointers below are DEVICE pointers on the context's GPU.  The factor lives in the packed
 * block-column layout described in DESIGN.md: n_pad = gprc_pad(n); panel p holds rows
 * [p*NB, n_pad) x columns [p*NB, (p+1)*NB), column-major with leading dimension n_pad - p*NB, at
 * element offset gprc_panel_offset(n_pad, p); NB = gprc_panel_width().  Every panel is therefore one
 * contiguous buffer: the unit RCCL broadcasts.  winv holds one 128 x 128 block per 128 columns
 * (the inverse of the diagonal block of L), gprc_winv_size(n_pad) doubles. */
GPRC_API int64_t gprc_panel_width(void);
GPRC_API int64_t gprc_pad(int64_t n);
GPRC_API int64_t gprc_panel_count(int64_t n_pad);
GPRC_API int64_t gprc_panel_offset(int64_t n_pad, int64_t p);
GPRC_API int64_t gprc_panel_elems(int64_t n_pad, int64_t p);
GPRC_API int64_t gprc_packed_size(int64_t n_pad);
GPRC_API int64_t gprc_winv_size(int64_t n_pad);
/* fill panel p of K + noise*I (identity in the padding) */
GPRC_API int gprc_dev_fill_panel(gprc_ctx* ctx, int kernel, const double* params_host, int n_params, const double* X,
                        int64_t d, int64_t n, int64_t n_pad, double noise, double* packed, int64_t p);
/* factor panel p in place (diagonal blocks in LDS, panel solves, in-panel updates); info_dev is a
 * device int the first non-PD column (1-based) is written to (must be zeroed by the caller) */
GPRC_API int gprc_dev_factor_panel(gprc_ctx* ctx, double* packed, int64_t n_pad, int64_t p, double* winv, int* info_dev);
/* the same in steps, j = 0..3 (128 columns each, in order).  part 1 (or 0): block j's columns receive the contributions of
 * the blocks 0..j-1 of the panel (one K = 128 j pass), the diagonal block is factored + inverted and the rows below it
 * are solved -- after it the columns [128 j, 128 (j+1)) of the panel, a contiguous range of the packed buffer, are
 * final, so a multi-rank driver can start broadcasting them.  part 2: nothing (accepted so that drivers written for a
 * right-looking "factor, then update the rest of the panel" split keep working). */
GPRC_API int gprc_dev_factor_subpanel(gprc_ctx* ctx, double* packed, int64_t n_pad, int64_t p, int j, int part, double* winv,
                             int* info_dev);
/* All panels of an already filled packed matrix on ONE GPU, asynchronously on the context's stream (the one-rank form of
 * the factor_panel / update_trailing sweep: the panels in groups, a left-looking pass per group, inside the group two persistent
 * launches side by side -- the factor service (the panels' dependent chains) and the sweep kernel (every other tile and strip of the
 * group, dealt by tickets; GPRC_SWEEP=0: one launch per panel instead); one group below n_pad = 20480): results bit-identical to that sweep.
 * Below n_pad = 20480 the chain's 128^3 tiles are split in 32-row slices over four helper workgroups of the service (GPRC_CHAIN_SPLIT=0 / 1
 * forces the form); from n_pad = 13312 the service's 4-wave roles share their CUs with one sweep workgroup each (GPRC_SERVICE_SHARE=0 / 1).
 * Every combination gives the same bits.
 * info_dev: one device int, zeroed by the caller, receives LAPACK's info (first non-PD leading minor) if any.
 * inv: NULL, or gprc_solve_inv_size(n_pad) doubles that receive what gprc_dev_solve_prepare would compute for all panels. */
GPRC_API int gprc_dev_factor_all(gprc_ctx* ctx, double* packed, int64_t n_pad, double* winv, int* info_dev, double* inv);
/* The factor service (one persistent launch carrying the panel chains beside the caller's kernels) needs kernels of two streams to
 * run CONCURRENTLY.  Where they cannot -- a tool that serialises dispatches, e.g. rocprofv3 --pmc -- its device-side waits run out
 * after a few seconds and info becomes GPRC_INFO_WAIT_TIMEOUT (-99).  The fit entry points (gprc_gpr_fit*, gprc_gpc_fit, ...) then
 * switch the service off for the process, rebuild the matrix and factor again with one launch per panel; callers of
 * gprc_dev_factor_all (asynchronous: info stays on the device) do the same with this switch: on = 0 off, 1 on, -1 query;
 * returns the previous state.  (GPRC_SERVICE=0 in the environment never uses it.) */
GPRC_API int gprc_factor_service(int on);
#define GPRC_INFO_WAIT_TIMEOUT (-99)
/* trailing update of panels q = q_begin, q_begin + q_stride, ... < q_end with factored panel p */
GPRC_API int gprc_dev_update_trailing(gprc_ctx* ctx, double* packed, int64_t n_pad, int64_t p, int64_t q_begin,
                             int64_t q_end, int64_t q_stride);
/* the same for a RANGE of source panels [p_begin, p_end) in one pass (the C tiles stay in the accumulators across the
 * whole range): bit-identical to p_end - p_begin single-panel updates in order.  Targets q_begin, q_begin + q_stride,
 * ... < q_end must all lie behind the range (q_begin >= p_end). */
GPRC_API int gprc_dev_update_range(gprc_ctx* ctx, double* packed, int64_t n_pad, int64_t p_begin, int64_t p_end, int64_t q_begin,
                          int64_t q_end, int64_t q_stride);
/* The solves with VECTORS (alpha <- solve(t(L), solve(L, y)), R/GPRclass.R:152, R/GPCclass.R:82-83) work with the explicit inverse
 * of every panel's NB x NB diagonal block: inv, gprc_solve_inv_size(n_pad) doubles, panel p's inverse (transposed, ld = NB) at
 * inv + p NB NB.  gprc_dev_solve_prepare computes it for the factored panels [p_begin, p_end) from packed and winv (one launch);
 * gprc_dev_factor_all does it itself when given inv. */
GPRC_API int64_t gprc_solve_inv_size(int64_t n_pad);
GPRC_API int gprc_dev_solve_prepare(gprc_ctx* ctx, const double* packed, const double* winv, int64_t n_pad, double* inv, int64_t p_begin,
                           int64_t p_end);
/* b := L^-1 b (transpose == 0) or L^-T b (transpose != 0), one launch per panel (launch p: the product of panel p and the
 * diagonal step of the next panel); work: gprc_trsv_work_size(n_pad) doubles */
GPRC_API int64_t gprc_trsv_work_size(int64_t n_pad);
GPRC_API int gprc_dev_trsv(gprc_ctx* ctx, const double* packed, const double* inv, int64_t n_pad, double* b, int transpose,
                  double* work);
/* one panel step of that solve (forward: p = 0, 1, ...; transposed: p = P-1, ..., 0): x_p, then its contribution to the
 * rest of the right-hand side.  The forward solve needs only panels <= p, so a sweep that produces the panels in order can
 * run it beside the factorisation.  Element by element the arithmetic of gprc_dev_trsv: identical bits.  work as above (one
 * buffer for the whole sequence of steps). */
GPRC_API int gprc_dev_trsv_step(gprc_ctx* ctx, const double* packed, const double* inv, int64_t n_pad, double* b, int transpose,
                       int64_t p, double* work);
/* vt (m_pad x n_pad, leading dimension ld >= m_pad, ld even, m_pad % 128 == 0) = K(X_star, X), zero in the
 * padding.  Keep ld off powers of two (e.g. m_pad + 128): a 2^k-byte column stride aliases HBM channels. */
GPRC_API int gprc_dev_fill_cross(gprc_ctx* ctx, int kernel, const double* params_host, int n_params, const double* X_star,
                        int64_t d, int64_t m, int64_t m_pad, const double* X, int64_t n, int64_t n_pad, double* vt,
                        int64_t ld);
/* out[i] = sum_j vt[i + j*ld] * w[j]  (w == NULL: sum_j vt[i,j]^2); work: rows * gprc_rowreduce_splits(cols) */
GPRC_API int64_t gprc_rowreduce_splits(int64_t cols);
GPRC_API int gprc_dev_row_reduce(gprc_ctx* ctx, const double* vt, int64_t ld, int64_t rows, int64_t cols, const double* w,
                        double* out, double* work);
/* out_dev[0] = -0.5 y.alpha - sum(log(diag L)) - n/2 log(2 pi)  (R/GPRclass.R:153) */
GPRC_API int gprc_dev_logp(gprc_ctx* ctx, const double* packed, int64_t n_pad, int64_t n, const double* y, const double* alpha,
                  double* out_dev);
/* Wrap device buffers a driver factored itself (multi-GPU: every rank ends up with the full packed L,
 * winv and alpha) into a model handle so gprc_gpr_predict() can run on this rank's slice of X_star.
 * Nothing is copied or computed; the buffers are BORROWED and must outlive the handle.
 * y and alpha hold n_pad doubles (zero padded). */
GPRC_API int gprc_gpr_model_from_device(gprc_ctx* ctx, int kernel, const double* params, int n_params, const double* X,
                               int64_t d, int64_t n, const double* y, double* packed, double* winv, double* alpha,
                               double noise, double logp, gprc_model** model_out);
/* vt := vt * L^-T  (row i becomes (L^-1 k_i)^T) */
GPRC_API int gprc_dev_solve_rows(gprc_ctx* ctx, const double* packed, const double* winv, int64_t n_pad, double* vt,
                        int64_t ld, int64_t m_pad);

/* ---- multi-GPU from ONE process (SURVEY 8b "Threading", 8e): the form the R `.Call` boundary can use ------------ *
 * The reference's host is a single R process; it cannot be forked per GPU.  A gprc_mgpu drives G ranks -- one per
 * listed device, each with its own streams and buffers -- from the calling thread and runs the SAME sweep as the
 * one-process-per-GPU driver: panel p of the packed factor is owned by rank p mod G (1-D block-cyclic), the owner
 * factors it on a side stream beside the trailing update of the previous panel (look-ahead), every rank receives every
 * panel -- the ONE exchange step -- so L ends up replicated, alpha is solved on every rank, and the test points of a
 * predict are sliced over the ranks with no exchange.  Replaces GPR$initialize / GPR$predict (R/GPRclass.R:127-170)
 * exactly as gprc_gpr_fit / gprc_gpr_predict do, with bit-identical results.
 *   devices  n_ranks device indices.  A device may be listed SEVERAL times: "virtual ranks" that share a GPU and
 *            exchange panels by device-to-device copies -- how a one-GPU box exercises the G = 2, 3, 8 sweeps.
 *   flags    GPRC_MGPU_RCCL: the exchange is ncclBroadcast over xGMI (single-process ncclCommInitAll + group calls;
 *            distinct devices only; librccl.so.1 is resolved at run time).  0: peer copies (hipMemcpyPeerAsync).
 *            GPRC_MGPU_NO_LOOKAHEAD: factor panel p+1 only after the whole trailing update of panel p.
 * All data pointers of these calls are HOST memory (REAL(x) of the `.Call` case). */
#define GPRC_MGPU_RCCL 1
#define GPRC_MGPU_NO_LOOKAHEAD 2
/*            GPRC_MGPU_SCATTER_ALLGATHER: the large-message form of the exchange step (last row).  xGMI is
 *            point-to-point: a rooted broadcast leaves the owner G - 1 times at one link's rate each; here the owner hands
 *            piece r (1/G of the panel, 4 KiB granules) to rank r over all its links at once and every rank then collects the
 *            other pieces from their holders (RCCL: grouped ncclSend / ncclRecv; otherwise event-ordered pull copies).  Panels
 *            below 1 MiB and the 512 KiB of inverses still travel as one rooted broadcast.  Pure data movement: same bits.
 *            GPRC_MGPU_AUTO_EXCHANGE: gprc_mgpu_create runs gprc_mgpu_calibrate and keeps the faster form. */
#define GPRC_MGPU_SCATTER_ALLGATHER 4
#define GPRC_MGPU_AUTO_EXCHANGE 8
typedef struct gprc_mgpu gprc_mgpu;
typedef struct gprc_mgpu_model gprc_mgpu_model;
GPRC_API int gprc_mgpu_create(const int* devices, int n_ranks, int flags, gprc_mgpu** mgpu_out);
GPRC_API int gprc_mgpu_destroy(gprc_mgpu* mgpu);
GPRC_API int gprc_mgpu_ranks(const gprc_mgpu* mgpu, int* n_ranks_out);
/* Start-up calibration of the exchange step: both forms move a `doubles`-sized buffer from a rotating root to every rank
 * `reps` times (after one warm-up), must deliver identical data, and scatter + all-gather is adopted when it is at least 10 %
 * faster.  choice_out (may be NULL): 0 rooted broadcast, 1 scatter + all-gather; ms_out (may be NULL): the two times. */
GPRC_API int gprc_mgpu_calibrate(gprc_mgpu* mgpu, int64_t doubles, int reps, int* choice_out, double* ms_out);
/* 0 peer copies / rooted, 1 RCCL broadcast, 2 peer copies scatter + all-gather, 3 RCCL scatter + all-gather */
GPRC_API int gprc_mgpu_exchange_mode(const gprc_mgpu* mgpu, int* mode_out);
/* What the last fit (and the last predict) on this gprc_mgpu did -- a schedule-rehearsal record, not a benchmark.  out[0..n):
 * [0] ranks G, [1] panels P, [2] exchange mode, [3] exchange operations issued, [4] bytes received per rank, [5] event record /
 * wait pairs, [6] far-update passes, [7] look-ahead updates, [8] fit wall ms, [9] predict wall ms, then per rank r
 * [10 + 3r] fill + sweep ms, [11 + 3r] alpha / logp ms (HIP events on the rank's main stream), [12 + 3r] predict ms of its slice. */
GPRC_API int gprc_mgpu_stats(const gprc_mgpu* mgpu, double* out, int n);
/* GPR$initialize, one Cholesky attempt (as gprc_gpr_fit) / the ten-step jitter loop (as gprc_gpr_fit_retry) */
GPRC_API int gprc_mgpu_gpr_fit(gprc_mgpu* mgpu, int kernel, const double* params, int n_params, const double* X, int64_t d,
                      int64_t n, const double* y, double noise, gprc_mgpu_model** model_out);
GPRC_API int gprc_mgpu_gpr_fit_retry(gprc_mgpu* mgpu, int kernel, const double* params, int n_params, const double* X, int64_t d,
                            int64_t n, const double* y, double noise, gprc_mgpu_model** model_out, double* noise_used,
                            int* attempts);
/* GPR$predict(X_star, pointwise_var = TRUE): rank r predicts the r-th contiguous slice of the test points */
GPRC_API int gprc_mgpu_gpr_predict(gprc_mgpu_model* model, const double* X_star, int64_t n_star, double* mean_out,
                          double* var_out);
GPRC_API int gprc_mgpu_gpr_get_alpha(gprc_mgpu_model* model, double* alpha_out);
GPRC_API int gprc_mgpu_gpr_get_logp(gprc_mgpu_model* model, double* logp_out);
GPRC_API int gprc_mgpu_gpr_get_noise(gprc_mgpu_model* model, double* noise_out);
/* rank r's replica as an ordinary (borrowed) model handle: gprc_gpr_predict(pointwise = 0), gprc_model_get_L, ...
 * It belongs to the gprc_mgpu_model and dies with it.
 * Lifetime: a gprc_mgpu_model may be freed after its gprc_mgpu was destroyed (garbage collectors finalise in any order); its
 * predict then fails with GPRC_ERR_ARG instead of touching the destroyed ranks. */
GPRC_API int gprc_mgpu_model_rank(gprc_mgpu_model* model, int rank, gprc_model** model_out);
GPRC_API int gprc_mgpu_model_free(gprc_mgpu_model* model);

/* ---- sampling: multivariate_normal(n, mean, covariance, tol = 1e-6)  (R/GPRclass.R:360-370) --------------------------
 * L = t(chol(covariance)); if the Cholesky fails (the usual case for a posterior covariance K(X*,X*) - t(v) %*% v,
 * which is numerically rank deficient) L = eigen$vectors %*% diag(sqrt(pmax(eigen$values, 0))) after
 * stopifnot(all(eigen$values > -tol * abs(eigen$values[1]))) -> GPRC_ERR_NOT_PD.  The standard normal matrix Z
 * (m x n_draws, column-major) comes from the caller: R draws it with rnorm on the host and so does the binding, so
 * the generator stays the caller's.  out = drop(mean) + L %*% Z, m x n_draws.  *method_out: 1 Cholesky, 2 eigen.
 * Eigenvectors are defined up to sign (and basis within repeated eigenvalues); with the eigen branch the draws for a
 * given Z therefore differ between LAPACK builds, this library and R, while L %*% t(L) -- the distribution -- agrees. */
GPRC_API int gprc_mvn_factor(gprc_ctx* ctx, const double* cov, int64_t ld, int64_t m, double tol, double* L_out, int* method_out);
GPRC_API int gprc_mvn_sample(gprc_ctx* ctx, const double* cov, int64_t ld, int64_t m, const double* mean, double tol, const double* Z,
                    int64_t n_draws, double* out, int* method_out);
/* eigen(A, symmetric = TRUE): eigenvalues in decreasing order and orthonormal eigenvectors (columns, m x m, may be
 * NULL); only the lower triangle of A is read (LAPACK dsyevr 'L', as R calls it).  Cyclic two-sided Jacobi with a
 * round-robin pair order on the device, m <= 16384; *sweeps_out = sweeps used. */
GPRC_API int gprc_sym_eigen(gprc_ctx* ctx, const double* A, int64_t lda, int64_t m, double* values_out, double* vectors_out,
                   int* sweeps_out);

/* combine_all(lst) (R/simulation.R:338-349), the test grid of the simulate_* harness (:101-102, :223-224): all
 * combinations of the axis values, one point per column (d x prod(lengths), column-major), the LAST axis varying
 * fastest.  axis_values: the axes concatenated (sum(lengths) doubles); lengths: d host integers; d <= 64. */
GPRC_API int gprc_combine_all(gprc_ctx* ctx, const double* axis_values, const int64_t* lengths, int d, double* out);
/* ---- measurement: per-kernel-kind HIP-event timing (bench.py's live roofline numbers) ------------ *
 * When enabled, every launch is bracketed by two hipEvents on the stream it is launched on.  Kinds:
 * 0 fill, 1 potf2_inv, 2 trsm_panel, 3 in-panel GEMM (K=128), 4 trailing update, 5 predict right
 * update (K=512), 6 trsv, 7 row reductions, 8 covariance SYRK, 9 derivative row sums, 10 Jacobi sweep, 11 predict
 * left-looking update, 12 trailing left-looking update, 13 fused panel factorisation, 14 fused in-panel solve of the predict.  flops/bytes are the ALGORITHMIC
 * figures of DESIGN.md for the launches seen, not counter readings. */
GPRC_API int gprc_prof_enable(int on);
/* With the environment variable GPRC_PANEL_TRACE=<p> set, the factor role of the fused panel kernel of panel p leaves
 * s_memrealtime stamps (100 MHz ticks) of its stages: [0] start, then per 128-column sub-step j: [1+6j] diagonal block
 * factored, [2+6j] W_j published, [3+6j] E_{j+1} seen, [4+6j] L(j+1,j) solved, [5+6j] R_{j+1} published, [6+6j] block
 * (j+1,j+1) updated.  side != 0: the context's look-ahead stream's launches.  Measurement only. */
GPRC_API int gprc_prof_panel_trace(gprc_ctx* ctx, int side, int64_t* ticks_out, int n);
/* With GPRC_SERVICE_TRACE set, a factor-service sweep (gprc_dev_factor_all and every fit at n <= 24576) leaves 16 stamps per panel
 * p (100 MHz ticks; 0 = not reached): [14] factor role arrives, [0] its diagonal block is complete: chain starts, [1] chain done;
 * [2]/[3] look-ahead strips start / done; [4] next-diagonal-block tiles may start, [5] their last k-chunk starts, [6] block
 * complete; [7]/[8] first ordinary-strip workgroup starts / ends; trailing update: [9] first tile starts, [10] sees the
 * look-ahead rows, [11] has published, [12] first tile of the block after the next published, [13] last tile done.
 * panels <= 48.  Measurement only. */
GPRC_API int gprc_prof_service_trace(gprc_ctx* ctx, int64_t* ticks_out, int panels);
/* When a factorisation ends in a device-side wait timeout (info = -99: the factor service's persistent launch and the caller's kernels did
 * not run concurrently, or a dependency never arrived), every wait that was unsatisfied at that moment has left a record: out[0] = records
 * written, then 8 ints per record from out[8] on -- site (1 flag, 2 count, 3 field, 4 chain helper, 5 sweep kernel, 6 residency gate; +10: the
 * wait was a bystander that left because somebody else's bound had run out), workgroup, grid size, value needed, value seen, index of the
 * awaited word inside its panel's flags, threads per workgroup, low 32 bits of those flags' address.  ints: capacity of out (at most
 * 8 * 49 are written).  Reading clears the records.  The fit entry points put the same records, as text, into gprc_last_error.  Diagnostics only. */
GPRC_API int gprc_prof_wait_timeout(int* out, int ints);
GPRC_API int gprc_prof_reset(void);
GPRC_API int gprc_prof_kinds(void);
GPRC_API int gprc_prof_summary(int kind, int64_t* count_out, double* ms_out, double* flops_out, double* bytes_out);

#ifdef __cplusplus
}
#endif
#endif /* GPRC_NATIVE_H */
