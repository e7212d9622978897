// gprc_internal.h -- shared declarations of the gfx950 implementation behind include/gprc_native.h.
// Host-side launchers live next to their kernels; the C ABI (gprc_api.hip) only composes them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/gprc_native.h"

namespace gprc {

// ---- geometry (DESIGN.md "Data layout") -------------------------------------------------------
#ifndef GPRC_NB
#define GPRC_NB 512
#endif
constexpr int NB = GPRC_NB;  // outer panel width: K-depth of the trailing update (NB/8 flop per byte of C traffic)
constexpr int NBI = 128;  // inner block: one LDS-resident diagonal factorisation, one GEMM tile edge
constexpr int TPP = NB / NBI;  // 128-wide tile columns per panel
constexpr int MAX_PARAMS = 256;  // kernel parameters travel as kernel arguments; only linear's per-coordinate sigma needs more than 2
constexpr int MAX_DEVICES = 64;  // per-device one-time setup flags
// written to the LAPACK-info word by a fused kernel whose device-side dependency wait ran out (never seen in practice:
// it takes a faulted or never-scheduled producer); hosts turn it into GPRC_ERR_HIP instead of using the factor
// GPRC_INFO_WAIT_TIMEOUT (-99, include/gprc_native.h): what info becomes when a device-side dependency wait runs out
static_assert(NB % NBI == 0 && NB >= NBI, "panel width must be a multiple of the 128 block");

__host__ __device__ static inline int64_t pad_up(int64_t n, int64_t m) { return (n + m - 1) / m * m; }
__host__ __device__ static inline int64_t panel_offset(int64_t n_pad, int64_t p) { return (int64_t)NB * (p * n_pad - (int64_t)NB * p * (p - 1) / 2); }
__host__ __device__ static inline int64_t panel_ld(int64_t n_pad, int64_t p) { return n_pad - p * NB; }

// ---- error plumbing ---------------------------------------------------------------------------
void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define GPRC_HIP(call)                                                          \
  do {                                                                          \
    hipError_t e__ = (call);                                                    \
    if (e__ != hipSuccess) return ::gprc::hip_fail(e__, #call, __FILE__, __LINE__); \
  } while (0)
#define GPRC_TRY(call)          \
  do {                          \
    int rc__ = (call);          \
    if (rc__ != 0) return rc__; \
  } while (0)
#define GPRC_LAUNCH_CHECK() GPRC_HIP(hipGetLastError())

// ---- in-library event profiler (bench.py's live roofline numbers) ---------------------------------
enum ProfKind { PK_FILL = 0, PK_POTF2 = 1, PK_TRSM_PANEL = 2, PK_GEMM_INNER = 3, PK_TRAILING = 4, PK_SOLVE_UPDATE = 5,
                PK_TRSV = 6, PK_ROWREDUCE = 7, PK_COV_SYRK = 8, PK_DERIV = 9, PK_JACOBI = 10, PK_SOLVE_LEFT = 11, PK_TRAILING_LEFT = 12, PK_PANEL_FUSED = 13, PK_SOLVE_PANEL = 14, PK_COUNT = 15 };
bool prof_enabled();
void prof_begin(hipStream_t s, int kind);
void prof_end(hipStream_t s, int kind, double flops, double bytes);
struct ProfScope {  // brackets one launch (or one launch sequence) with a pair of HIP events when profiling is on
  hipStream_t s; int kind; double flops, bytes; bool on;
  ProfScope(hipStream_t s_, int kind_, double flops_, double bytes_) : s(s_), kind(kind_), flops(flops_), bytes(bytes_), on(prof_enabled()) {
    if (on) prof_begin(s, kind);
  }
  ~ProfScope() { if (on) prof_end(s, kind, flops, bytes); }
};

struct KernelSpec {
  int id;
  int n_params;
  double p[MAX_PARAMS];
};

// how the out-of-range part of a fill is written
enum PadMode { PAD_NONE = 0,      // exact extents, bounds-checked stores (user-facing covariance_matrix)
               PAD_IDENTITY = 1,  // symmetric K + noise*I, identity outside n (the factor's padding)
               PAD_ZERO = 2 };    // zero outside the valid extents (cross-covariance chunks)

// ---- launchers (kernels_fill.hip) --------------------------------------------------------------
// out[(i-row0) + (j-col0)*ld] = k(A_i, B_j) for i in [row0,row0+nrows), j in [col0,col0+ncols)
int launch_fill(hipStream_t s, const KernelSpec& ks, const double* A, int64_t nA, const double* B, int64_t nB, int64_t d,
                double* out, int64_t ld, int64_t row0, int64_t nrows, int64_t col0, int64_t ncols, PadMode mode,
                double noise);
int launch_colwise(hipStream_t s, const KernelSpec& ks, const double* x, const double* y, int64_t d, int64_t m, double* out);
// predict chunk K(X_star, X) with the fused epilogue: per-column-tile partials of K*^T w and an optional column scale
int64_t fill_mean_tiles(int64_t cols);
int launch_fill_cross_fused(hipStream_t s, const KernelSpec& ks, const double* Xs, int64_t m, const double* X, int64_t n, int64_t d,
                            double* vt, int64_t ld, int64_t m_pad, int64_t n_pad, const double* w, double* mpart, const double* colscale);
// S[r + n*i] = sum_c deriv_i(X[,r], X[,c]; v): row sums of the parameter derivatives of K(X,X) as cov_dict$...$deriv
// (R/fit.R:4-31) defines them; n_deriv (1 or 2) components.  Kernels: sqrexp, gammaexp, polynomial, rationalquadratic.
int launch_deriv_rowsum(hipStream_t s, int kernel, double v0, double v1, const double* X, int64_t d, int64_t n, double* S);
int launch_set_identity_rows(hipStream_t s, double* vt, int64_t ld, int64_t rows, int64_t cols, int64_t row0);  // vt[i,j] = (row0+i == j)

// ---- launchers (kernels_chol.hip) --------------------------------------------------------------
// factor the 128x128 diagonal block at A (ld) in LDS, write L in place and its inverse to winv
int launch_potf2_inv(hipStream_t s, double* A, int64_t lda, double* winv, int* info_dev, int col0);
// the whole of panel p (four diagonal blocks, panel solves, in-panel updates) in ONE launch; sync16: 64 bytes of device
// memory the launch may use for its flags (zeroed by the launcher, stream-ordered: one buffer serves a whole stream)
int launch_panel_fused(hipStream_t s, double* packed, int64_t n_pad, int64_t p, double* winv, int* info_dev, void* sync16);
// the predict's in-panel solve of panel p in one launch (see solve_panel_fused_kernel)
int launch_solve_panel_fused(hipStream_t s, double* vt, int64_t ldv, int64_t m_pad, const double* packed, int64_t n_pad, int64_t p,
                             const double* winv, double* sspart, int64_t ss_stride = 0);   // ss_stride: rows per block of sspart (0: m_pad)
// X[M x 128] := X * W^T for lower-triangular 128x128 W (= inverse of a diagonal block of L)
int launch_trsm_panel(hipStream_t s, double* X, int64_t ldx, int64_t M, const double* winv, double* ssq = nullptr);
// C[M x N] -= A[M x K] * B[N x K]^T; lower_diag >= 0: row tile r / col tile c with r + lower_diag < c is skipped
int launch_gemm_nt(hipStream_t s, double* C, int64_t ldc, const double* A, int64_t lda, const double* B, int64_t ldb,
                   int64_t M, int64_t N, int64_t K, int lower, int kind);
// trailing update of packed panels q_begin, q_begin+q_stride, ... < q_end with factored panel p
// left-looking predict-solve step: vt[:, j NB:(j+G) NB] -= vt[:, 0:j NB] * L[j NB:(j+G) NB, 0:j NB]^T   (L packed)
// tri_row0 >= 0: the rows of vt are rows tri_row0.. of the identity (zero left of their own column): per tile the pass starts there
int launch_solve_left(hipStream_t s, double* vt, int64_t ldv, int64_t m_pad, const double* packed, int64_t n_pad, int64_t j, int64_t G,
                      int64_t tri_row0 = -1);
int launch_trailing_update(hipStream_t s, double* packed, int64_t n_pad, int64_t p, int64_t q_begin, int64_t q_end,
                           int64_t q_stride);
// factor service (one-GPU right-looking sweep): the dependent chain of all panels in one persistent launch (side stream) + per panel
// the ordinary strips and the trailing update without the next diagonal block (caller's stream)
size_t panel_service_sync_bytes(int64_t P);
int launch_panel_service(hipStream_t s, double* packed, int64_t n_pad, double* winv, int* info_dev, void* sync, void* trace, double* inv,
                         int64_t p_begin, int64_t p_end, int part = 0);
bool service_shared(int64_t n_pad);   // the service's 4-wave roles share their CUs with one sweep workgroup each (two launches: parts 1 and 2)
// inv (n_pad x NB doubles): per panel the explicit inverse of its NB x NB diagonal block, transposed -- what the vector solves use
int launch_inv512(hipStream_t s, const double* packed, int64_t n_pad, const double* winv, double* inv, int64_t p_begin, int64_t p_end);
int launch_service_gate(hipStream_t s, int64_t n_pad, int* info_dev, void* sync, int launches);
int launch_panel_strips(hipStream_t s, double* packed, int64_t n_pad, int64_t p, double* winv, int* info_dev, void* sync, void* trace);
int launch_trailing_service(hipStream_t s, double* packed, int64_t n_pad, int64_t p, double* winv, int* info_dev, void* sync, void* trace, int64_t q_end);
int launch_trailing_sweep(hipStream_t s, double* packed, int64_t n_pad, int64_t g0, int64_t g1, double* winv, int* info_dev, void* sync, void* trace,
                          int service_wgs);
int service_workgroups(bool with_inverse, int64_t n_pad);
std::string wait_timeout_report();   // who gave up first in the last timed-out factorisation (kernels_chol.hip), "" if nobody

// ---- launchers (kernels_vec.hip) ---------------------------------------------------------------
int launch_trailing_left(hipStream_t s, double* packed, int64_t n_pad, int64_t q_begin, int64_t q_end);
int launch_trailing_range(hipStream_t s, double* packed, int64_t n_pad, int64_t p_begin, int64_t p_end, int64_t q_begin, int64_t q_end,
                          int64_t q_stride);
// vector solves: inv = the explicit diagonal-block inverses (launch_inv512 / the factor service), work = gprc_trsv_work_size doubles
int launch_trsv_step(hipStream_t s, const double* packed, const double* inv, int64_t n_pad, double* b, int transpose, int p, double* work);
int launch_trsv(hipStream_t s, const double* packed, const double* inv, int64_t n_pad, double* b, int transpose, double* work);
int64_t rowreduce_splits(int64_t cols);
int launch_row_reduce(hipStream_t s, const double* vt, int64_t ld, int64_t rows, int64_t cols, const double* w, double* out,
                      double* work);
// out[0] = -0.5*y.alpha - sum(log(diag L)) - n/2 log(2 pi)   (R/GPRclass.R:153); n valid entries
int launch_logp(hipStream_t s, const double* packed, int64_t n_pad, int64_t n, const double* y, const double* alpha, double* out);
// unpack the factor into a dense n x n lower matrix (upper = 0)
int launch_unpack_L(hipStream_t s, const double* packed, int64_t n_pad, int64_t n, double* out, int64_t ld_out);
// out[i] = (minuend ? minuend[i] : 0) -/+ sum_{t < nparts} part[t * stride + i], summed in t order: the tail of the fused
// predict epilogues (mean = sum of the fill's partials; var = k(x*,x*) - sum of the solve's per-block sums of squares)
int launch_sum_partials(hipStream_t s, const double* part, int64_t nparts, int64_t stride, int64_t rows, const double* minuend, double* out);
// GPC vector stages (R/GPCclass.R:78-86, 99-103, 110-114)
int launch_gpc_pre(hipStream_t s, const double* f, const double* y, int64_t n, double* sw, double* b);
int launch_gpc_scale(hipStream_t s, const double* sw, const double* v, double* out, int64_t n);                   // out = sw*v
int launch_gpc_a(hipStream_t s, const double* b, const double* sw, const double* t, double* a, int64_t n);        // a = b - sw*t
int launch_gpc_objective(hipStream_t s, const double* a, const double* f, const double* y, int64_t n, double* out);
int launch_gpc_build_B(hipStream_t s, const double* Kfull, int64_t n_pad, const double* sw, double* packed);
int launch_gpc_grad(hipStream_t s, const double* f, const double* y, int64_t n, double* g, double* sw);           // g=(y+1)/2-P
int launch_gpc_class_prob(hipStream_t s, const double* fs, const double* vf, double* out, int64_t n);              // R/GPCclass.R:116-117
// sampling support (kernels_eig.hip)
int launch_pack_dense(hipStream_t s, const double* A, int64_t lda, int64_t m, int64_t n_pad, double* packed);
int launch_sym_copy(hipStream_t s, const double* A, int64_t lda, int64_t m, double* W, double* V);
int launch_jacobi_sweep(hipStream_t s, double* W, double* V, int m, double* cs);
int launch_jacobi_offnorm(hipStream_t s, const double* W, int m, double* off, double* dg);
int launch_gather_scale_cols(hipStream_t s, const double* V, int m, const int* perm, const double* scale, double* out, int64_t ldo);
int launch_affine_lz(hipStream_t s, const double* L, int64_t ldl, int64_t m, const double* mean, const double* Z, int64_t ldz,
                     int64_t ndraws, double* out, int64_t ldo, int lower);
int launch_combine_all(hipStream_t s, const double* vals_dev, const int64_t* lengths, int d, double* out);
int launch_diag_sum(hipStream_t s, const double* packed, int64_t n_pad, int64_t n, double* out);                  // sum(diag(L))

}  // namespace gprc
