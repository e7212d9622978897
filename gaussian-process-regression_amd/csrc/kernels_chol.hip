// kernels_chol.hip -- blocked right-looking fp64 Cholesky pieces for gfx950 (MI355X).
//
// Replaces L <- t(chol(K + noise * diag(n)))  (reference R/GPRclass.R:142, LAPACK dpotrf) and, through
// the same GEMM tile, v <- solve(L, K_star) (R/GPRclass.R:162, which the reference runs as a general
// pivoted dgesv).  Three kernels:
//   potf2_inv_kernel  one workgroup factors a 128x128 diagonal block entirely in LDS and also forms
//                     its inverse (so every panel / right-hand-side solve below is a GEMM);
//   gemm tile core    128x128 output tile per 256-thread workgroup, 4 waves x (64x64) of
//                     v_mfma_f64_16x16x4_f64, A/B strips staged through double-buffered LDS;
//   wrappers          panel solve (X := X * Winv^T, in place), in-panel / general C -= A*B^T, and the
//                     trailing update over the packed block-column layout (lower tiles only).
// The trailing update is the dominant kernel of the whole path: n^3/3 of the fit and n^2 n* of the
// predict go through gemm_tile_128().
#include "gprc_internal.h"

namespace gprc {

typedef double double4_t __attribute__((ext_vector_type(4)));

namespace {

// ------------------------------------------------------------------------------------------------
// Diagonal block: Cholesky + inverse in one LDS-resident sweep.
//
// S is a 128 x 129 column-major LDS array.  At step j, column j of S holds, by row c:
//   c > j : L[c][j]                 (the Cholesky column, final after the scaling)
//   c < j : X[j][c] = (L^-1)[j][c]  (row j of the inverse, stored transposed in the upper triangle)
// so a single multiplier m_c = S[c + j*LD] (c != j; Dinv[j] for c == j) drives both the Cholesky
// rank-1 update (targets L[i][c], c > j) and the elimination that builds the inverse
// (targets X[i][c], c <= j) of every row i > j.  The odd leading dimension keeps both the
// column-walking and the row-walking accesses free of bank conflicts.
// ------------------------------------------------------------------------------------------------
constexpr int PB = 128;
constexpr int PLD = 129;

__global__ __launch_bounds__(1024) void potf2_inv_kernel(double* A, int64_t lda, double* winv, int* info, int col0) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* S = sm;                 // PB * PLD
  double* Dinv = sm + PB * PLD;   // PB
  const int t = threadIdx.x;
  const int i = t & 127, ty = t >> 7;  // ty in 0..7
  for (int c = ty; c < PB; c += 8) S[i + c * PLD] = (i >= c) ? A[i + (int64_t)c * lda] : 0.0;
  if (t < PB) Dinv[t] = 1.0;
  __syncthreads();

  for (int j = 0; j < PB; ++j) {
    const double d = S[j + j * PLD];
    const bool bad = !(d > 0.0);
    if (bad && t == 0) atomicCAS(info, 0, col0 + j + 1);  // LAPACK info: first non-PD leading minor
    const double ljj = sqrt(d);
    __syncthreads();  // everyone has read the pivot before it is overwritten
    if (t < PB) {
      if (t == j) { S[j + j * PLD] = ljj; Dinv[j] = 1.0 / ljj; }
      else S[t + j * PLD] = S[t + j * PLD] / ljj;  // L[c][j] (c>j) and X[j][c] (c<j) alike
    }
    __syncthreads();
    if (i > j) {
      const double lij = S[i + j * PLD];
      for (int c = ty; c <= i; c += 8) {
        if (c > j) {
          S[i + c * PLD] = fma(-lij, S[c + j * PLD], S[i + c * PLD]);
        } else {
          const double m = (c == j) ? Dinv[j] : S[c + j * PLD];
          S[c + i * PLD] = fma(-lij, m, S[c + i * PLD]);
        }
      }
    }
    __syncthreads();
  }
  // write L (lower, in place) and Winv = L^-1 (dense 128x128, upper zero)
  for (int c = ty; c < PB; c += 8) {
    if (i >= c) A[i + (int64_t)c * lda] = S[i + c * PLD];
    double w = 0.0;
    if (i == c) w = Dinv[i];
    else if (i > c) w = S[c + i * PLD];
    winv[i + c * PB] = w;
  }
}

// ------------------------------------------------------------------------------------------------
// GEMM tile core: acc(128x128) = A(128 x K) * B(128 x K)^T, A and B column-major strips.
//
// LDS image per operand and buffer: [KB=16][LDT=144] doubles, i.e. one k-slice of the strip per row
// of 128 contiguous matrix rows + 16 pad.  The pad moves consecutive k-slices by 32 banks, so the
// ds_read_b64 of an MFMA operand (16 matrix rows x 2 k per 32-lane half) is conflict-free, and the
// 16-byte staging stores of one wave cover 1 KiB contiguous.
// MFMA operands are swapped (A-operand <- B strip, B-operand <- A strip): the accumulator then holds
// C[row = 16m + (lane&15)][col = 16n + (lane>>4) + 4r], i.e. 16 consecutive ROWS per lane group,
// which is the contiguous direction of the column-major C tile.
// ------------------------------------------------------------------------------------------------
constexpr int G_KB = 16;
constexpr int G_LDT = 144;
constexpr int G_BUF = G_KB * G_LDT;           // doubles per operand per buffer
constexpr int G_SMEM_DOUBLES = 4 * G_BUF;     // A,B x 2 buffers = 73,728 B -> 2 workgroups per CU

template <bool SET>
__device__ __forceinline__ void gemm_tile_128(double* C, int64_t ldc, const double* A, int64_t lda, const double* B,
                                              int64_t ldb, int K, double* smem) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int fk = lane >> 4, fr = lane & 15;
  double* As = smem;
  double* Bs = smem + 2 * G_BUF;

  double4_t acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (double4_t){0.0, 0.0, 0.0, 0.0};

  // staging: wave w moves k-slices w, w+4, w+8, w+12; lane l moves rows 2l, 2l+1 (16 B)
  const double* Ag = A + 2 * lane + (int64_t)wave * lda;
  const double* Bg = B + 2 * lane + (int64_t)wave * ldb;
  const int soff = wave * G_LDT + 2 * lane;
  double2 ra[4], rb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ra[i] = *reinterpret_cast<const double2*>(Ag + (int64_t)(4 * i) * lda);
    rb[i] = *reinterpret_cast<const double2*>(Bg + (int64_t)(4 * i) * ldb);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<double2*>(As + soff + 4 * i * G_LDT) = ra[i];
    *reinterpret_cast<double2*>(Bs + soff + 4 * i * G_LDT) = rb[i];
  }
  __syncthreads();

  const int KT = K / G_KB;
  int cur = 0;
  for (int kt = 0; kt < KT; ++kt) {
    const bool more = (kt + 1 < KT);
    if (more) {
      Ag += (int64_t)G_KB * lda;
      Bg += (int64_t)G_KB * ldb;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const double2*>(Ag + (int64_t)(4 * i) * lda);
        rb[i] = *reinterpret_cast<const double2*>(Bg + (int64_t)(4 * i) * ldb);
      }
    }
    const double* Ac = As + cur * G_BUF + wr * 64 + fr + fk * G_LDT;
    const double* Bc = Bs + cur * G_BUF + wc * 64 + fr + fk * G_LDT;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double a[4], b[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = Ac[kk * 4 * G_LDT + m * 16];
#pragma unroll
      for (int n = 0; n < 4; ++n) b[n] = Bc[kk * 4 * G_LDT + n * 16];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[n], a[m], acc[m][n], 0, 0, 0);
    }
    if (more) {
      const int nb = (cur ^ 1) * G_BUF;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<double2*>(As + nb + soff + 4 * i * G_LDT) = ra[i];
        *reinterpret_cast<double2*>(Bs + nb + soff + 4 * i * G_LDT) = rb[i];
      }
    }
    __syncthreads();
    cur ^= 1;
  }

  double* Cw = C + (wr * 64 + fr) + (int64_t)(wc * 64 + fk) * ldc;
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        double* p = Cw + m * 16 + (int64_t)(n * 16 + 4 * r) * ldc;
        if (SET) *p = acc[m][n][r];
        else *p = *p - acc[m][n][r];
      }
}

// blockIdx -> logical id so that each XCD (blocks b, b+8, ... share one) owns a contiguous id range
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// C[M x N] -= A * B^T, 1-D grid of (M/128)*(N/128) tiles visited in 8-row groups
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(double* C, int64_t ldc, const double* A, int64_t lda,
                                                         const double* B, int64_t ldb, int tiles_m, int tiles_n, int K,
                                                         int lower) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const unsigned id = xcd_remap(blockIdx.x, gridDim.x);
  const int width = 8 * tiles_n;
  const int g = id / width, first_m = g * 8;
  const int gsize = (tiles_m - first_m < 8) ? (tiles_m - first_m) : 8;
  const int tr = first_m + (int)(id % width) % gsize;
  const int tc = (int)(id % width) / gsize;
  if (lower && tc > tr) return;
  gemm_tile_128<false>(C + (int64_t)tr * 128 + (int64_t)tc * 128 * ldc, ldc, A + (int64_t)tr * 128, lda,
                       B + (int64_t)tc * 128, ldb, K, smem);
}

// X[M x 128] := X * W^T (W = inverse of the diagonal block, lower triangular), in place: a workgroup
// owns a full 128-row strip, and every load of it precedes the epilogue stores.
__global__ __launch_bounds__(256, 2) void trsm_panel_kernel(double* X, int64_t ldx, const double* winv) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Xs = X + (int64_t)blockIdx.x * 128;
  gemm_tile_128<true>(Xs, ldx, Xs, ldx, winv, 128, 128, smem);
}

// Trailing update over the packed layout: for every target panel q in {q_begin, q_begin+stride, ..}
// C_q -= L_p[rows of q] * L_p[rows of q's diagonal block]^T, lower tiles only.
__global__ __launch_bounds__(256, 2) void trailing_kernel(double* packed, int64_t n_pad, int p, int q_begin, int q_stride,
                                                          int n_targets) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int P = (int)(n_pad / NB);
  int id = (int)xcd_remap(blockIdx.x, gridDim.x);
  // locate the target panel: panel q holds 16*(P-q) - 6 lower tiles
  int q = q_begin, s = 0;
  for (; s < n_targets; ++s, q += q_stride) {
    const int tq = 16 * (P - q) - 6;
    if (id < tq) break;
    id -= tq;
  }
  if (s >= n_targets) return;
  int tr, tc;
  if (id < 10) {  // the diagonal 512x512 block: lower tiles (0,0) (1,0) (1,1) (2,0) ...
    tr = (id >= 6) ? 3 : (id >= 3) ? 2 : (id >= 1) ? 1 : 0;
    tc = id - tr * (tr + 1) / 2;
  } else {
    tr = 4 + ((id - 10) >> 2);
    tc = (id - 10) & 3;
  }
  const int64_t ldp = panel_ld(n_pad, p), ldq = panel_ld(n_pad, q);
  const double* Lp = packed + panel_offset(n_pad, p) + (int64_t)(q - p) * NB;  // row q*NB of panel p
  double* Cq = packed + panel_offset(n_pad, q);
  gemm_tile_128<false>(Cq + (int64_t)tr * 128 + (int64_t)tc * 128 * ldq, ldq, Lp + (int64_t)tr * 128, ldp,
                       Lp + (int64_t)tc * 128, ldp, NB, smem);
}

}  // namespace

int launch_potf2_inv(hipStream_t s, double* A, int64_t lda, double* winv, int* info_dev, int col0) {
  const size_t smem = (size_t)(PB * PLD + PB) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(potf2_inv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_set = true;
  }
  hipLaunchKernelGGL(potf2_inv_kernel, dim3(1), dim3(1024), smem, s, A, lda, winv, info_dev, col0);
  GPRC_LAUNCH_CHECK();
  return 0;
}

static int ensure_gemm_attrs() {
  static bool done = false;
  if (done) return 0;
  const int smem = (int)(G_SMEM_DOUBLES * sizeof(double));
  GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trsm_panel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trailing_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  done = true;
  return 0;
}

int launch_trsm_panel(hipStream_t s, double* X, int64_t ldx, int64_t M, const double* winv) {
  if (M <= 0) return 0;
  if (M % 128) { set_error("trsm_panel: M must be a multiple of 128"); return GPRC_ERR_ARG; }
  GPRC_TRY(ensure_gemm_attrs());
  hipLaunchKernelGGL(trsm_panel_kernel, dim3((unsigned)(M / 128)), dim3(256), G_SMEM_DOUBLES * sizeof(double), s, X, ldx, winv);
  GPRC_LAUNCH_CHECK();
  return 0;
}

int launch_gemm_nt(hipStream_t s, double* C, int64_t ldc, const double* A, int64_t lda, const double* B, int64_t ldb,
                   int64_t M, int64_t N, int64_t K, int lower) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (M % 128 || N % 128 || K % G_KB || (lda & 1) || (ldb & 1)) { set_error("gemm_nt: bad shape"); return GPRC_ERR_ARG; }
  GPRC_TRY(ensure_gemm_attrs());
  const int64_t tiles = (M / 128) * (N / 128);
  if (tiles > 0x7fffffff) { set_error("gemm_nt: too many tiles"); return GPRC_ERR_ARG; }
  hipLaunchKernelGGL(gemm_nt_kernel, dim3((unsigned)tiles), dim3(256), G_SMEM_DOUBLES * sizeof(double), s, C, ldc, A, lda, B,
                     ldb, (int)(M / 128), (int)(N / 128), (int)K, lower);
  GPRC_LAUNCH_CHECK();
  return 0;
}

int launch_trailing_update(hipStream_t s, double* packed, int64_t n_pad, int64_t p, int64_t q_begin, int64_t q_end,
                           int64_t q_stride) {
  const int64_t P = n_pad / NB;
  if (q_begin <= p || q_stride <= 0) { set_error("trailing_update: bad panel range"); return GPRC_ERR_ARG; }
  if (q_end > P) q_end = P;
  int64_t tiles = 0, nt = 0;
  for (int64_t q = q_begin; q < q_end; q += q_stride) { tiles += 16 * (P - q) - 6; ++nt; }
  if (tiles == 0) return 0;
  GPRC_TRY(ensure_gemm_attrs());
  hipLaunchKernelGGL(trailing_kernel, dim3((unsigned)tiles), dim3(256), G_SMEM_DOUBLES * sizeof(double), s, packed, n_pad, (int)p,
                     (int)q_begin, (int)q_stride, (int)nt);
  GPRC_LAUNCH_CHECK();
  return 0;
}

}  // namespace gprc
