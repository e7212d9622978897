// Sampling support: multivariate_normal() of R/GPRclass.R:360-370 needs t(chol(covariance)) or,
// when the Cholesky fails -- the usual case for a posterior covariance K(X*,X*) - t(v) %*% v, which is numerically
// rank deficient -- eigen(covariance, symmetric = TRUE).  This file holds
//   * pack_dense: a dense symmetric matrix (lower triangle read) -> the packed block-column layout of the Cholesky;
//   * a two-sided cyclic Jacobi eigensolver with round-robin ordering: every round rotates m/2 DISJOINT index pairs,
//     so all rotations of a round commute and are applied by two bandwidth-bound kernels (columns of A and V, then
//     rows of A); ~6 m^2 doubles of traffic per round, m-1 rounds per sweep, quadratic convergence;
//   * out = mean + L %*% Z for a few draws (one pass over L).
// All HBM-bound elementwise / GEMV-shaped work: coalesced along the contiguous (row) index, no MFMA.
#include <hip/hip_runtime.h>

#include "gprc_internal.h"

namespace gprc {
namespace {

// dense (lower triangle, mirrored) -> packed panels with identity padding
__global__ __launch_bounds__(256) void pack_dense_kernel(const double* A, int64_t lda, int64_t m, int64_t n_pad, double* packed) {
  for (int64_t j = blockIdx.y; j < n_pad; j += gridDim.y) {
    const int64_t p = j / NB;
    const int64_t i = p * NB + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pad) continue;
    double v;
    if (i < m && j < m) v = (i >= j) ? A[i + j * lda] : A[j + i * lda];
    else v = (i == j) ? 1.0 : 0.0;
    packed[panel_offset(n_pad, p) + (i - p * NB) + (j - p * NB) * panel_ld(n_pad, p)] = v;
  }
}

// W = symmetric copy of A's lower triangle (what eigen(symmetric = TRUE) / dsyevr('L') reads); V = I
__global__ __launch_bounds__(256) void sym_copy_kernel(const double* A, int64_t lda, int64_t m, double* W, double* V) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  for (int64_t j = blockIdx.y; j < m; j += gridDim.y) {
    W[i + j * m] = (i >= j) ? A[i + j * lda] : A[j + i * lda];
    V[i + j * m] = (i == j) ? 1.0 : 0.0;
  }
}

// round-robin ("circle") schedule on mm = m rounded up to even players: round r in [0, mm-1), pair t in [0, mm/2)
__device__ __forceinline__ bool jacobi_pair(int m, int mm, int r, int t, int& p, int& q) {
  int a, b;
  if (t == 0) { a = mm - 1; b = r; }
  else { a = (r + t) % (mm - 1); b = (r - t + (mm - 1)) % (mm - 1); }
  p = a < b ? a : b;
  q = a < b ? b : a;
  return q < m;  // q == m is the dummy player of an odd m
}

// rotation (c, s) that zeroes W[p,q] in J^T W J, J = [[c, s], [-s, c]] in the (p, q) plane (Golub & Van Loan 8.4)
__global__ __launch_bounds__(256) void jacobi_angles_kernel(const double* W, int m, int mm, int r, double* cs) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= mm / 2) return;
  int p, q;
  double c = 1.0, s = 0.0;
  if (jacobi_pair(m, mm, r, t, p, q)) {
    const double apq = W[p + (int64_t)q * m];
    if (apq != 0.0) {
      const double tau = (W[q + (int64_t)q * m] - W[p + (int64_t)p * m]) / (2.0 * apq);
      const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
      c = 1.0 / sqrt(1.0 + tt * tt);
      s = tt * c;
    }
  }
  cs[2 * t] = c;
  cs[2 * t + 1] = s;
}

// W <- W J and V <- V J: thread = row i (contiguous), blockIdx.y strides over the pairs
__global__ __launch_bounds__(256) void jacobi_cols_kernel(double* W, double* V, int m, int mm, int r, const double* cs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  for (int t = blockIdx.y; t < mm / 2; t += gridDim.y) {
    int p, q;
    if (!jacobi_pair(m, mm, r, t, p, q)) continue;
    const double c = cs[2 * t], s = cs[2 * t + 1];
    if (s == 0.0) continue;
    double* wp = W + (int64_t)p * m; double* wq = W + (int64_t)q * m;
    const double a = wp[i], b = wq[i];
    wp[i] = c * a - s * b;
    wq[i] = s * a + c * b;
    double* vp = V + (int64_t)p * m; double* vq = V + (int64_t)q * m;
    const double e = vp[i], f = vq[i];
    vp[i] = c * e - s * f;
    vq[i] = s * e + c * f;
  }
}

// W <- J^T W: thread = pair t, blockIdx.y strides over the columns; the pairs of a round cover every row once, so a
// wavefront still consumes whole cache lines of column j
__global__ __launch_bounds__(256) void jacobi_rows_kernel(double* W, int m, int mm, int r, const double* cs) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= mm / 2) return;
  int p, q;
  if (!jacobi_pair(m, mm, r, t, p, q)) return;
  const double c = cs[2 * t], s = cs[2 * t + 1];
  if (s == 0.0) return;
  for (int j = blockIdx.y; j < m; j += gridDim.y) {
    double* col = W + (int64_t)j * m;
    const double a = col[p], b = col[q];
    col[p] = c * a - s * b;
    col[q] = s * a + c * b;
  }
}

// per column j: off[j] = sum_{i != j} W[i,j]^2, dg[j] = W[j,j]   (summed on the host in index order: deterministic)
__global__ __launch_bounds__(256) void jacobi_offnorm_kernel(const double* W, int m, double* off, double* dg) {
  const int j = blockIdx.x;
  const double* col = W + (int64_t)j * m;
  double a = 0.0;
  for (int i = threadIdx.x; i < m; i += 256)
    if (i != j) a += col[i] * col[i];
  __shared__ double red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) { off[j] = (red[0] + red[1]) + (red[2] + red[3]); dg[j] = col[j]; }
}

// out[:, k] = V[:, perm[k]] * scale[k]      (eigenvectors in decreasing eigenvalue order; scale = 1 or sqrt(max(l, 0)))
__global__ __launch_bounds__(256) void gather_scale_cols_kernel(const double* V, int m, const int* perm, const double* scale, double* out,
                                                                int64_t ldo) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  for (int k = blockIdx.y; k < m; k += gridDim.y) out[i + (int64_t)k * ldo] = V[i + (int64_t)perm[k] * m] * (scale ? scale[k] : 1.0);
}

// out[i, j] = mean[i] + sum_k L[i,k] Z[k,j]  for up to 8 draws j per pass; thread = row i
template <int ND>
__global__ __launch_bounds__(256) void affine_lz_kernel(const double* L, int64_t ldl, int64_t m, const double* mean, const double* Z,
                                                        int64_t ldz, double* out, int64_t ldo, int lower) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  double acc[ND];
#pragma unroll
  for (int j = 0; j < ND; ++j) acc[j] = 0.0;
  const int64_t kend = lower ? i + 1 : m;  // a Cholesky factor is lower triangular: skip the structural zeros
  for (int64_t k = 0; k < kend; ++k) {
    const double l = L[i + k * ldl];
#pragma unroll
    for (int j = 0; j < ND; ++j) acc[j] += l * Z[k + j * ldz];
  }
#pragma unroll
  for (int j = 0; j < ND; ++j) out[i + j * ldo] = mean[i] + acc[j];
}

// combine_all(lst) of R/simulation.R:338-349: every combination of the axis values, one point per column, the LAST
// axis varying fastest: out[k, c] = lst[[k]][(c / each_k) % len_k], each_k = prod_{j > k} len_j.  Axis values are
// concatenated in `vals`; `offs[k]` is where axis k starts.  d <= 64 (kernel argument struct).
struct GridSpec { int d; int64_t len[64], each[64], offs[64]; };
__global__ __launch_bounds__(256) void combine_all_kernel(GridSpec g, const double* vals, int64_t total, double* out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= total) return;
  for (int k = 0; k < g.d; ++k) out[c * g.d + k] = vals[g.offs[k] + (c / g.each[k]) % g.len[k]];
}

inline unsigned blocks(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }

}  // namespace

int launch_pack_dense(hipStream_t s, const double* A, int64_t lda, int64_t m, int64_t n_pad, double* packed) {
  const unsigned gy = (unsigned)(n_pad < 16384 ? n_pad : 16384);
  hipLaunchKernelGGL(pack_dense_kernel, dim3(blocks(n_pad, 256), gy), dim3(256), 0, s, A, lda, m, n_pad, packed);
  GPRC_LAUNCH_CHECK();
  return 0;
}

int launch_sym_copy(hipStream_t s, const double* A, int64_t lda, int64_t m, double* W, double* V) {
  const unsigned gy = (unsigned)(m < 16384 ? m : 16384);
  hipLaunchKernelGGL(sym_copy_kernel, dim3(blocks(m, 256), gy), dim3(256), 0, s, A, lda, m, W, V);
  GPRC_LAUNCH_CHECK();
  return 0;
}

// one sweep = mm - 1 rounds
int launch_jacobi_sweep(hipStream_t s, double* W, double* V, int m, double* cs) {
  const int mm = m + (m & 1);
  const int npairs = mm / 2;
  ProfScope ps(s, PK_JACOBI, 12.0 * (double)m * m * (mm - 1), 48.0 * (double)m * m * (mm - 1));
  const unsigned gyc = (unsigned)(npairs < 1024 ? npairs : 1024), gyr = (unsigned)(m < 1024 ? m : 1024);
  for (int r = 0; r < mm - 1; ++r) {
    hipLaunchKernelGGL(jacobi_angles_kernel, dim3(blocks(npairs, 256)), dim3(256), 0, s, W, m, mm, r, cs);
    hipLaunchKernelGGL(jacobi_cols_kernel, dim3(blocks(m, 256), gyc), dim3(256), 0, s, W, V, m, mm, r, cs);
    hipLaunchKernelGGL(jacobi_rows_kernel, dim3(blocks(npairs, 256), gyr), dim3(256), 0, s, W, m, mm, r, cs);
  }
  GPRC_LAUNCH_CHECK();
  return 0;
}

int launch_jacobi_offnorm(hipStream_t s, const double* W, int m, double* off, double* dg) {
  hipLaunchKernelGGL(jacobi_offnorm_kernel, dim3((unsigned)m), dim3(256), 0, s, W, m, off, dg);
  GPRC_LAUNCH_CHECK();
  return 0;
}

int launch_gather_scale_cols(hipStream_t s, const double* V, int m, const int* perm, const double* scale, double* out, int64_t ldo) {
  const unsigned gy = (unsigned)(m < 16384 ? m : 16384);
  hipLaunchKernelGGL(gather_scale_cols_kernel, dim3(blocks(m, 256), gy), dim3(256), 0, s, V, m, perm, scale, out, ldo);
  GPRC_LAUNCH_CHECK();
  return 0;
}

int launch_affine_lz(hipStream_t s, const double* L, int64_t ldl, int64_t m, const double* mean, const double* Z, int64_t ldz,
                     int64_t ndraws, double* out, int64_t ldo, int lower) {
  if (m <= 0 || ndraws <= 0) return 0;
  const dim3 grid(blocks(m, 256)), block(256);
  int64_t j = 0;
  for (; j + 8 <= ndraws; j += 8)
    hipLaunchKernelGGL((affine_lz_kernel<8>), grid, block, 0, s, L, ldl, m, mean, Z + j * ldz, ldz, out + j * ldo, ldo, lower);
  for (; j < ndraws; ++j)
    hipLaunchKernelGGL((affine_lz_kernel<1>), grid, block, 0, s, L, ldl, m, mean, Z + j * ldz, ldz, out + j * ldo, ldo, lower);
  GPRC_LAUNCH_CHECK();
  return 0;
}

int launch_combine_all(hipStream_t s, const double* vals_dev, const int64_t* lengths, int d, double* out) {
  if (d < 1 || d > 64) { set_error("combine_all: 1 <= number of axes <= 64"); return GPRC_ERR_ARG; }
  GridSpec g{};
  g.d = d;
  int64_t off = 0, each = 1;
  for (int k = 0; k < d; ++k) { g.len[k] = lengths[k]; g.offs[k] = off; off += lengths[k]; }
  for (int k = d - 1; k >= 0; --k) { g.each[k] = each; each *= lengths[k]; }
  if (each <= 0) return 0;
  hipLaunchKernelGGL(combine_all_kernel, dim3(blocks(each, 256)), dim3(256), 0, s, g, vals_dev, each, out);
  GPRC_LAUNCH_CHECK();
  return 0;
}

}  // namespace gprc
