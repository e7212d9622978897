// kernels_vec.hip -- the HBM-bound vector stages around the factorisation (gfx950).
//
//   trsv            alpha <- solve(t(L), solve(L, y))           reference R/GPRclass.R:152, R/GPCclass.R:82-83
//   row_reduce      t(K_star) %*% alpha ; colSums(v * v) ; K %*% b   R/GPRclass.R:161,164 ; R/GPCclass.R:82,85,113,115
//   logp / diag_sum -0.5 y.alpha - sum(log(diag(L))) - n/2 log(2 pi)  R/GPRclass.R:153 ; R/GPCclass.R:103
//   gpc_*           the elementwise IRLS stages                  R/GPCclass.R:78-86
// All reductions use a fixed summation order (no floating-point atomics): results are bitwise
// reproducible run to run.
#include "gprc_internal.h"

namespace gprc {

namespace {

// address of L[r][c] (global indices, r >= panel start) in the packed block-column layout
__device__ __forceinline__ const double* packed_at(const double* packed, int64_t n_pad, int64_t r, int64_t c) {
  const int64_t p = c / NB;
  return packed + panel_offset(n_pad, p) + (r - p * NB) + (c - p * NB) * panel_ld(n_pad, p);
}

// deterministic block sum of one value per thread (blockDim.x <= 1024, power of two)
__device__ __forceinline__ double block_sum(double v, double* red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// ---- forward substitution by 128-blocks --------------------------------------------------------
// x_blk = Winv_blk * b_blk (in place), 128 threads
__device__ __forceinline__ void diag_apply(const double* W, double* bblk, double* sh) {
  const int i = threadIdx.x;
  sh[i] = bblk[i];
  __syncthreads();
  double s = 0.0;
  for (int c = 0; c <= i; ++c) s = fma(W[i + c * 128], sh[c], s);
  bblk[i] = s;
}

__global__ __launch_bounds__(128) void trsv_fwd_first(const double* winv, double* b) {
  __shared__ double sh[128];
  diag_apply(winv, b, sh);
}

// after x_blk is known: b[rb] -= L[rb, blk] * x_blk for every block row rb > blk; the workgroup that
// owns rb == blk+1 then finishes x_{blk+1}.
__global__ __launch_bounds__(128) void trsv_fwd_step(const double* packed, const double* winv, int64_t n_pad, int blk, double* b) {
  __shared__ double xs[128];
  __shared__ double sh[128];
  const int i = threadIdx.x;
  const int64_t rb = (int64_t)blk + 1 + blockIdx.x;
  xs[i] = b[(int64_t)blk * 128 + i];
  __syncthreads();
  const int64_t row = rb * 128 + i;
  const double* Lr = packed_at(packed, n_pad, row, (int64_t)blk * 128);
  const int64_t ld = panel_ld(n_pad, ((int64_t)blk * 128) / NB);
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int c = 0; c < 128; c += 4) {
    s0 = fma(Lr[(int64_t)c * ld], xs[c], s0);
    s1 = fma(Lr[(int64_t)(c + 1) * ld], xs[c + 1], s1);
    s2 = fma(Lr[(int64_t)(c + 2) * ld], xs[c + 2], s2);
    s3 = fma(Lr[(int64_t)(c + 3) * ld], xs[c + 3], s3);
  }
  b[row] -= (s0 + s1) + (s2 + s3);
  if (blockIdx.x == 0) {
    __syncthreads();
    diag_apply(winv + rb * 128 * 128, b + rb * 128, sh);
  }
}

// ---- backward substitution (L^T) ---------------------------------------------------------------
constexpr int BW_ROWS = 1024;  // rows of L per workgroup in the partial dot products

// part[g][c] = sum over this workgroup's rows r (below block blk) of L[r][blk*128 + c] * x[r]
__global__ __launch_bounds__(256) void trsv_bwd_partial(const double* packed, int64_t n_pad, int blk, const double* x, double* part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = ((int64_t)blk + 1) * 128 + (int64_t)blockIdx.x * BW_ROWS;
  const int64_t rend = (r0 + BW_ROWS < n_pad) ? r0 + BW_ROWS : n_pad;
  const int64_t c0 = (int64_t)blk * 128;
  const int64_t ld = panel_ld(n_pad, c0 / NB);
  const double* Lb = packed_at(packed, n_pad, r0, c0);
  for (int c = wave; c < 128; c += 4) {
    const double* col = Lb + (int64_t)c * ld;
    double s = 0.0;
    for (int64_t r = r0 + lane; r < rend; r += 64) s = fma(col[r - r0], x[r], s);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) part[(int64_t)blockIdx.x * 128 + c] = s;
  }
}

// x_blk = Winv_blk^T * (z_blk - sum_g part[g])
__global__ __launch_bounds__(128) void trsv_bwd_diag(const double* winv, int blk, int nparts, const double* part, double* x) {
  __shared__ double sh[128];
  const int i = threadIdx.x;
  double v = x[(int64_t)blk * 128 + i];
  for (int g = 0; g < nparts; ++g) v -= part[(int64_t)g * 128 + i];
  sh[i] = v;
  __syncthreads();
  const double* W = winv + (int64_t)blk * 128 * 128;
  double s = 0.0;
  for (int c = i; c < 128; ++c) s = fma(W[c + i * 128], sh[c], s);
  x[(int64_t)blk * 128 + i] = s;
}

// ---- row reductions over a tall column-major matrix --------------------------------------------
constexpr int RR_COLS = 512;  // columns per split

// part[split][row] = sum_{j in split} vt[row + j*ld] * (w ? w[j] : vt[row + j*ld])
__global__ __launch_bounds__(128) void row_reduce_partial(const double* vt, int64_t ld, int64_t cols, const double* w, double* part,
                                                          int64_t rows) {
  const int64_t row = (int64_t)blockIdx.x * 128 + threadIdx.x;
  const int64_t j0 = (int64_t)blockIdx.y * RR_COLS;
  const int64_t j1 = (j0 + RR_COLS < cols) ? j0 + RR_COLS : cols;
  const double* p = vt + row + j0 * ld;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int64_t j = j0;
  if (w) {
    for (; j + 4 <= j1; j += 4, p += 4 * ld) {
      s0 = fma(p[0], w[j], s0);
      s1 = fma(p[ld], w[j + 1], s1);
      s2 = fma(p[2 * ld], w[j + 2], s2);
      s3 = fma(p[3 * ld], w[j + 3], s3);
    }
    for (; j < j1; ++j, p += ld) s0 = fma(p[0], w[j], s0);
  } else {
    for (; j + 4 <= j1; j += 4, p += 4 * ld) {
      const double a = p[0], b = p[ld], c = p[2 * ld], e = p[3 * ld];
      s0 = fma(a, a, s0);
      s1 = fma(b, b, s1);
      s2 = fma(c, c, s2);
      s3 = fma(e, e, s3);
    }
    for (; j < j1; ++j, p += ld) { const double a = p[0]; s0 = fma(a, a, s0); }
  }
  part[(int64_t)blockIdx.y * rows + row] = (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(256) void row_reduce_final(const double* part, int64_t rows, int splits, double* out) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  double s = 0.0;
  for (int g = 0; g < splits; ++g) s += part[(int64_t)g * rows + row];
  out[row] = s;
}

// ---- scalars -----------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void logp_kernel(const double* packed, int64_t n_pad, int64_t n, const double* y, const double* alpha,
                                                    double* out) {
  __shared__ double red[1024];
  double ya = 0.0, sl = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    ya = fma(y[i], alpha[i], ya);
    sl += log(*packed_at(packed, n_pad, i, i));
  }
  const double tya = block_sum(ya, red);
  const double tsl = block_sum(sl, red);
  if (threadIdx.x == 0) out[0] = -0.5 * tya - tsl - (double)n / 2.0 * log(2.0 * M_PI);
}

__global__ __launch_bounds__(1024) void diag_sum_kernel(const double* packed, int64_t n_pad, int64_t n, double* out) {
  __shared__ double red[1024];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) s += *packed_at(packed, n_pad, i, i);
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = t;
}

__global__ __launch_bounds__(256) void unpack_kernel(const double* packed, int64_t n_pad, int64_t n, double* out, int64_t ld_out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  for (int64_t j = blockIdx.y; j < n; j += gridDim.y) out[i + j * ld_out] = (i >= j) ? *packed_at(packed, n_pad, i, j) : 0.0;
}

__global__ __launch_bounds__(256) void sub_kernel(const double* a, const double* b, double* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = a[i] - b[i];
}

// ---- GPC (Laplace / IRLS) stages ----------------------------------------------------------------
__device__ __forceinline__ double sigmoid(double x) { return 1.0 / (1.0 + exp(-x)); }  // R/GPCclass.R:63

// P = sigmoid(f); W = (1-P)*P; sw = sqrt(W); b = W*f + (y+1)/2 - P   (R/GPCclass.R:78-81); zero in the padding
__global__ __launch_bounds__(256) void gpc_pre_kernel(const double* f, const double* y, int64_t n, int64_t n_pad, double* sw, double* b) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pad) return;
  if (i >= n) { sw[i] = 0.0; b[i] = 0.0; return; }
  const double P = sigmoid(f[i]);
  const double W = (1.0 - P) * P;
  sw[i] = sqrt(W);
  b[i] = W * f[i] + (y[i] + 1.0) / 2.0 - P;
}
// g = (y+1)/2 - P ; sw = sqrt(P*(1-P))   (R/GPCclass.R:110-113)
__global__ __launch_bounds__(256) void gpc_grad_kernel(const double* f, const double* y, int64_t n, int64_t n_pad, double* g, double* sw) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pad) return;
  if (i >= n) { sw[i] = 0.0; g[i] = 0.0; return; }
  const double P = sigmoid(f[i]);
  sw[i] = sqrt(P * (1.0 - P));
  g[i] = (y[i] + 1.0) / 2.0 - P;
}
__global__ __launch_bounds__(256) void mul_kernel(const double* a, const double* b, double* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = a[i] * b[i];
}
__global__ __launch_bounds__(256) void gpc_a_kernel(const double* b, const double* sw, const double* t, double* a, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = b[i] - sw[i] * t[i];  // R/GPCclass.R:84
}
// objective = -sum(a*f)/2 - sum(log(1 + exp(-y*f)))   (R/GPCclass.R:86)
__global__ __launch_bounds__(1024) void gpc_objective_kernel(const double* a, const double* f, const double* y, int64_t n, double* out) {
  __shared__ double red[1024];
  double s1 = 0.0, s2 = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    s1 = fma(a[i], f[i], s1);
    s2 += log(1.0 + exp(-y[i] * f[i]));
  }
  const double t1 = block_sum(s1, red);
  const double t2 = block_sum(s2, red);
  if (threadIdx.x == 0) out[0] = -t1 / 2.0 - t2;
}
// packed lower part of B = I + (sw sw^T) o K   (R/GPCclass.R:80); K is dense n_pad x n_pad, zero padded
__global__ __launch_bounds__(256) void gpc_build_B_kernel(const double* K, int64_t n_pad, const double* sw, double* packed) {
  for (int64_t j = blockIdx.y; j < n_pad; j += gridDim.y) {  // global column
    const int64_t p = j / NB;
    const int64_t i = p * NB + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pad) continue;
    const double v = (i == j ? 1.0 : 0.0) + (sw[i] * sw[j]) * K[i + j * n_pad];
    packed[panel_offset(n_pad, p) + (i - p * NB) + (j - p * NB) * panel_ld(n_pad, p)] = v;
  }
}
__global__ __launch_bounds__(256) void scale_cols_kernel(double* vt, int64_t ld, int64_t rows, int64_t cols, const double* cs) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows) return;
  for (int64_t j = blockIdx.y; j < cols; j += gridDim.y) vt[i + j * ld] *= cs[j];
}

inline unsigned blocks(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

int64_t rowreduce_splits(int64_t cols) { return (cols + RR_COLS - 1) / RR_COLS; }

int launch_trsv(hipStream_t s, const double* packed, const double* winv, int64_t n_pad, double* b, int transpose, double* work) {
  const int nblk = (int)(n_pad / 128);
  ProfScope ps(s, PK_TRSV, (double)n_pad * n_pad, 8.0 * 0.5 * n_pad * n_pad);
  if (!transpose) {
    hipLaunchKernelGGL(trsv_fwd_first, dim3(1), dim3(128), 0, s, winv, b);
    for (int blk = 0; blk + 1 < nblk; ++blk)
      hipLaunchKernelGGL(trsv_fwd_step, dim3(nblk - 1 - blk), dim3(128), 0, s, packed, winv, n_pad, blk, b);
  } else {
    for (int blk = nblk - 1; blk >= 0; --blk) {
      const int64_t below = n_pad - ((int64_t)blk + 1) * 128;
      const int nparts = (int)((below + BW_ROWS - 1) / BW_ROWS);
      if (nparts > 0) hipLaunchKernelGGL(trsv_bwd_partial, dim3(nparts), dim3(256), 0, s, packed, n_pad, blk, b, work);
      hipLaunchKernelGGL(trsv_bwd_diag, dim3(1), dim3(128), 0, s, winv, blk, nparts, work, b);
    }
  }
  GPRC_LAUNCH_CHECK();
  return 0;
}

int launch_row_reduce(hipStream_t s, const double* vt, int64_t ld, int64_t rows, int64_t cols, const double* w, double* out,
                      double* work) {
  if (rows <= 0) return 0;
  if (rows % 128) { set_error("row_reduce: rows must be a multiple of 128"); return GPRC_ERR_ARG; }
  const int64_t splits = rowreduce_splits(cols);
  if (splits > 65535) { set_error("row_reduce: too many column splits"); return GPRC_ERR_ARG; }
  ProfScope ps(s, PK_ROWREDUCE, 2.0 * rows * cols, 8.0 * rows * cols);
  hipLaunchKernelGGL(row_reduce_partial, dim3((unsigned)(rows / 128), (unsigned)splits), dim3(128), 0, s, vt, ld, cols, w, work, rows);
  hipLaunchKernelGGL(row_reduce_final, dim3(blocks(rows, 256)), dim3(256), 0, s, work, rows, (int)splits, out);
  GPRC_LAUNCH_CHECK();
  return 0;
}

int launch_logp(hipStream_t s, const double* packed, int64_t n_pad, int64_t n, const double* y, const double* alpha, double* out) {
  hipLaunchKernelGGL(logp_kernel, dim3(1), dim3(1024), 0, s, packed, n_pad, n, y, alpha, out);
  GPRC_LAUNCH_CHECK();
  return 0;
}
int launch_diag_sum(hipStream_t s, const double* packed, int64_t n_pad, int64_t n, double* out) {
  hipLaunchKernelGGL(diag_sum_kernel, dim3(1), dim3(1024), 0, s, packed, n_pad, n, out);
  GPRC_LAUNCH_CHECK();
  return 0;
}
int launch_unpack_L(hipStream_t s, const double* packed, int64_t n_pad, int64_t n, double* out, int64_t ld_out) {
  if (n <= 0) return 0;
  const unsigned gy = (unsigned)(n < 16384 ? n : 16384);
  hipLaunchKernelGGL(unpack_kernel, dim3(blocks(n, 256), gy), dim3(256), 0, s, packed, n_pad, n, out, ld_out);
  GPRC_LAUNCH_CHECK();
  return 0;
}
int launch_sub(hipStream_t s, const double* a, const double* b, double* out, int64_t n) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(sub_kernel, dim3(blocks(n, 256)), dim3(256), 0, s, a, b, out, n);
  GPRC_LAUNCH_CHECK();
  return 0;
}
int launch_gpc_pre(hipStream_t s, const double* f, const double* y, int64_t n, double* sw, double* b) {
  const int64_t n_pad = pad_up(n, NB);
  hipLaunchKernelGGL(gpc_pre_kernel, dim3(blocks(n_pad, 256)), dim3(256), 0, s, f, y, n, n_pad, sw, b);
  GPRC_LAUNCH_CHECK();
  return 0;
}
int launch_gpc_grad(hipStream_t s, const double* f, const double* y, int64_t n, double* g, double* sw) {
  const int64_t n_pad = pad_up(n, NB);
  hipLaunchKernelGGL(gpc_grad_kernel, dim3(blocks(n_pad, 256)), dim3(256), 0, s, f, y, n, n_pad, g, sw);
  GPRC_LAUNCH_CHECK();
  return 0;
}
int launch_gpc_scale(hipStream_t s, const double* sw, const double* v, double* out, int64_t n) {
  hipLaunchKernelGGL(mul_kernel, dim3(blocks(n, 256)), dim3(256), 0, s, sw, v, out, n);
  GPRC_LAUNCH_CHECK();
  return 0;
}
int launch_gpc_a(hipStream_t s, const double* b, const double* sw, const double* t, double* a, int64_t n) {
  hipLaunchKernelGGL(gpc_a_kernel, dim3(blocks(n, 256)), dim3(256), 0, s, b, sw, t, a, n);
  GPRC_LAUNCH_CHECK();
  return 0;
}
int launch_gpc_objective(hipStream_t s, const double* a, const double* f, const double* y, int64_t n, double* out) {
  hipLaunchKernelGGL(gpc_objective_kernel, dim3(1), dim3(1024), 0, s, a, f, y, n, out);
  GPRC_LAUNCH_CHECK();
  return 0;
}
int launch_gpc_build_B(hipStream_t s, const double* Kfull, int64_t n_pad, const double* sw, double* packed) {
  const unsigned gy = (unsigned)(n_pad < 16384 ? n_pad : 16384);
  hipLaunchKernelGGL(gpc_build_B_kernel, dim3(blocks(n_pad, 256), gy), dim3(256), 0, s, Kfull, n_pad, sw, packed);
  GPRC_LAUNCH_CHECK();
  return 0;
}
int launch_scale_cols(hipStream_t s, double* vt, int64_t ld, int64_t rows, int64_t cols, const double* colscale) {
  if (rows <= 0 || cols <= 0) return 0;
  const unsigned gy = (unsigned)(cols < 16384 ? cols : 16384);
  hipLaunchKernelGGL(scale_cols_kernel, dim3(blocks(rows, 256), gy), dim3(256), 0, s, vt, ld, rows, cols, colscale);
  GPRC_LAUNCH_CHECK();
  return 0;
}

}  // namespace gprc
