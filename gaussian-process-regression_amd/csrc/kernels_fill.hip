// kernels_fill.hip -- fused pairwise-distance + covariance-function fill for gfx950.
//
// Replaces covariance_matrix() = outer(1:nA, 1:nB, function(i,j) k(A[,i], B[,j]))
// (reference R/GPRclass.R:355-357) together with the column-wise kernel bodies (R/GPRclass.R:382-402).
// The reference materialises two d x (nA*nB) gathers and 4-6 temporaries of that size; here each
// 128 x 64 output tile stages its 128 + 64 points through LDS once and the only HBM traffic is the
// 8 B/element store (16 B per lane, 1 KiB contiguous per wave) -- the kernel is HBM-write-bound.
//
// Differences from the reference arithmetic (within the 1e-10 normwise tolerance):
// colSums() accumulates in 80-bit long double in R; here the d-term sum is an fp64 FMA chain.
// The direct sum (x-y)^2 form is kept (no Gram trick: ||x||^2+||y||^2-2x.y cancels catastrophically).
#include "gprc_internal.h"

namespace gprc {

namespace {

constexpr int FT_R = 128;  // tile rows: 2 consecutive rows per lane x 64 lanes
constexpr int FT_C = 64;   // tile cols: 16 per wave x 4 waves
constexpr int FD = 16;     // coordinates staged per pass

struct FillArgs {
  const double* A;
  const double* B;
  double* out;
  int64_t nA, nB, d, ld;
  int64_t row0, nrows, col0, ncols;
  int mode;
  double noise;
  KernelSpec ks;
  // fused predict epilogue (FUSE instantiations only; R/GPRclass.R:161, R/GPCclass.R:113-114)
  const double* w;         // mpart[tile_col][row] = sum over the tile's 64 columns of k(A_row, B_col) * w[col]   (K*^T alpha)
  double* mpart;           // (number of column tiles) x mpart_rows
  int64_t mpart_rows;
  const double* colscale;  // stored value = k(.,.) * colscale[col]   (GPC: sqrt(W) * K_star); the mean uses the unscaled value
};

// base R `^` for doubles (arithmetic.c R_POW / R_pow): x^2 is x*x, the rest is libm pow
__device__ __forceinline__ double r_pow(double x, double y) {
  if (y == 2.0) return x * x;
  if (x == 1.0 || y == 0.0) return 1.0;
  if (x == 0.0) return y > 0.0 ? 0.0 : (y < 0.0 ? __builtin_huge_val() : y);
  return pow(x, y);
}

// per-coordinate accumulation term
template <int KID>
__device__ __forceinline__ double accum(double s, double a, double b, double sig) {
  if constexpr (KID == GPRC_LINEAR) return fma(sig * a, b, s);  // colSums(sigma * x * y)
  else if constexpr (KID == GPRC_POLYNOMIAL) return fma(a, b, s);  // colSums(x * y)
  else if constexpr (KID == GPRC_CONSTANT) return s;
  else {
    double t = a - b;  // colSums((x - y)^2)
    return fma(t, t, s);
  }
}

template <int KID>
__device__ __forceinline__ double finish(double s, const KernelSpec& ks) {
  if constexpr (KID == GPRC_CONSTANT) return ks.p[0];
  else if constexpr (KID == GPRC_LINEAR) return s;
  else if constexpr (KID == GPRC_POLYNOMIAL) {
    // (x . y + sigma)^p.  p = 2 is R's x * x; an integer degree 3..8 (flagged by make_fill_spec in p[2]; the base may be negative) is formed by
    // multiplications -- within 3 ulp of a correctly rounded pow, inside the 1e-13 gate of the fills -- instead of libm pow (~100 instructions: the
    // polynomial fill ran at 1.05 TB/s against 3.6 for the linear kernel); every other degree stays with R_pow.
    const double b = s + ks.p[0];
    const int h = (int)ks.p[2];
    if (h != 0) {                       // (wave-uniform: a launch constant)
      const double b2 = b * b, b4 = b2 * b2;
      switch (h) {
        case 3: return b2 * b;
        case 4: return b4;
        case 5: return b4 * b;
        case 6: return b4 * b2;
        case 7: return b4 * (b2 * b);
        default: return b4 * b4;
      }
    }
    return r_pow(b, ks.p[1]);
  }
  else if constexpr (KID == GPRC_SQREXP) {
    // -s / (2 l^2): the divisor is a launch constant, so divide by reciprocal + one fma correction (the residual
    // step of the usual division sequence) instead of the ~15-instruction generic fp64 division
    const double c = ks.p[1], rc = ks.p[2];  // 2 l^2 and its reciprocal, precomputed on the host (make_fill_spec)
    double q = s * rc;
    q = fma(fma(-q, c, s), rc, q);
    return exp(-q);
  }
  else if constexpr (KID == GPRC_GAMMAEXP) {
    // exp(-(sqrt(s) / l)^gamma).  gamma == 2 is R's x^2 = x * x (R_pow); any other gamma > 0 went through libm pow -- a special-case ladder and
    // an extended-precision log / exp, ~3x the work of a plain log and exp: the gamma-exponential fill ran at 0.8 TB/s against 3.5 for sqexp.
    // (sqrt(s) / l)^gamma = exp(gamma / 2 * log(s / l^2)): the quotient by the launch constant l^2 as for sqexp (reciprocal + one fma correction),
    // no square root, no division.  s = 0 (the diagonal) gives log = -Inf, exp(-Inf) = 0, exp(-0) = 1, as R's 0^gamma = 0 does; the relative
    // difference of the power from a correctly rounded pow is <= (gamma / 2 |log(s / l^2)| + 2) ulp, and the kernel value is exp(-power) <= 1:
    // inside the 1e-13 gate of the fills by two orders of magnitude.
    const double g = ks.p[1];
    if (g == 2.0 || !(g > 0.0)) return exp(-r_pow(sqrt(s) / ks.p[0], g));   // (wave-uniform: a launch constant)
    const double c = ks.p[2], rc = ks.p[3];                                   // l^2 and its reciprocal (make_fill_spec)
    double u = s * rc;
    u = fma(fma(-u, c, s), rc, u);
    // gamma = 0.5, 1, 1.5 (flagged by make_fill_spec in p[4] = 2 gamma; gamma = 1 is the exponential kernel): u^(gamma / 2) from one or two square
    // roots -- correctly rounded each -- instead of a log and an exp
    const int h = (int)ks.p[4];
    if (h != 0) {                       // (wave-uniform: a launch constant)
      const double r = sqrt(u);
      if (h == 2) return exp(-r);
      const double rr = sqrt(r);
      return exp(h == 1 ? -rr : -(r * rr));
    }
    return exp(-exp(0.5 * g * log(u)));
  }
  else {
    // (1 + s / (2 alpha l^2))^(-alpha).  The quotient by the launch constant 2 alpha l^2 is formed exactly as a division would
    // (reciprocal + one fma correction, as for sqexp); q >= 1, so q^(-alpha) = exp(-alpha log q) -- two transcendental
    // evaluations of ~25 instructions each instead of the generic pow (special-case ladder + extended-precision log/exp,
    // ~3x the work: the rational-quadratic fill ran at 1.06 TB/s against 3.5 for sqexp).  The relative difference from a
    // correctly rounded pow is <= (alpha log q + 1) ulp -- 3e-16 for the C3 inputs, far inside the 1e-13 gate of the fills.
    // alpha == 2 keeps R's x^2 = x * x special case (R_pow).
    const double al = ks.p[1], c = ks.p[2], rc = ks.p[3];   // c = 2 alpha l^2, rc = 1 / c (make_fill_spec)
    double x = s * rc;
    x = fma(fma(-x, c, s), rc, x);
    const double q = 1.0 + x;
    if (al == 2.0) return 1.0 / (q * q);
    // Half-integer alpha (2 alpha = 1 .. 8, flagged by make_fill_spec in p[4]): q^(-alpha) = (1 / sqrt(q))^(2 alpha) -- v_rsq_f64 with two Newton
    // steps (9 operations) and at most three multiplications instead of a log and an exp (~70).  The fill is VALU-bound (~90 double-precision
    // operations per element at d = 8 with the log / exp form: 1.5 TB/s); this form: see profiles/r03_c3_bench.json.  Error <= ~4 ulp of a
    // correctly rounded pow (1 ulp of the root, amplified by 2 alpha <= 8): inside the 1e-13 gate of the fills by three orders of magnitude.
    const int h = (int)ks.p[4];
    if (h != 0) {
      double r = __builtin_amdgcn_rsq(q);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const double e = fma(-(q * r), r, 1.0);
        r = fma(0.5 * r, e, r);
      }
      const double r2 = r * r, r4 = r2 * r2;
      switch (h) {                      // (wave-uniform: a launch constant)
        case 1: return r;
        case 2: return r2;
        case 3: return r2 * r;
        case 4: return r4;
        case 5: return r4 * r;
        case 6: return r4 * r2;
        case 7: return r4 * (r2 * r);
        default: return r4 * r4;
      }
    }
    return exp(-al * log(q));
  }
}

// FUSE (cross-covariance chunks of the predict only): besides storing the tile, every workgroup leaves the partial
// product of its 128 rows with w over its 64 columns -- summed in a fixed order (a lane's 16 columns ascending, then
// the four waves), so the mean assembled from the partials is bitwise independent of where a row sits in a chunk --
// and may scale the stored columns (GPC).  The K*^T alpha pass over the stored matrix (one full HBM read) disappears.
template <int KID, bool VEC2, int MODE, bool FUSE>
__global__ __launch_bounds__(256) void fill_kernel(FillArgs a) {
  __shared__ __attribute__((aligned(16))) double As[FD][FT_R];
  __shared__ double Bs[FT_C][FD + 1];
  __shared__ double Sg[FD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t ti = a.row0 + (int64_t)blockIdx.x * FT_R;  // first global row of the tile
  const int64_t tj = a.col0 + (int64_t)blockIdx.y * FT_C;
  double s0[16], s1[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) { s0[c] = 0.0; s1[c] = 0.0; }

  for (int64_t r0 = 0; r0 < a.d; r0 += FD) {
    const int dc = (int)((a.d - r0 < FD) ? (a.d - r0) : FD);
    __syncthreads();
    for (int e = t; e < FT_R * dc; e += 256) {  // A points: point-major in memory
      int i = e / dc, r = e - i * dc;
      int64_t gi = ti + i;
      As[r][i] = (gi < a.nA) ? a.A[gi * a.d + r0 + r] : 0.0;
    }
    for (int e = t; e < FT_C * dc; e += 256) {
      int j = e / dc, r = e - j * dc;
      int64_t gj = tj + j;
      Bs[j][r] = (gj < a.nB) ? a.B[gj * a.d + r0 + r] : 0.0;
    }
    if (t < dc) Sg[t] = (KID == GPRC_LINEAR) ? (a.ks.n_params == 1 ? a.ks.p[0] : a.ks.p[r0 + t]) : 1.0;
    __syncthreads();
    for (int r = 0; r < dc; ++r) {
      const double2 av = *reinterpret_cast<const double2*>(&As[r][2 * lane]);
      const double sg = Sg[r];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const double b = Bs[wave * 16 + c][r];
        s0[c] = accum<KID>(s0[c], av.x, b, sg);
        s1[c] = accum<KID>(s1[c], av.y, b, sg);
      }
    }
  }

  const int64_t gi0 = ti + 2 * lane;
  const int64_t row_end = a.row0 + a.nrows, col_end = a.col0 + a.ncols;
  double m0 = 0.0, m1 = 0.0;  // FUSE: this lane's two rows times w over its 16 columns
  // interior tile: every row/column is a valid point, inside the output window, and (identity mode) off the diagonal
  const bool interior = VEC2 && ti + FT_R <= a.nA && ti + FT_R <= row_end && tj + FT_C <= a.nB && tj + FT_C <= col_end &&
                        (MODE != PAD_IDENTITY || ti + FT_R <= tj || tj + FT_C <= ti);
  if (interior) {
    double* dst = a.out + (gi0 - a.row0) + (tj + wave * 16 - a.col0) * a.ld;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      double v0 = finish<KID>(s0[c], a.ks), v1 = finish<KID>(s1[c], a.ks);
      if constexpr (FUSE) {
        const int64_t gj = tj + wave * 16 + c;
        if (a.w) { const double wj = a.w[gj]; m0 = fma(v0, wj, m0); m1 = fma(v1, wj, m1); }
        if (a.colscale) { const double cs = a.colscale[gj]; v0 *= cs; v1 *= cs; }
      }
      *reinterpret_cast<double2*>(dst + c * a.ld) = make_double2(v0, v1);
    }
    if constexpr (!FUSE) return;
  } else {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int64_t gj = tj + wave * 16 + c;
      if (gj >= col_end) break;
      double v0 = finish<KID>(s0[c], a.ks), v1 = finish<KID>(s1[c], a.ks);
      if (MODE == PAD_IDENTITY) {
        const bool jin = gj < a.nB;
        v0 = (jin && gi0 < a.nA) ? v0 + (gi0 == gj ? a.noise : 0.0) : (gi0 == gj ? 1.0 : 0.0);
        v1 = (jin && gi0 + 1 < a.nA) ? v1 + (gi0 + 1 == gj ? a.noise : 0.0) : (gi0 + 1 == gj ? 1.0 : 0.0);
      } else if (MODE == PAD_ZERO) {
        const bool jin = gj < a.nB;
        v0 = (jin && gi0 < a.nA) ? v0 : 0.0;
        v1 = (jin && gi0 + 1 < a.nA) ? v1 : 0.0;
      }
      if constexpr (FUSE) {  // identical arithmetic to the interior branch: a padding entry is an exact zero and adds nothing
        if (a.w) { const double wj = a.w[gj]; m0 = fma(v0, wj, m0); m1 = fma(v1, wj, m1); }
        if (a.colscale) { const double cs = a.colscale[gj]; v0 *= cs; v1 *= cs; }
      }
      double* dst = a.out + (gi0 - a.row0) + (gj - a.col0) * a.ld;
      if constexpr (VEC2) {
        if (gi0 + 1 < row_end) *reinterpret_cast<double2*>(dst) = make_double2(v0, v1);
        else if (gi0 < row_end) dst[0] = v0;
      } else {
        if (gi0 < row_end) dst[0] = v0;
        if (gi0 + 1 < row_end) dst[1] = v1;
      }
    }
  }
  if constexpr (FUSE) {
    if (a.w) {  // the four waves' partials, summed in wave order
      __syncthreads();  // everybody has left the accumulation loop: As is free
      As[wave][2 * lane] = m0;
      As[wave][2 * lane + 1] = m1;
      __syncthreads();
      if (t < FT_R) a.mpart[(int64_t)blockIdx.y * a.mpart_rows + (int64_t)blockIdx.x * FT_R + t] = ((As[0][t] + As[1][t]) + As[2][t]) + As[3][t];
    }
  }
}

template <int KID>
__global__ __launch_bounds__(256) void colwise_kernel(KernelSpec ks, const double* x, const double* y, int64_t d, int64_t m, double* out) {
  int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= m) return;
  double s = 0.0;
  for (int64_t r = 0; r < d; ++r) {
    double sg = (KID == GPRC_LINEAR) ? (ks.n_params == 1 ? ks.p[0] : ks.p[r]) : 1.0;
    s = accum<KID>(s, x[c * d + r], y[c * d + r], sg);
  }
  out[c] = finish<KID>(s, ks);
}

// device-side copy of the spec with derived constants (sqexp: p[1] = 2 l^2, p[2] = 1 / (2 l^2))
KernelSpec make_fill_spec(const KernelSpec& ks) {
  KernelSpec d = ks;
  if (ks.id == GPRC_SQREXP) {
    d.p[1] = 2.0 * (ks.p[0] * ks.p[0]);
    d.p[2] = 1.0 / d.p[1];
  }
  if (ks.id == GPRC_POLYNOMIAL) d.p[2] = (ks.p[1] >= 3.0 && ks.p[1] <= 8.0 && ks.p[1] == (double)(int)ks.p[1]) ? ks.p[1] : 0.0;
  if (ks.id == GPRC_GAMMAEXP) {
    d.p[2] = ks.p[0] * ks.p[0];
    d.p[3] = 1.0 / d.p[2];
    const double h = 2.0 * ks.p[1];                  // gamma = 0.5, 1, 1.5: the square-root form of the power (finish<GPRC_GAMMAEXP>)
    d.p[4] = (h == 1.0 || h == 2.0 || h == 3.0) ? h : 0.0;
  }
  if (ks.id == GPRC_RATQUAD) {
    d.p[2] = 2.0 * ks.p[1] * (ks.p[0] * ks.p[0]);   // 2 * alpha * l^2, in R's evaluation order
    d.p[3] = 1.0 / d.p[2];
    const double h = 2.0 * ks.p[1];                  // half-integer alpha: the rsqrt form of the power (finish<GPRC_RATQUAD>)
    d.p[4] = (h >= 1.0 && h <= 8.0 && h == (double)(int)h) ? h : 0.0;
  }
  return d;
}

template <int KID, int MODE>
int do_fill_mode(hipStream_t s, const FillArgs& a) {
  dim3 grid((unsigned)((a.nrows + FT_R - 1) / FT_R), (unsigned)((a.ncols + FT_C - 1) / FT_C));
  const bool vec2 = (a.ld % 2 == 0) && ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0);
  if constexpr (MODE == PAD_ZERO) {
    if (a.w || a.colscale) {
      if (vec2) hipLaunchKernelGGL((fill_kernel<KID, true, MODE, true>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((fill_kernel<KID, false, MODE, true>), grid, dim3(256), 0, s, a);
      GPRC_LAUNCH_CHECK();
      return 0;
    }
  }
  if (vec2) hipLaunchKernelGGL((fill_kernel<KID, true, MODE, false>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((fill_kernel<KID, false, MODE, false>), grid, dim3(256), 0, s, a);
  GPRC_LAUNCH_CHECK();
  return 0;
}
template <int KID>
int do_fill(hipStream_t s, const FillArgs& a) {
  switch (a.mode) {
    case PAD_IDENTITY: return do_fill_mode<KID, PAD_IDENTITY>(s, a);
    case PAD_ZERO: return do_fill_mode<KID, PAD_ZERO>(s, a);
    default: return do_fill_mode<KID, PAD_NONE>(s, a);
  }
}

// ---- derivative row sums for fit()'s gradient (R/fit.R:126-139) ---------------------------------------------------
// deriv(x, y, v...) of cov_dict (R/fit.R:4-31), with v bound POSITIONALLY as the reference's do.call does:
//   sqrexp (l)            r = |x-y| :  r^2/l^3 * exp(-r^2/(2 l^2))
//   gammaexp (gamma, l)   r = |x-y| :  ( -exp(-(r/l)^gamma) (r/l)^gamma log(r/l) ,  exp(-(r/l)^gamma) gamma r^gamma / l^(gamma+1) )
//   polynomial (sigma, p) s = x.y + sigma :  ( p s^(p-1) ,  s^p log(s) )
//   rationalquadratic (alpha, l)  r = |x-y|^2, q = r/(2 l^2 alpha) + 1 :
//                         ( q^-alpha (r - (2 l^2 alpha + r) log q) / (2 l^2 alpha + r) ,  r q^(-alpha-1) / l^3 )
// (gammaexp's first component is 0 * -Inf = NaN at r = 0, i.e. on the diagonal: kept, it decides what optim does.)
template <int KID>
__device__ __forceinline__ void deriv_pair(double s, double v0, double v1, double& g0, double& g1) {
  if constexpr (KID == GPRC_SQREXP) {
    g0 = s / (v0 * v0 * v0) * exp(-s / ((v0 * v0) * 2.0));
    g1 = 0.0;
  } else if constexpr (KID == GPRC_GAMMAEXP) {
    const double r = sqrt(s), rl = r / v1, e = exp(-r_pow(rl, v0));
    g0 = -e * r_pow(rl, v0) * log(rl);
    g1 = e * v0 * r_pow(r, v0) / r_pow(v1, v0 + 1.0);
  } else if constexpr (KID == GPRC_POLYNOMIAL) {
    const double t = s + v0;
    g0 = v1 * r_pow(t, v1 - 1.0);
    g1 = r_pow(t, v1) * log(t);
  } else {  // rationalquadratic
    const double c = 2.0 * (v1 * v1) * v0, q = s / c + 1.0;
    g0 = (r_pow(q, -v0) * (s - (c + s) * log(q))) / (c + s);
    g1 = (s * r_pow(q, -v0 - 1.0)) / (v1 * v1 * v1);
  }
}

template <int KID>
__global__ __launch_bounds__(256) void deriv_rowsum_kernel(double v0, double v1, const double* X, int64_t d, int64_t n, double* S) {
  const int64_t r = blockIdx.x;
  const double* xr = X + r * d;
  double a0 = 0.0, a1 = 0.0;
  for (int64_t c = threadIdx.x; c < n; c += 256) {
    const double* xc = X + c * d;
    double s = 0.0;
    for (int64_t k = 0; k < d; ++k) {
      if constexpr (KID == GPRC_POLYNOMIAL) s += xr[k] * xc[k];
      else { const double t = xr[k] - xc[k]; s += t * t; }
    }
    double g0, g1;
    deriv_pair<KID>(s, v0, v1, g0, g1);
    a0 += g0;
    a1 += g1;
  }
  __shared__ double red[2][4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a0 += __shfl_down(a0, off, 64);
    a1 += __shfl_down(a1, off, 64);
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a0; red[1][threadIdx.x >> 6] = a1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    S[r] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    if (KID != GPRC_SQREXP) S[n + r] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

__global__ __launch_bounds__(256) void set_identity_rows_kernel(double* vt, int64_t ld, int64_t rows, int64_t cols, int64_t row0) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // row of the chunk (contiguous direction)
  if (i >= rows) return;
  for (int64_t j = blockIdx.y; j < cols; j += gridDim.y) vt[i + j * ld] = (row0 + i == j) ? 1.0 : 0.0;
}

}  // namespace

static int fill_dispatch(hipStream_t s, const FillArgs& a) {
  switch (a.ks.id) {
    case GPRC_CONSTANT: return do_fill<GPRC_CONSTANT>(s, a);
    case GPRC_LINEAR: return do_fill<GPRC_LINEAR>(s, a);
    case GPRC_POLYNOMIAL: return do_fill<GPRC_POLYNOMIAL>(s, a);
    case GPRC_SQREXP: return do_fill<GPRC_SQREXP>(s, a);
    case GPRC_GAMMAEXP: return do_fill<GPRC_GAMMAEXP>(s, a);
    case GPRC_RATQUAD: return do_fill<GPRC_RATQUAD>(s, a);
    default: set_error("unknown kernel id"); return GPRC_ERR_ARG;
  }
}

int launch_fill(hipStream_t s, const KernelSpec& ks, const double* A, int64_t nA, const double* B, int64_t nB, int64_t d,
                double* out, int64_t ld, int64_t row0, int64_t nrows, int64_t col0, int64_t ncols, PadMode mode,
                double noise) {
  if (nrows <= 0 || ncols <= 0) return 0;
  if ((ncols + FT_C - 1) / FT_C > 65535) { set_error("fill: too many column tiles in one launch"); return GPRC_ERR_ARG; }
  FillArgs a{A, B, out, nA, nB, d, ld, row0, nrows, col0, ncols, (int)mode, noise, make_fill_spec(ks), nullptr, nullptr, 0, nullptr};
  ProfScope ps(s, PK_FILL, (double)nrows * ncols * (3.0 * d + 20.0), 8.0 * nrows * ncols);
  return fill_dispatch(s, a);
}

int64_t fill_mean_tiles(int64_t cols) { return (cols + FT_C - 1) / FT_C; }

// The predict's cross-covariance chunk with its fused epilogue: vt (m_pad x n_pad, zero padded) = K(X_star chunk, X)
// [times colscale per column]; mpart[t * m_pad + i] = partial of (K*^T w)_i over column tile t (fill_mean_tiles(n_pad) tiles).
int launch_fill_cross_fused(hipStream_t s, const KernelSpec& ks, const double* Xs, int64_t m, const double* X, int64_t n, int64_t d,
                            double* vt, int64_t ld, int64_t m_pad, int64_t n_pad, const double* w, double* mpart, const double* colscale) {
  if (m_pad <= 0 || n_pad <= 0) return 0;
  if (m_pad % FT_R) { set_error("fill_cross_fused: m_pad must be a multiple of 128"); return GPRC_ERR_ARG; }
  if (fill_mean_tiles(n_pad) > 65535) { set_error("fill: too many column tiles in one launch"); return GPRC_ERR_ARG; }
  if (w && !mpart) { set_error("fill_cross_fused: partial buffer missing"); return GPRC_ERR_ARG; }
  FillArgs a{Xs, X, vt, m, n, d, ld, 0, m_pad, 0, n_pad, (int)PAD_ZERO, 0.0, make_fill_spec(ks), w, mpart, m_pad, colscale};
  ProfScope ps(s, PK_FILL, (double)m_pad * n_pad * (3.0 * d + 22.0), 8.0 * m_pad * n_pad);
  return fill_dispatch(s, a);
}

int launch_colwise(hipStream_t s, const KernelSpec& ks, const double* x, const double* y, int64_t d, int64_t m, double* out) {
  if (m <= 0) return 0;
  dim3 grid((unsigned)((m + 255) / 256));
  const KernelSpec ksd = make_fill_spec(ks);
  switch (ks.id) {
    case GPRC_CONSTANT: hipLaunchKernelGGL((colwise_kernel<GPRC_CONSTANT>), grid, dim3(256), 0, s, ksd, x, y, d, m, out); break;
    case GPRC_LINEAR: hipLaunchKernelGGL((colwise_kernel<GPRC_LINEAR>), grid, dim3(256), 0, s, ksd, x, y, d, m, out); break;
    case GPRC_POLYNOMIAL: hipLaunchKernelGGL((colwise_kernel<GPRC_POLYNOMIAL>), grid, dim3(256), 0, s, ksd, x, y, d, m, out); break;
    case GPRC_SQREXP: hipLaunchKernelGGL((colwise_kernel<GPRC_SQREXP>), grid, dim3(256), 0, s, ksd, x, y, d, m, out); break;
    case GPRC_GAMMAEXP: hipLaunchKernelGGL((colwise_kernel<GPRC_GAMMAEXP>), grid, dim3(256), 0, s, ksd, x, y, d, m, out); break;
    case GPRC_RATQUAD: hipLaunchKernelGGL((colwise_kernel<GPRC_RATQUAD>), grid, dim3(256), 0, s, ksd, x, y, d, m, out); break;
    default: set_error("unknown kernel id"); return GPRC_ERR_ARG;
  }
  GPRC_LAUNCH_CHECK();
  return 0;
}

int launch_deriv_rowsum(hipStream_t s, int kernel, double v0, double v1, const double* X, int64_t d, int64_t n, double* S) {
  if (n <= 0) return 0;
  ProfScope ps(s, PK_DERIV, (double)n * n * (3.0 * d + 40.0), 8.0 * ((double)n * d + 2.0 * n));
  const dim3 grid((unsigned)n), block(256);
  switch (kernel) {
    case GPRC_SQREXP: hipLaunchKernelGGL((deriv_rowsum_kernel<GPRC_SQREXP>), grid, block, 0, s, v0, v1, X, d, n, S); break;
    case GPRC_GAMMAEXP: hipLaunchKernelGGL((deriv_rowsum_kernel<GPRC_GAMMAEXP>), grid, block, 0, s, v0, v1, X, d, n, S); break;
    case GPRC_POLYNOMIAL: hipLaunchKernelGGL((deriv_rowsum_kernel<GPRC_POLYNOMIAL>), grid, block, 0, s, v0, v1, X, d, n, S); break;
    case GPRC_RATQUAD: hipLaunchKernelGGL((deriv_rowsum_kernel<GPRC_RATQUAD>), grid, block, 0, s, v0, v1, X, d, n, S); break;
    default: set_error("fit gradient: the reference defines it for sqrexp, gammaexp, polynomial, rationalquadratic only (R/fit.R:125)"); return GPRC_ERR_ARG;
  }
  GPRC_LAUNCH_CHECK();
  return 0;
}

int launch_set_identity_rows(hipStream_t s, double* vt, int64_t ld, int64_t rows, int64_t cols, int64_t row0) {
  if (rows <= 0 || cols <= 0) return 0;
  hipLaunchKernelGGL(set_identity_rows_kernel, dim3((unsigned)((rows + 255) / 256), (unsigned)(cols < 4096 ? cols : 4096)), dim3(256), 0, s, vt, ld, rows, cols, row0);
  GPRC_LAUNCH_CHECK();
  return 0;
}

}  // namespace gprc
