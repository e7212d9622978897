// kernels_vec.hip -- the HBM-bound vector stages around the factorisation (gfx950).
//
//   trsv            alpha <- solve(t(L), solve(L, y))           reference R/GPRclass.R:152, R/GPCclass.R:82-83
//   row_reduce      t(K_star) %*% alpha ; colSums(v * v) ; K %*% b   R/GPRclass.R:161,164 ; R/GPCclass.R:82,85,113,115
//   logp / diag_sum -0.5 y.alpha - sum(log(diag(L))) - n/2 log(2 pi)  R/GPRclass.R:153 ; R/GPCclass.R:103
//   gpc_*           the elementwise IRLS stages                  R/GPCclass.R:78-86
// All reductions use a fixed summation order (no floating-point atomics): results are bitwise
// reproducible run to run.
#include <cstdlib>
#include <cstring>

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

// ---- triangular solves with the packed factor: ONE launch per NB-wide panel ------------------------
// Forward  (L x = b):   for p = 0..P-1:  x_p = inv(L_pp) b_p, then b[below] -= L[below, panel p] x_p.
// Backward (L^T x = z): for p = P-1..0:  x_p = inv(L_pp)^T z_p, then z[q] -= L[panel-p rows, panel q cols]^T x_p for q < p.
// inv(L_pp) is explicit (inv512_row_role in kernels_chol.hip: T = inv(L_pp)^T per panel, row r of the inverse contiguous at
// T + r NB), so the diagonal step is one 512 x 512 triangular product, spread over the first TRSV_DIAG_WGS workgroups of the
// panel's launch (32 rows / columns each); they write x_p to zbuf and count themselves into the panel's gate behind a release; the
// other workgroups of the SAME launch -- the wide product -- wait for the gate, read x_p and stream their rows / columns.  (Before:
// two launches per panel, the diagonal solve a chain of eight dependent small products through the 128-block inverses in ONE
// workgroup -- 35 of the 56..70 us a panel step took; 0.80 / 1.8 / 9.1 ms per solve at n = 8192 / 16384 / 65536.)
// The diagonal workgroups have the lowest block ids: they are dispatched first, and the whole grid (<= 16 + n / 128 workgroups of
// 8 waves and 5 KB of LDS) is resident at once for n <= 65536; every wait is bounded all the same (on a timeout the waiting
// workgroup poisons its outputs with NaN: the solve is visibly invalid, never silently wrong).
// Every reduction has a fixed order: results are bitwise reproducible, and the per-step entry point (one panel) is the same launch.

__device__ __forceinline__ double wave_sum(double s) {
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  return s;
}

struct TrsvGate { int done; int near; };   // one per panel, zeroed before the solve: diagonal workgroups that have published x_p; near-row
                                           // workgroups of the previous panel's launch that have finished (whole-solve form only)
constexpr int TRSV_DIAG_WGS = 16;
constexpr int TRSV_DIAG_ROWS = NB / TRSV_DIAG_WGS;   // 32
constexpr int TRSV_ROWS = 32;                        // rows per forward product workgroup (x 4 column groups x 4 fma chains = 512 threads)
constexpr int TRSV_COLS = 16;                        // columns per backward product workgroup (8 waves x 2 columns)
constexpr int TRSV_NEAR_FWD = NB / TRSV_ROWS;        // workgroups of a forward launch that cover the next panel's 512 rows
constexpr int TRSV_NEAR_BWD = NB / TRSV_COLS;        // workgroups of a backward launch that cover the previous panel's 512 columns

// the whole workgroup calls it; returns false on a timeout
__device__ __forceinline__ bool trsv_count_wait(int* ctr, int need, int* sh) {
  if (threadIdx.x == 0) {
    int spins = 0, ok = 1;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 24)) { ok = 0; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *sh = ok;
  }
  __syncthreads();
  return *sh != 0;
}
__device__ __forceinline__ void trsv_count_arrive(int* ctr) {   // the whole workgroup calls it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// One row's share of L[row, panel p columns] * x_p for column group g (NB/4 columns) is four interleaved fma chains (chain k: columns
// k, k + 4, ...), combined (s0 + s1) + (s2 + s3).  One thread runs ONE chain; the four chains of a (row, group) sit in adjacent lanes.
__device__ __forceinline__ double gemv_chain(const double* Lr, int64_t ld, const double* xg, int k) {
  double s = 0.0;
#pragma unroll 8
  for (int c = k; c < NB / 4; c += 4) s = fma(Lr[(int64_t)c * ld], xg[c], s);
  return s;
}

// One column's dot product with x_p over the NB rows of panel p (lanes stride the rows, fixed fma chain + wave tree)
__device__ __forceinline__ double gemvt_column_dot(const double* col, const double* xs, int lane) {
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < NB; r += 64) s = fma(col[r + lane], xs[r + lane], s);
  return wave_sum(s);
}

// ---- forward pieces (512 threads) ----
// diagonal role `role` (0..15) of panel p: x_p rows 32 role .. 32 role + 31 (a wave: 4 rows; a lane: 8 consecutive columns of each
// row, ascending, then the wave tree) -> zbuf, gate; b_p itself receives x_p once every diagonal workgroup has read it
__device__ __forceinline__ void trsv_fwd_diag(const double* inv, int p, double* b, double* zbuf, TrsvGate* gate, int role, double* xs, int* sh_ok) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  double* bp = b + (int64_t)p * NB;
  double* zp = zbuf + (int64_t)p * NB;
  const int r0 = TRSV_DIAG_ROWS * role + 4 * w;
  const double* T = inv + (int64_t)p * NB * NB + (int64_t)r0 * NB + 8 * lane;
  double tv[4][8];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) tv[k][e] = (8 * lane + e <= r0 + k) ? T[(int64_t)k * NB + e] : 0.0;   // the loads fly while b_p is staged
  for (int c = t; c < NB; c += 512) xs[c] = bp[c];
  __syncthreads();
  double v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (8 * lane + e <= r0 + k) s = fma(tv[k][e], xs[8 * lane + e], s);
    v[k] = wave_sum(s);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) zp[r0 + k] = v[k];
  }
  trsv_count_arrive(&gate->done);
  const bool ok = trsv_count_wait(&gate->done, TRSV_DIAG_WGS, sh_ok);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) bp[r0 + k] = ok ? v[k] : __builtin_nan("");
  }
}

// TRSV_ROWS rows starting at `row0` (global row index) x the NB columns of panel p: b[row] -= L[row, panel p] x_p.  Thread =
// (chain k = t & 3, row i = (t >> 2) & 31, column group g = t >> 7): a row's 16 chain sums are combined per group as (s0 + s1) +
// (s2 + s3) with two lane exchanges, the four groups as (r0 + r1) + (r2 + r3) through LDS -- the summation tree of a row is fixed.
// (32 rows per workgroup instead of 128: the 2 MB block next to the diagonal, and at small n the whole product, is spread over four
// times as many CUs -- a CU streams ~100 GB/s.)
__device__ __forceinline__ void trsv_fwd_rows(const double* packed, int64_t n_pad, int p, double* b, const double* zbuf, int64_t row0, bool ok,
                                              double* xs, double (*red)[TRSV_ROWS]) {
  const int t = threadIdx.x, k = t & 3, i = (t >> 2) & (TRSV_ROWS - 1), g = t >> 7;
  const int64_t ld = panel_ld(n_pad, p);
  const int64_t row = row0 + i;
  const double* Lr = packed + panel_offset(n_pad, p) + (row - (int64_t)p * NB) + (int64_t)(g * (NB / 4)) * ld;
  for (int c = t; c < NB; c += 512) xs[c] = zbuf[(int64_t)p * NB + c];
  __syncthreads();
  double s = gemv_chain(Lr, ld, xs + g * (NB / 4), k);
  s += __shfl_xor(s, 1, 64);                      // lanes k = 0, 1: s0 + s1; lanes 2, 3: s2 + s3 (addition commutes bitwise)
  s += __shfl_xor(s, 2, 64);                      // (s0 + s1) + (s2 + s3)
  if (k == 0) red[g][i] = s;
  __syncthreads();
  if (t < TRSV_ROWS) {
    const int64_t r = row0 + t;
    b[r] = ok ? b[r] - ((red[0][t] + red[1][t]) + (red[2][t] + red[3][t])) : __builtin_nan("");
  }
}

// The same product for 128 rows per workgroup (thread = row x column group, the four chains of a group in one thread): 1 KB
// contiguous per column instead of 256 B -- the better HBM access pattern, used for the rows far from the diagonal of large factors
// (n = 65536: 4.6 against 5.1 ms per forward solve); the summation tree of a row is the same, so are the bits.
__device__ __forceinline__ void trsv_fwd_rows128(const double* packed, int64_t n_pad, int p, double* b, const double* zbuf, int64_t row0, bool ok,
                                                 double* xs, double (*red)[128]) {
  const int t = threadIdx.x, i = t & 127, g = t >> 7;
  const int64_t ld = panel_ld(n_pad, p);
  const int64_t row = row0 + i;
  const double* Lr = packed + panel_offset(n_pad, p) + (row - (int64_t)p * NB) + (int64_t)(g * (NB / 4)) * ld;
  const double* xg = xs + g * (NB / 4);
  for (int c = t; c < NB; c += 512) xs[c] = zbuf[(int64_t)p * NB + c];
  __syncthreads();
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
  for (int c = 0; c < NB / 4; c += 4) {
    s0 = fma(Lr[(int64_t)c * ld], xg[c], s0);
    s1 = fma(Lr[(int64_t)(c + 1) * ld], xg[c + 1], s1);
    s2 = fma(Lr[(int64_t)(c + 2) * ld], xg[c + 2], s2);
    s3 = fma(Lr[(int64_t)(c + 3) * ld], xg[c + 3], s3);
  }
  red[g][i] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0) b[row] = ok ? b[row] - ((red[0][i] + red[1][i]) + (red[2][i] + red[3][i])) : __builtin_nan("");
}

// Forward launch of panel p.  far128 != 0: the rows behind the next panel's are taken 128 per workgroup (large factors).
// FUSED = false (gprc_dev_trsv_step): blocks [0, 16) the diagonal step of panel p; blocks >= 16: 32 rows below the panel each, after
//   the gate.
// FUSED = true (inside a whole solve; x_p is already in zbuf -- the previous launch left it): blocks [0, 16) the rows of panel p + 1,
//   counting into that panel's `near`; blocks [16, 32) the DIAGONAL STEP OF PANEL p + 1, which so runs while the other blocks
//   (>= 32: the rows behind panel p + 1) are still streaming; nobody waits for anything but those sixteen.
template <bool FUSED>
__global__ __launch_bounds__(512) void trsv_fwd_kernel(const double* packed, const double* inv, int64_t n_pad, int p, double* b, double* zbuf,
                                                       TrsvGate* gates, int far128) {
  __shared__ double xs[NB];
  __shared__ double red[4][TRSV_ROWS];
  __shared__ double red128[4][128];
  __shared__ int sh_ok;
  const int bid = (int)blockIdx.x;
  if constexpr (!FUSED) {
    if (bid < TRSV_DIAG_WGS) { trsv_fwd_diag(inv, p, b, zbuf, gates + p, bid, xs, &sh_ok); return; }
    const bool ok = trsv_count_wait(&gates[p].done, TRSV_DIAG_WGS, &sh_ok);
    const int rb = bid - TRSV_DIAG_WGS;                  // the next panel's rows always 32 per workgroup, the rest 32 or 128
    if (rb < TRSV_NEAR_FWD || !far128) trsv_fwd_rows(packed, n_pad, p, b, zbuf, (int64_t)(p + 1) * NB + (int64_t)rb * TRSV_ROWS, ok, xs, red);
    else trsv_fwd_rows128(packed, n_pad, p, b, zbuf, (int64_t)(p + 2) * NB + (int64_t)(rb - TRSV_NEAR_FWD) * 128, ok, xs, red128);
  } else {
    if (bid < TRSV_NEAR_FWD) {
      trsv_fwd_rows(packed, n_pad, p, b, zbuf, (int64_t)(p + 1) * NB + (int64_t)bid * TRSV_ROWS, true, xs, red);
      trsv_count_arrive(&gates[p + 1].near);
    } else if (bid < TRSV_NEAR_FWD + TRSV_DIAG_WGS) {
      const bool ok = trsv_count_wait(&gates[p + 1].near, TRSV_NEAR_FWD, &sh_ok);
      trsv_fwd_diag(inv, p + 1, b, zbuf, gates + p + 1, bid - TRSV_NEAR_FWD, xs, &sh_ok);
      if (!ok && threadIdx.x < TRSV_DIAG_ROWS) b[(int64_t)(p + 1) * NB + TRSV_DIAG_ROWS * (bid - TRSV_NEAR_FWD) + threadIdx.x] = __builtin_nan("");
    } else {
      const int rb = bid - TRSV_NEAR_FWD - TRSV_DIAG_WGS;
      if (!far128) trsv_fwd_rows(packed, n_pad, p, b, zbuf, (int64_t)(p + 2) * NB + (int64_t)rb * TRSV_ROWS, true, xs, red);
      else trsv_fwd_rows128(packed, n_pad, p, b, zbuf, (int64_t)(p + 2) * NB + (int64_t)rb * 128, true, xs, red128);
    }
  }
}

// ---- backward pieces (512 threads) ----
// diagonal role `role` of panel p: x_p columns 32 role .. 32 role + 31 (thread = column x one of 16 row groups r = g, g + 16, ..;
// the 16 partials of a column are added in group order)
__device__ __forceinline__ void trsv_bwd_diag(const double* inv, int p, double* x, double* zbuf, TrsvGate* gate, int role, double* xs,
                                              double (*red)[TRSV_DIAG_ROWS], int* sh_ok) {
  const int t = threadIdx.x;
  double* xp = x + (int64_t)p * NB;
  double* zp = zbuf + (int64_t)p * NB;
  const int cc = t & (TRSV_DIAG_ROWS - 1), g = t >> 5;
  const int c = TRSV_DIAG_ROWS * role + cc;
  const double* T = inv + (int64_t)p * NB * NB + c;        // T[c + r NB] = inv(L_pp)[r, c]
  double tv[NB / 16];
#pragma unroll
  for (int k = 0; k < NB / 16; ++k) {
    const int r = g + 16 * k;
    tv[k] = (r >= c) ? T[(int64_t)r * NB] : 0.0;
  }
  for (int r = t; r < NB; r += 512) xs[r] = xp[r];
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < NB / 16; ++k) {
    const int r = g + 16 * k;
    if (r >= c) s = fma(tv[k], xs[r], s);
  }
  red[g][cc] = s;
  __syncthreads();
  double v = 0.0;
  if (g == 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) v += red[q][cc];
    zp[c] = v;
  }
  trsv_count_arrive(&gate->done);
  const bool ok = trsv_count_wait(&gate->done, TRSV_DIAG_WGS, sh_ok);
  if (g == 0) xp[c] = ok ? v : __builtin_nan("");
}

// TRSV_COLS columns (group cg) of the earlier panel q: x[q NB + c] -= L[panel-p rows, that column]^T x_p; a wave per column, two
// columns per wave
__device__ __forceinline__ void trsv_bwd_cols(const double* packed, int64_t n_pad, int p, double* x, const double* zbuf, int q, int cg, bool ok, double* xs) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int64_t ldq = panel_ld(n_pad, q);
  const double* blk = packed + panel_offset(n_pad, q) + (int64_t)(p - q) * NB;  // rows of panel p inside panel q
  for (int r = t; r < NB; r += 512) xs[r] = zbuf[(int64_t)p * NB + r];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < TRSV_COLS / 8; ++k) {
    const int c = cg * TRSV_COLS + w * (TRSV_COLS / 8) + k;
    const double s = gemvt_column_dot(blk + (int64_t)c * ldq, xs, lane);
    if (lane == 0) x[(int64_t)q * NB + c] = ok ? x[(int64_t)q * NB + c] - s : __builtin_nan("");
  }
}

// Backward launch of panel p; FUSED as above with "previous panel" = p - 1: blocks [0, 32) its 512 columns (-> near), blocks [32, 48)
// its diagonal step, blocks >= 48 the panels q < p - 1.
template <bool FUSED>
__global__ __launch_bounds__(512) void trsv_bwd_kernel(const double* packed, const double* inv, int64_t n_pad, int p, double* x, double* zbuf,
                                                       TrsvGate* gates) {
  __shared__ double xs[NB];
  __shared__ double red[16][TRSV_DIAG_ROWS];
  __shared__ int sh_ok;
  const int bid = (int)blockIdx.x;
  if constexpr (!FUSED) {
    if (bid < TRSV_DIAG_WGS) { trsv_bwd_diag(inv, p, x, zbuf, gates + p, bid, xs, red, &sh_ok); return; }
    const int cb = bid - TRSV_DIAG_WGS;
    const bool ok = trsv_count_wait(&gates[p].done, TRSV_DIAG_WGS, &sh_ok);
    trsv_bwd_cols(packed, n_pad, p, x, zbuf, cb / (NB / TRSV_COLS), cb % (NB / TRSV_COLS), ok, xs);
  } else {
    if (bid < TRSV_NEAR_BWD) {
      trsv_bwd_cols(packed, n_pad, p, x, zbuf, p - 1, bid, true, xs);
      trsv_count_arrive(&gates[p - 1].near);
    } else if (bid < TRSV_NEAR_BWD + TRSV_DIAG_WGS) {
      const bool ok = trsv_count_wait(&gates[p - 1].near, TRSV_NEAR_BWD, &sh_ok);
      trsv_bwd_diag(inv, p - 1, x, zbuf, gates + p - 1, bid - TRSV_NEAR_BWD, xs, red, &sh_ok);
      if (!ok && threadIdx.x < TRSV_DIAG_ROWS) x[(int64_t)(p - 1) * NB + TRSV_DIAG_ROWS * (bid - TRSV_NEAR_BWD) + threadIdx.x] = __builtin_nan("");
    } else {
      const int cb = bid - TRSV_NEAR_BWD - TRSV_DIAG_WGS;
      trsv_bwd_cols(packed, n_pad, p, x, zbuf, cb / (NB / TRSV_COLS), cb % (NB / TRSV_COLS), true, xs);
    }
  }
}

// ---- row reductions ------------------------------------------------------------------------------
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

// out[i] = sum_t part[t * stride + i] (minuend == nullptr) or minuend[i] - sum_t part[t * stride + i]; t ascending
__global__ __launch_bounds__(256) void sum_partials_kernel(const double* part, int64_t nparts, int64_t stride, int64_t rows,
                                                           const double* minuend, double* out) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  double s = 0.0;
  for (int64_t t = 0; t < nparts; ++t) s += part[t * stride + row];
  out[row] = minuend ? minuend[row] - s : s;
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

// ---- GPC class probability: P(y* = +1) = integral of sigmoid(z) * N(z; mean = fs_bar, sd = Vfs) dz ------------
// Replaces the per-point stats::integrate() call of R/GPCclass.R:116-117 (QUADPACK dqagi, rel.tol 1.2e-4); the
// reference passes the VARIANCE Vfs as dnorm's sd -- kept.  Composite 16-point Gauss-Legendre, ~1e-14 accurate:
//   sd < 1 : in t = (z - mu)/sd over [-12, 12] (24 panels): the Gaussian sets the scale, sigmoid(mu + sd t) is slow;
//   sd >= 1: in z over [-40, 40] n [mu - 12 sd, mu + 12 sd] with panels <= 2 wide: the sigmoid sets the scale; beyond
//            z = 40 sigmoid is 1 to 4e-18 and the remaining Gaussian mass is added in closed form (erfc).
// sd <= 0 or NaN -> NaN (dnorm is undefined / a point mass there and integrate() stops in the reference).
__constant__ double GL_X[16] = {
    -9.89400934991649939e-01, -9.44575023073232600e-01, -8.65631202387831755e-01, -7.55404408355002999e-01,
    -6.17876244402643771e-01, -4.58016777657227370e-01, -2.81603550779258915e-01, -9.50125098376374544e-02,
    9.50125098376374544e-02,  2.81603550779258915e-01,  4.58016777657227370e-01,  6.17876244402643771e-01,
    7.55404408355002999e-01,  8.65631202387831755e-01,  9.44575023073232600e-01,  9.89400934991649939e-01};
__constant__ double GL_W[16] = {
    2.71524594117540374e-02, 6.22535239386477063e-02, 9.51585116824925914e-02, 1.24628971255534030e-01,
    1.49595988816576764e-01, 1.69156519395002619e-01, 1.82603415044923612e-01, 1.89450610455068585e-01,
    1.89450610455068585e-01, 1.82603415044923612e-01, 1.69156519395002619e-01, 1.49595988816576764e-01,
    1.24628971255534030e-01, 9.51585116824925914e-02, 6.22535239386477063e-02, 2.71524594117540374e-02};

__global__ __launch_bounds__(64) void gpc_class_prob_kernel(const double* fs, const double* vf, double* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const double mu = fs[i], sd = vf[i];
  if (!(sd > 0.0) || !(mu == mu)) { out[i] = __builtin_nan(""); return; }
  const double inv_sqrt_2pi = 0.3989422804014326779;
  double acc = 0.0;
  if (sd < 1.0) {
    for (int p = 0; p < 24; ++p) {
      const double c = -11.5 + p;  // panel [c - 0.5, c + 0.5]
      double s = 0.0;
      for (int q = 0; q < 16; ++q) {
        const double t = c + 0.5 * GL_X[q];
        s = fma(GL_W[q], sigmoid(mu + sd * t) * exp(-0.5 * t * t), s);
      }
      acc = fma(0.5 * inv_sqrt_2pi, s, acc);
    }
  } else {
    const double lo = fmax(-40.0, mu - 12.0 * sd), hi = fmin(40.0, mu + 12.0 * sd);
    if (hi > lo) {
      const int np = (int)ceil((hi - lo) * 0.5);
      const double h = (hi - lo) / np;
      for (int p = 0; p < np; ++p) {
        const double c = lo + (p + 0.5) * h;
        double s = 0.0;
        for (int q = 0; q < 16; ++q) {
          const double z = c + 0.5 * h * GL_X[q];
          const double t = (z - mu) / sd;
          s = fma(GL_W[q], sigmoid(z) * exp(-0.5 * t * t), s);
        }
        acc = fma(0.5 * h * inv_sqrt_2pi / sd, s, acc);
      }
    }
    if (mu + 12.0 * sd > 40.0) acc += 0.5 * erfc((fmax(40.0, lo) - mu) / (sd * 1.4142135623730951));  // sigmoid = 1 above z = 40
  }
  out[i] = acc;
}

inline unsigned blocks(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

int64_t rowreduce_splits(int64_t cols) { return (cols + RR_COLS - 1) / RR_COLS; }

// work (gprc_trsv_work_size(n_pad) doubles): x_p staging (n_pad doubles) and one gate per panel behind it
static inline TrsvGate* trsv_gates(double* work, int64_t n_pad) { return reinterpret_cast<TrsvGate*>(work + n_pad); }

// rows behind the next panel's in 128-row workgroups from this size on (the HBM access pattern outweighs the parallelism)
static inline int trsv_far128(int64_t n_pad) { return n_pad >= 32768 ? 1 : 0; }
// workgroups of a forward launch for `rows` rows below the panel (first the next panel's NB rows, 32 per workgroup)
static inline unsigned trsv_fwd_row_wgs(int64_t rows, int far128) {
  if (rows <= 0) return 0;
  const int64_t far = rows - NB;
  return (unsigned)(TRSV_NEAR_FWD + (far > 0 ? far / (far128 ? 128 : TRSV_ROWS) : 0));
}

// one panel step of the solve: x_p, then its contribution to the rest of the right-hand side (one launch)
int launch_trsv_step(hipStream_t s, const double* packed, const double* inv, int64_t n_pad, double* b, int transpose, int p, double* work) {
  const int P = (int)(n_pad / NB);
  if (p < 0 || p >= P) { set_error("trsv_step: panel out of range"); return GPRC_ERR_ARG; }
  if (!inv || !work) { set_error("trsv_step: the explicit diagonal inverses and the work buffer are required"); return GPRC_ERR_ARG; }
  TrsvGate* gates = trsv_gates(work, n_pad);
  GPRC_HIP(hipMemsetAsync(gates + p, 0, sizeof(TrsvGate), s));
  const int f128 = trsv_far128(n_pad);
  if (!transpose) {
    const int64_t below = n_pad - (int64_t)(p + 1) * NB;
    hipLaunchKernelGGL(trsv_fwd_kernel<false>, dim3(TRSV_DIAG_WGS + trsv_fwd_row_wgs(below, f128)), dim3(512), 0, s, packed, inv, n_pad, p, b, work, gates, f128);
  } else {
    hipLaunchKernelGGL(trsv_bwd_kernel<false>, dim3((unsigned)(TRSV_DIAG_WGS + p * (NB / TRSV_COLS))), dim3(512), 0, s, packed, inv, n_pad, p, b, work, gates);
  }
  GPRC_LAUNCH_CHECK();
  return 0;
}

// The whole solve: P launches.  The first is the diagonal step of the first panel alone; launch p then carries the product of
// panel p AND the diagonal step of the next panel (fused form): that step -- two dependent hand-offs -- runs beside the streaming of
// the product instead of in front of the next launch.  Element by element the arithmetic is that of the per-step launches
// (the same device functions): identical bits.  GPRC_TRSV=steps runs the per-step launches instead.
int launch_trsv(hipStream_t s, const double* packed, const double* inv, int64_t n_pad, double* b, int transpose, double* work) {
  const int P = (int)(n_pad / NB);
  if (!inv || !work) { set_error("trsv: the explicit diagonal inverses and the work buffer are required"); return GPRC_ERR_ARG; }
  ProfScope ps(s, PK_TRSV, (double)n_pad * n_pad, 8.0 * 0.5 * n_pad * n_pad);
  TrsvGate* gates = trsv_gates(work, n_pad);
  GPRC_HIP(hipMemsetAsync(gates, 0, sizeof(TrsvGate) * (size_t)P, s));
  static const bool steps = [] { const char* e = std::getenv("GPRC_TRSV"); return e && std::strcmp(e, "steps") == 0; }();
  const int f128 = trsv_far128(n_pad);
  if (!transpose) {
    if (steps) {
      for (int p = 0; p < P; ++p) {
        const int64_t below = n_pad - (int64_t)(p + 1) * NB;
        hipLaunchKernelGGL(trsv_fwd_kernel<false>, dim3(TRSV_DIAG_WGS + trsv_fwd_row_wgs(below, f128)), dim3(512), 0, s, packed, inv, n_pad, p, b, work, gates, f128);
      }
    } else {
      hipLaunchKernelGGL(trsv_fwd_kernel<false>, dim3(TRSV_DIAG_WGS), dim3(512), 0, s, packed, inv, n_pad, 0, b, work, gates, f128);
      for (int p = 0; p + 1 < P; ++p) {
        const int64_t behind = n_pad - (int64_t)(p + 2) * NB;
        hipLaunchKernelGGL(trsv_fwd_kernel<true>, dim3((unsigned)(TRSV_NEAR_FWD + TRSV_DIAG_WGS + behind / (f128 ? 128 : TRSV_ROWS))), dim3(512), 0, s, packed, inv,
                           n_pad, p, b, work, gates, f128);
      }
    }
  } else {
    if (steps) {
      for (int p = P - 1; p >= 0; --p)
        hipLaunchKernelGGL(trsv_bwd_kernel<false>, dim3((unsigned)(TRSV_DIAG_WGS + p * (NB / TRSV_COLS))), dim3(512), 0, s, packed, inv, n_pad, p, b, work, gates);
    } else {
      hipLaunchKernelGGL(trsv_bwd_kernel<false>, dim3(TRSV_DIAG_WGS), dim3(512), 0, s, packed, inv, n_pad, P - 1, b, work, gates);
      for (int p = P - 1; p >= 1; --p)
        hipLaunchKernelGGL(trsv_bwd_kernel<true>, dim3((unsigned)(TRSV_NEAR_BWD + TRSV_DIAG_WGS + (p - 1) * (NB / TRSV_COLS))), dim3(512), 0, s, packed, inv, n_pad, p,
                           b, work, gates);
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

int launch_sum_partials(hipStream_t s, const double* part, int64_t nparts, int64_t stride, int64_t rows, const double* minuend, double* out) {
  if (rows <= 0) return 0;
  ProfScope ps(s, PK_ROWREDUCE, (double)rows * nparts, 8.0 * rows * nparts);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(blocks(rows, 256)), dim3(256), 0, s, part, nparts, stride, rows, minuend, out);
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
int launch_gpc_class_prob(hipStream_t s, const double* fs, const double* vf, double* out, int64_t n) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gpc_class_prob_kernel, dim3(blocks(n, 64)), dim3(64), 0, s, fs, vf, out, n);
  GPRC_LAUNCH_CHECK();
  return 0;
}

}  // namespace gprc
