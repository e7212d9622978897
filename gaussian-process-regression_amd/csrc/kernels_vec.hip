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

// ---- triangular solves with the packed factor, one NB-wide panel per step -------------------------
// Forward  (L x = b):  for p = 0..P-1:  x_p = L_pp^-1 b_p  (one workgroup, using the 128-block inverses),
//                      then b[below] -= L[below, panel p] * x_p  (one wide GEMV, thread groups per row).
// Backward (L^T x = z): for p = P-1..0: x_p = L_pp^-T z_p, then z[q] -= L[panel-p rows, panel q cols]^T x_p for q < p.
// Every reduction has a fixed order: results are bitwise reproducible.

__device__ __forceinline__ double wave_sum(double s) {
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  return s;
}

// x_p = L_pp^-1 b_p in place.  1024 threads: (i = t & 127) x (ty = t >> 7, eight column groups).
// One workgroup, latency-bound: every phase first issues ALL its global loads (fixed trip counts, fully unrolled --
// 16 J + 16 of them in flight per thread) and only then runs the FMA chain, in the original summation order.  With the
// loads issued one per loop iteration the kernel spent ~48 dependent HBM round trips per solve (89 us).
template <int J>
__device__ __forceinline__ void trsv_fwd_phase(const double* pan, int64_t ld, const double* W, double* z, double (*red)[128], int i, int ty) {
  if constexpr (J > 0) {  // z_J -= L[J-th row block, columns 0 .. 128 J) * x[0 .. 128 J)
    const double* Lr = pan + J * 128 + i;
    double lv[16 * J];
#pragma unroll
    for (int k = 0; k < 16 * J; ++k) lv[k] = Lr[(int64_t)(ty + 8 * k) * ld];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16 * J; ++k) s = fma(lv[k], z[ty + 8 * k], s);
    red[ty][i] = s;
    __syncthreads();
    if (ty == 0) {
      double v = z[J * 128 + i];
      for (int g = 0; g < 8; ++g) v -= red[g][i];
      z[J * 128 + i] = v;
    }
    __syncthreads();
  }
  double wv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int c = ty + 8 * k;
    wv[k] = (c <= i) ? W[i + c * 128] : 0.0;
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int c = ty + 8 * k;
    if (c <= i) s = fma(wv[k], z[J * 128 + c], s);
  }
  red[ty][i] = s;
  __syncthreads();
  if (ty == 0) {
    double v = 0.0;
    for (int g = 0; g < 8; ++g) v += red[g][i];
    z[J * 128 + i] = v;
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void trsv_diag_fwd(const double* packed, const double* winv, int64_t n_pad, int p, double* b) {
  __shared__ double z[NB];
  __shared__ double red[8][128];
  static_assert(TPP == 4, "trsv_diag_fwd is written for four 128-blocks per panel");
  const int t = threadIdx.x, i = t & 127, ty = t >> 7;
  const int64_t ld = panel_ld(n_pad, p);
  const double* pan = packed + panel_offset(n_pad, p);
  const double* W = winv + (int64_t)p * TPP * 128 * 128;
  double* bp = b + (int64_t)p * NB;
  if (t < NB) z[t] = bp[t];
  __syncthreads();
  trsv_fwd_phase<0>(pan, ld, W, z, red, i, ty);
  trsv_fwd_phase<1>(pan, ld, W + 128 * 128, z, red, i, ty);
  trsv_fwd_phase<2>(pan, ld, W + 2 * 128 * 128, z, red, i, ty);
  trsv_fwd_phase<3>(pan, ld, W + 3 * 128 * 128, z, red, i, ty);
  if (t < NB) bp[t] = z[t];
}

// One row's share of L[row, panel p columns] * x_p for column group g (NB/4 columns): four interleaved fma chains,
// combined (s0 + s1) + (s2 + s3).  The per-panel step kernel and the single-launch flag kernel both call this, so the
// two forms of the solve produce the same bits.
__device__ __forceinline__ double gemv_group_partial(const double* Lr, int64_t ld, const double* xg) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
  for (int c = 0; c < NB / 4; c += 4) {
    s0 = fma(Lr[(int64_t)c * ld], xg[c], s0);
    s1 = fma(Lr[(int64_t)(c + 1) * ld], xg[c + 1], s1);
    s2 = fma(Lr[(int64_t)(c + 2) * ld], xg[c + 2], s2);
    s3 = fma(Lr[(int64_t)(c + 3) * ld], xg[c + 3], s3);
  }
  return (s0 + s1) + (s2 + s3);
}

// b[r] -= sum_c L[r, p*NB + c] * x_p[c] for the rows below panel p.  512 threads = 128 rows x 4 column groups.
__global__ __launch_bounds__(512) void trsv_gemv_below(const double* packed, int64_t n_pad, int p, double* b) {
  __shared__ double xs[NB];
  __shared__ double red[4][128];
  const int t = threadIdx.x, i = t & 127, g = t >> 7;
  for (int c = t; c < NB; c += 512) xs[c] = b[(int64_t)p * NB + c];
  __syncthreads();
  const int64_t ld = panel_ld(n_pad, p);
  const int64_t row = (int64_t)(p + 1) * NB + (int64_t)blockIdx.x * 128 + i;
  const double* Lr = packed + panel_offset(n_pad, p) + (row - (int64_t)p * NB) + (int64_t)(g * (NB / 4)) * ld;
  red[g][i] = gemv_group_partial(Lr, ld, xs + g * (NB / 4));
  __syncthreads();
  if (g == 0) b[row] -= (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// x_p = L_pp^-T z_p in place.  16 waves; a wave owns columns w, w+16, ... and reduces over rows with its lanes.
// One backward phase: block J of the panel's diagonal (16 waves; wave w handles columns w, w + 16, ...; lanes stride the
// rows).  All loads of the eight columns a wave owns are issued before the first reduction.
template <int J>
__device__ __forceinline__ void trsv_bwd_phase(const double* pan, int64_t ld, const double* W, double* z, double* v, int lane, int w) {
  constexpr int NR = (TPP - 1 - J) * 2;  // rows below block J inside the panel, 64 per step
  {  // v[c] = z_J[c] - sum_{r >= 128 (J+1)} L[r, 128 J + c] * x[r]
    double lv[8][NR > 0 ? NR : 1];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const double* col = pan + (int64_t)(J * 128 + w + 16 * k) * ld;
#pragma unroll
      for (int q = 0; q < NR; ++q) lv[k][q] = col[(J + 1) * 128 + lane + 64 * q];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < NR; ++q) s = fma(lv[k][q], z[(J + 1) * 128 + lane + 64 * q], s);
      s = wave_sum(s);
      if (lane == 0) v[w + 16 * k] = z[J * 128 + w + 16 * k] - s;
    }
  }
  __syncthreads();
  {  // x_J[c'] = sum_{c >= c'} W_J[c, c'] * v[c]
    double wv[8][2];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int cp = w + 16 * k;
      const double* wc = W + cp * 128;
#pragma unroll
      for (int q = 0; q < 2; ++q) wv[k][q] = (lane + 64 * q >= cp) ? wc[lane + 64 * q] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int cp = w + 16 * k;
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (lane + 64 * q >= cp) s = fma(wv[k][q], v[lane + 64 * q], s);
      s = wave_sum(s);
      if (lane == 0) z[J * 128 + cp] = s;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void trsv_diag_bwd(const double* packed, const double* winv, int64_t n_pad, int p, double* x) {
  __shared__ double z[NB];
  __shared__ double v[128];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int64_t ld = panel_ld(n_pad, p);
  const double* pan = packed + panel_offset(n_pad, p);
  const double* W = winv + (int64_t)p * TPP * 128 * 128;
  double* xp = x + (int64_t)p * NB;
  if (t < NB) z[t] = xp[t];
  __syncthreads();
  trsv_bwd_phase<3>(pan, ld, W + 3 * 128 * 128, z, v, lane, w);
  trsv_bwd_phase<2>(pan, ld, W + 2 * 128 * 128, z, v, lane, w);
  trsv_bwd_phase<1>(pan, ld, W + 128 * 128, z, v, lane, w);
  trsv_bwd_phase<0>(pan, ld, W, z, v, lane, w);
  if (t < NB) xp[t] = z[t];
}

// One column's dot product with x_p over the NB rows of panel p (lanes stride the rows, fixed fma chain + wave tree):
// shared by the per-panel step kernel and the flag kernel (same bits).
__device__ __forceinline__ double gemvt_column_dot(const double* col, const double* xs, int lane) {
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < NB; r += 64) s = fma(col[r + lane], xs[r + lane], s);
  return wave_sum(s);
}

// z[q*NB + c] -= sum_r L[p*NB + r, q*NB + c] * x_p[r] for every earlier panel q < p.  Block = (q, 32 columns).
__global__ __launch_bounds__(256) void trsv_gemvt_above(const double* packed, int64_t n_pad, int p, double* z) {
  __shared__ double xs[NB];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int c = t; c < NB; c += 256) xs[c] = z[(int64_t)p * NB + c];
  __syncthreads();
  const int q = blockIdx.x / (NB / 32), cg = blockIdx.x % (NB / 32);
  const int64_t ld = panel_ld(n_pad, q);
  const double* blk = packed + panel_offset(n_pad, q) + (int64_t)(p - q) * NB;  // rows of panel p inside panel q
#pragma unroll 2
  for (int k = 0; k < 8; ++k) {
    const int c = cg * 32 + w * 8 + k;
    const double s = gemvt_column_dot(blk + (int64_t)c * ld, xs, lane);
    if (lane == 0) z[(int64_t)q * NB + c] -= s;
  }
}

// ------------------------------------------------------------------------------------------------
// Whole solves, ONE launch per panel (GPRC_TRSV=chain; NOT the default -- measured slower, see launch_trsv): the kernel that subtracts panel p's contribution from the
// rest of the right-hand side ALSO solves the diagonal block of the next panel -- in the workgroup that owns that panel's 512
// rows / columns, right after it has updated them.  The per-panel form above needs two dependent launches per panel (diagonal
// block, then the wide product: 70 us, of which the one-workgroup diagonal kernel is half); here the diagonal solve of panel
// p +- 1 rides on the product of panel p while the other workgroups are still streaming their rows.  Every row / column goes
// through gemv_group_partial / gemvt_column_dot and the trsv_*_phase code in the same order as in the per-panel kernels:
// identical bits (tests/test_gpu_device_level.py).  (gprc_dev_trsv_step keeps the per-panel kernels: in the multi-rank sweep the
// next panel does not exist yet when a step runs.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void trsv_fwd_chain_kernel(const double* packed, const double* winv, int64_t n_pad, int p, double* b) {
  __shared__ double xs[NB];
  __shared__ double z[NB];
  __shared__ double red[4][NB];      // workgroup 0: [group][row of the next panel]; others: [sub * 4 + g][i] in the first 8 x 128
  static_assert(TPP == 4, "written for four 128-blocks per panel");
  const int t = threadIdx.x;
  if (t < NB) xs[t] = b[(int64_t)p * NB + t];
  __syncthreads();
  const int64_t ld = panel_ld(n_pad, p);
  const double* pan = packed + panel_offset(n_pad, p);
  if (blockIdx.x == 0) {             // the 512 rows of panel p + 1, then that panel's diagonal block
    const int r = t & (NB - 1), h = t >> 9;
    const int64_t row = (int64_t)(p + 1) * NB + r;
    const double* Lr = pan + (row - (int64_t)p * NB);
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
      const int g = 2 * h + gg;
      red[g][r] = gemv_group_partial(Lr + (int64_t)(g * (NB / 4)) * ld, ld, xs + g * (NB / 4));
    }
    __syncthreads();
    if (t < NB) z[t] = b[row] - ((red[0][t] + red[1][t]) + (red[2][t] + red[3][t]));   // (row == (p+1) NB + t for t < NB)
    __syncthreads();
    const int64_t ld1 = panel_ld(n_pad, p + 1);
    const double* pan1 = packed + panel_offset(n_pad, p + 1);
    const double* W = winv + (int64_t)(p + 1) * TPP * 128 * 128;
    double (*red8)[128] = reinterpret_cast<double (*)[128]>(&red[0][0]);
    const int i = t & 127, ty = t >> 7;
    trsv_fwd_phase<0>(pan1, ld1, W, z, red8, i, ty);
    trsv_fwd_phase<1>(pan1, ld1, W + 128 * 128, z, red8, i, ty);
    trsv_fwd_phase<2>(pan1, ld1, W + 2 * 128 * 128, z, red8, i, ty);
    trsv_fwd_phase<3>(pan1, ld1, W + 3 * 128 * 128, z, red8, i, ty);
    if (t < NB) b[(int64_t)(p + 1) * NB + t] = z[t];
    return;
  }
  // rows behind panel p + 1: 256 per workgroup, the thread layout of trsv_gemv_below twice
  const int sub = t >> 9, i = t & 127, g = (t >> 7) & 3;
  const int64_t row = (int64_t)(p + 2) * NB + (int64_t)(blockIdx.x - 1) * 256 + sub * 128 + i;
  const double* Lr = pan + (row - (int64_t)p * NB) + (int64_t)(g * (NB / 4)) * ld;
  double (*red8)[128] = reinterpret_cast<double (*)[128]>(&red[0][0]);
  red8[sub * 4 + g][i] = gemv_group_partial(Lr, ld, xs + g * (NB / 4));
  __syncthreads();
  if (g == 0) b[row] -= (red8[sub * 4][i] + red8[sub * 4 + 1][i]) + (red8[sub * 4 + 2][i] + red8[sub * 4 + 3][i]);
}

__global__ __launch_bounds__(1024) void trsv_bwd_chain_kernel(const double* packed, const double* winv, int64_t n_pad, int p, double* x) {
  __shared__ double xs[NB];
  __shared__ double z[NB];
  __shared__ double v[128];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t < NB) xs[t] = x[(int64_t)p * NB + t];
  __syncthreads();
  if (blockIdx.x == 0) {             // the 512 columns of panel p - 1, then that panel's diagonal block
    const int q = p - 1;
    const int64_t ldq = panel_ld(n_pad, q);
    const double* panq = packed + panel_offset(n_pad, q);
    const double* blk = panq + NB;   // rows of panel p inside panel q = p - 1
    if (t < NB) z[t] = x[(int64_t)q * NB + t];
    __syncthreads();
    for (int k = 0; k < NB / 16; ++k) {          // wave w: columns w * 32 .. w * 32 + 31
      const int c = w * (NB / 16) + k;
      const double s = gemvt_column_dot(blk + (int64_t)c * ldq, xs, lane);
      if (lane == 0) z[c] -= s;
    }
    __syncthreads();
    const double* W = winv + (int64_t)q * TPP * 128 * 128;
    trsv_bwd_phase<3>(panq, ldq, W + 3 * 128 * 128, z, v, lane, w);
    trsv_bwd_phase<2>(panq, ldq, W + 2 * 128 * 128, z, v, lane, w);
    trsv_bwd_phase<1>(panq, ldq, W + 128 * 128, z, v, lane, w);
    trsv_bwd_phase<0>(panq, ldq, W, z, v, lane, w);
    if (t < NB) x[(int64_t)q * NB + t] = z[t];
    return;
  }
  // panels q < p - 1: 128 columns per workgroup, 8 per wave
  const int cb = blockIdx.x - 1, q = cb / TPP, c0 = (cb % TPP) * 128;
  const int64_t ldq = panel_ld(n_pad, q);
  const double* blk = packed + panel_offset(n_pad, q) + (int64_t)(p - q) * NB;
  for (int k = 0; k < 8; ++k) {
    const int c = c0 + w * 8 + k;
    const double s = gemvt_column_dot(blk + (int64_t)c * ldq, xs, lane);
    if (lane == 0) x[(int64_t)q * NB + c] -= s;
  }
}

// ------------------------------------------------------------------------------------------------
// The whole solve in ONE launch (forward and backward): "sync-free" triangular solve over 256-row strips.
//
// The per-panel form above costs two launches per 512 columns, 2 x 128 dependent launches at n = 65536: 9 ms per solve,
// all of it launch and drain latency.  Here every strip (half a panel) is a 1024-thread workgroup that
//   * takes its strip number from a ticket counter (strip order = start order: a workgroup only ever waits for strips
//     whose workgroups are already running -- no deadlock whatever the dispatch order or residency),
//   * walks through the panels that precede it (forward) / follow it (backward), waiting on ONE monotone counter
//     `done` (strips complete strictly in order) and subtracting each panel's contribution from its right-hand side
//     with gemv_group_partial / gemvt_column_dot -- the SAME per-row / per-column arithmetic, in the same order, as
//     the per-panel kernels, so both forms give identical bits (tests/test_gpu_device_level.py),
//   * solves its half of the panel's diagonal block with the trsv_*_phase code of the per-panel form,
//   * publishes its part of x and bumps `done`.
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): producer = plain stores, every storing wave's
// s_waitcnt vmcnt(0), workgroup barrier, one lane's agent-scope release fence + vmcnt(0), relaxed agent-scope store of
// the counter; consumer = one lane polls the counter with relaxed agent-scope loads (s_sleep between polls), agent-scope
// acquire fence + vmcnt(0), workgroup barrier, plain loads.
// While a strip waits for the panel right in front of it, it runs that panel's product once with whatever x is there
// and throws the result away: the L block is then in this XCD's L2 when the real product is on the critical path.
// ------------------------------------------------------------------------------------------------
struct TrsvSync { int ticket; int done; int failed; int pad; };

__device__ __forceinline__ void strip_wait(TrsvSync* sy, int need, int& seen, int* sh) {
  if (seen >= need) return;                      // uniform over the workgroup: `seen` is the same in every thread
  if (threadIdx.x == 0) {
    int d, spins = 0;
    while ((d = __hip_atomic_load(&sy->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
      __builtin_amdgcn_s_sleep(2);
      // Exit condition every wave reaches: a predecessor that never publishes (a fault elsewhere) must not leave this
      // workgroup spinning on the GPU for ever.  ~2^25 polls is tens of seconds -- far beyond any legitimate wait; the
      // counter is then pushed past every strip so the whole grid drains (the solve's result is garbage and the
      // `failed` word says so).
      if (++spins > (1 << 25)) {
        __hip_atomic_store(&sy->failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sy->done, 1 << 30, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        d = 1 << 30;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *sh = d;
  }
  __syncthreads();
  seen = *sh;
  __syncthreads();
}

__device__ __forceinline__ void strip_publish(TrsvSync* sy, int value) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores of x
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_max(&sy->done, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // max: never lowers a counter the bail-out pushed past the end
  }
}

__global__ __launch_bounds__(1024) void trsv_fwd_flag_kernel(const double* packed, const double* winv, int64_t n_pad, double* b, TrsvSync* sy) {
  __shared__ double z[NB];          // the panel's right-hand side / solution as the diagonal phases see it
  __shared__ double zs[256];        // this strip's 256 entries while the earlier panels are being subtracted
  __shared__ double xs[NB];         // x of the panel being applied
  __shared__ double red[8][128];
  __shared__ int sh;
  static_assert(TPP == 4, "the flag solve is written for four 128-blocks per panel");
  const int t = threadIdx.x;
  if (t == 0) sh = atomicAdd(&sy->ticket, 1);
  __syncthreads();
  const int q = sh;                 // strip number: panel ps, half h
  __syncthreads();
  const int ps = q >> 1, h = q & 1;
  const int64_t row0 = (int64_t)ps * NB + h * 256;
  if (t < 256) zs[t] = b[row0 + t];
  int seen = 0;
  // rows of this strip: sub-strip (t >> 9), row i, column group g -- the thread layout of trsv_gemv_below, twice
  const int sub = t >> 9, i = t & 127, g = (t >> 7) & 3;
  const int64_t row = row0 + sub * 128 + i;
  __syncthreads();
  for (int p = 0; p < ps; ++p) {
    const int64_t ld = panel_ld(n_pad, p);
    const double* Lr = packed + panel_offset(n_pad, p) + (row - (int64_t)p * NB) + (int64_t)(g * (NB / 4)) * ld;
    if (seen < 2 * p + 2) {         // will have to wait: warm this XCD's L2 with the block first (result discarded)
      const double warm = gemv_group_partial(Lr, ld, xs + g * (NB / 4));
      if (warm == 1.2345e-300) red[g + 4][i] = warm;   // never true in practice; keeps the loads alive
      strip_wait(sy, 2 * p + 2, seen, &sh);
    }
    if (t < NB) xs[t] = b[(int64_t)p * NB + t];
    __syncthreads();
    red[sub * 4 + g][i] = gemv_group_partial(Lr, ld, xs + g * (NB / 4));
    __syncthreads();
    if (g == 0) zs[sub * 128 + i] -= (red[sub * 4][i] + red[sub * 4 + 1][i]) + (red[sub * 4 + 2][i] + red[sub * 4 + 3][i]);
    __syncthreads();
  }
  // the diagonal block: phases 2h and 2h + 1 of trsv_diag_fwd; the second strip of a panel needs the first one's x
  const int64_t ld = panel_ld(n_pad, ps);
  const double* pan = packed + panel_offset(n_pad, ps);
  const double* W = winv + (int64_t)ps * TPP * 128 * 128;
  const int di = t & 127, ty = t >> 7;
  if (h == 1) {
    strip_wait(sy, q, seen, &sh);
    if (t < 256) z[t] = b[(int64_t)ps * NB + t];
  }
  if (t < 256) z[h * 256 + t] = zs[t];
  __syncthreads();
  if (h == 0) {
    trsv_fwd_phase<0>(pan, ld, W, z, red, di, ty);
    trsv_fwd_phase<1>(pan, ld, W + 128 * 128, z, red, di, ty);
  } else {
    trsv_fwd_phase<2>(pan, ld, W + 2 * 128 * 128, z, red, di, ty);
    trsv_fwd_phase<3>(pan, ld, W + 3 * 128 * 128, z, red, di, ty);
  }
  if (t < 256) b[row0 + t] = z[h * 256 + t];
  strip_publish(sy, q + 1);
}

__global__ __launch_bounds__(1024) void trsv_bwd_flag_kernel(const double* packed, const double* winv, int64_t n_pad, double* x, TrsvSync* sy) {
  __shared__ double z[NB];
  __shared__ double zs[256];
  __shared__ double xs[NB];
  __shared__ double v[128];
  __shared__ int sh;
  const int P = (int)(n_pad / NB);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) sh = atomicAdd(&sy->ticket, 1);
  __syncthreads();
  const int q = sh;                 // completion order: the LAST strip of the matrix is q = 0
  __syncthreads();
  const int ps = P - 1 - (q >> 1), h = 1 - (q & 1);
  const int64_t col0 = (int64_t)ps * NB + h * 256;
  if (t < 256) zs[t] = x[col0 + t];
  int seen = 0;
  const int64_t ldq = panel_ld(n_pad, ps);
  const double* mine = packed + panel_offset(n_pad, ps) + (int64_t)(h * 256) * ldq;   // column h*256 of my panel, row ps*NB
  __syncthreads();
  for (int p = P - 1; p > ps; --p) {          // panels behind me, in the order they complete
    const double* blk = mine + (int64_t)(p - ps) * NB;   // rows of panel p
    const int need = 2 * (P - p);
    if (seen < need) {
      double warm = 0.0;
      for (int k = 0; k < 16; ++k) warm += gemvt_column_dot(blk + (int64_t)(w * 16 + k) * ldq, xs, lane);
      if (warm == 1.2345e-300) v[lane] = warm;
      strip_wait(sy, need, seen, &sh);
    }
    if (t < NB) xs[t] = x[(int64_t)p * NB + t];
    __syncthreads();
    for (int k = 0; k < 16; ++k) {             // wave w: columns w*16 .. w*16 + 15 of the strip
      const int c = w * 16 + k;
      const double s = gemvt_column_dot(blk + (int64_t)c * ldq, xs, lane);
      if (lane == 0) zs[c] -= s;
    }
    __syncthreads();
  }
  const double* pan = packed + panel_offset(n_pad, ps);
  const double* W = winv + (int64_t)ps * TPP * 128 * 128;
  if (h == 0) {
    strip_wait(sy, q, seen, &sh);
    if (t < 256) z[256 + t] = x[(int64_t)ps * NB + 256 + t];
  }
  if (t < 256) z[h * 256 + t] = zs[t];
  __syncthreads();
  if (h == 1) {
    trsv_bwd_phase<3>(pan, ldq, W + 3 * 128 * 128, z, v, lane, w);
    trsv_bwd_phase<2>(pan, ldq, W + 2 * 128 * 128, z, v, lane, w);
  } else {
    trsv_bwd_phase<1>(pan, ldq, W + 128 * 128, z, v, lane, w);
    trsv_bwd_phase<0>(pan, ldq, W, z, v, lane, w);
  }
  if (t < 256) x[col0 + t] = z[h * 256 + t];
  strip_publish(sy, q + 1);
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

// one panel step of the solve: the diagonal block of panel p, then its contribution to the rest of the right-hand side
int launch_trsv_step(hipStream_t s, const double* packed, const double* winv, int64_t n_pad, double* b, int transpose, int p) {
  const int P = (int)(n_pad / NB);
  if (p < 0 || p >= P) { set_error("trsv_step: panel out of range"); return GPRC_ERR_ARG; }
  if (!transpose) {
    hipLaunchKernelGGL(trsv_diag_fwd, dim3(1), dim3(1024), 0, s, packed, winv, n_pad, p, b);
    const int64_t below = n_pad - (int64_t)(p + 1) * NB;
    if (below > 0) hipLaunchKernelGGL(trsv_gemv_below, dim3((unsigned)(below / 128)), dim3(512), 0, s, packed, n_pad, p, b);
  } else {
    hipLaunchKernelGGL(trsv_diag_bwd, dim3(1), dim3(1024), 0, s, packed, winv, n_pad, p, b);
    if (p > 0) hipLaunchKernelGGL(trsv_gemvt_above, dim3((unsigned)(p * (NB / 32))), dim3(256), 0, s, packed, n_pad, p, b);
  }
  GPRC_LAUNCH_CHECK();
  return 0;
}

// work: gprc_trsv_work_size(n_pad) doubles; its first bytes hold the ticket / progress counters of the flag kernels.
// Default: the two-launch per-panel form.  GPRC_TRSV=chain: one launch per panel (trsv_*_chain_kernel; slower, see below).
// GPRC_TRSV=flag selects the single-launch strip kernels (same bits): measured SLOWER on
// MI355X -- 12.7 ms against 9.1 ms per solve at n = 65536, 1.33 against 0.80 ms at n = 8192 -- because a strip's critical
// path (the 256 x 512 block next to the diagonal + two diagonal phases, ~1.4 MB) is read by ONE compute unit at its
// ~100 GB/s, where the per-panel form spreads the same bytes over four.
int launch_trsv(hipStream_t s, const double* packed, const double* winv, int64_t n_pad, double* b, int transpose, double* work) {
  const int P = (int)(n_pad / NB);
  ProfScope ps(s, PK_TRSV, (double)n_pad * n_pad, 8.0 * 0.5 * n_pad * n_pad);
  static const bool flag = [] { const char* e = std::getenv("GPRC_TRSV"); return e && std::strcmp(e, "flag") == 0; }();
  if (flag && work) {
    TrsvSync* sy = reinterpret_cast<TrsvSync*>(work);
    GPRC_HIP(hipMemsetAsync(sy, 0, sizeof(TrsvSync), s));
    if (!transpose) hipLaunchKernelGGL(trsv_fwd_flag_kernel, dim3((unsigned)(2 * P)), dim3(1024), 0, s, packed, winv, n_pad, b, sy);
    else hipLaunchKernelGGL(trsv_bwd_flag_kernel, dim3((unsigned)(2 * P)), dim3(1024), 0, s, packed, winv, n_pad, b, sy);
    GPRC_LAUNCH_CHECK();
    return 0;
  }
  static const bool chain = [] { const char* e = std::getenv("GPRC_TRSV"); return e && std::strcmp(e, "chain") == 0; }();
  if (!chain) {  // default: two launches per panel (the form gprc_dev_trsv_step exposes)
    if (!transpose)
      for (int p = 0; p < P; ++p) GPRC_TRY(launch_trsv_step(s, packed, winv, n_pad, b, 0, p));
    else
      for (int p = P - 1; p >= 0; --p) GPRC_TRY(launch_trsv_step(s, packed, winv, n_pad, b, 1, p));
    return 0;
  }
  // GPRC_TRSV=chain: one launch per panel -- the product of panel p carries the diagonal solve of the next panel.  Same bits,
  // measured SLOWER (0.80 -> 1.09 ms per solve at n = 8192, 9.1 -> 11.6 ms at n = 65536): the next panel's 512 x 512 product
  // (2 MB) then goes through ONE compute unit in front of the diagonal solve, where the two-launch form spreads it over four.
  if (!transpose) {
    hipLaunchKernelGGL(trsv_diag_fwd, dim3(1), dim3(1024), 0, s, packed, winv, n_pad, 0, b);
    for (int p = 0; p + 1 < P; ++p) {
      const int64_t behind = n_pad - (int64_t)(p + 2) * NB;   // rows behind panel p + 1
      hipLaunchKernelGGL(trsv_fwd_chain_kernel, dim3((unsigned)(1 + behind / 256)), dim3(1024), 0, s, packed, winv, n_pad, p, b);
    }
  } else {
    hipLaunchKernelGGL(trsv_diag_bwd, dim3(1), dim3(1024), 0, s, packed, winv, n_pad, P - 1, b);
    for (int p = P - 1; p >= 1; --p)
      hipLaunchKernelGGL(trsv_bwd_chain_kernel, dim3((unsigned)(1 + (p - 1) * TPP)), dim3(1024), 0, s, packed, winv, n_pad, p, b);
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
