// kernels_chol.hip -- blocked right-looking fp64 Cholesky pieces for gfx950 (MI355X).
//
// Replaces L <- t(chol(K + noise * diag(n)))  (reference R/GPRclass.R:142, LAPACK dpotrf) and, through
// the same GEMM tile, v <- solve(L, K_star) (R/GPRclass.R:162, which the reference runs as a general
// pivoted dgesv).  The pieces:
//   potf2_blocked_body  one workgroup factors a 128x128 diagonal block entirely in LDS and also forms
//                     its inverse (so every panel / right-hand-side solve below is a GEMM);
//   gemm tile core    128x128 output tile per 256-thread workgroup, 4 waves x (64x64) of
//                     v_mfma_f64_16x16x4_f64, A/B strips staged through double-buffered LDS;
//   wrappers          panel solve (X := X * Winv^T, in place), in-panel / general C -= A*B^T, and the
//                     trailing update over the packed block-column layout (lower tiles only);
//   panel_fused_kernel / panel_service_kernel   a whole panel's dependent chain in one launch / every panel's in one
//                     persistent launch (the factor service), with the caller's-stream kernels that go with it.
// The trailing update is the dominant kernel of the whole path: n^3/3 of the fit and n^2 n* of the
// predict go through gemm_tile_128().
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <utility>

#include "gprc_internal.h"

namespace gprc {

typedef double double4_t __attribute__((ext_vector_type(4)));

namespace {

// ------------------------------------------------------------------------------------------------
// Diagonal block: Cholesky + inverse of a 128 x 128 block in one sweep by one workgroup.
// ------------------------------------------------------------------------------------------------
constexpr int PB = 128;

// ------------------------------------------------------------------------------------------------
// Blocked diagonal-block kernel (default): the same factor + inverse, 16 columns at a time.
// The 128 x 128 block lives in LDS (leading dimension 144: MFMA operand reads conflict-free).  Per 16-column step:
//   A  wave 0 factors the 16 x 16 diagonal sub-block and inverts it with a register-resident scalar sweep (diag16:
//      16 sequential pivots, a matrix row per lane, operands exchanged inside the 16-lane row by DPP row_newbcast; sqrt /
//      reciprocal from a Newton-refined v_rsq_f64 -- ~60 dependent cycles instead of the ~600 of the library
//      sqrt + division);
//   B  panel rows below: X_I = A_I * Wd^T, and row s of the inverse: X_sJ = Wd * Y_sJ   (4 MFMAs per 16x16 block);
//   C  Cholesky trailing blocks C_IJ -= X_I X_J^T and inverse blocks Y_IJ -= L_Is X_sJ  (4 MFMAs per block); wave 0
//      updates the next diagonal block first and runs phase A of step s+1 beside the other waves' blocks,
// one barrier after B and one after C.  L is kept in the lower triangle, the (unscaled-free) inverse transposed in the
// upper triangle, its diagonal in a side array -- the layout of the output.  The sequential depth drops from 128
// whole-workgroup steps to 128 single-wave steps on 16-row data.
// ------------------------------------------------------------------------------------------------
constexpr int BLD = 144;
constexpr int PB_SMEM_DOUBLES = PB * BLD + 512 + PB;  // S, 2 x Wd, Wdiag

// sqrt(d) and 1/sqrt(d) from v_rsq_f64 + two Newton steps (+ one correction of the root)
__device__ __forceinline__ void sqrt_rsqrt(double d, double& root, double& rinv) {
  double r = __builtin_amdgcn_rsq(d);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double e = fma(-(d * r), r, 1.0);  // 1 - d r^2
    r = fma(0.5 * r, e, r);
  }
  double l = d * r;
  l = fma(0.5 * r, fma(-l, l, d), l);
  root = l;
  rinv = r;
}

// Phase A: one wave factors the 16x16 block at D (LDS, ld BLD; lower triangle valid) in place, writes its inverse dense to
// Wd[16][16] (column-major), transposed-strict-lower into D's upper triangle and the diagonal to wdiag.
//
// Outer-product Cholesky with the rank-1 update on the matrix core.  The wave holds the full SYMMETRIC working block A and a
// unit-lower-triangular V (Gauss-Jordan on [A | I]: W = diag(rinv) V at the end) in the accumulator layout of
// v_mfma_f64_16x16x4: lane (q = lane >> 4, r = lane & 15), register rr <-> element (x = q + 4 rr, y = r).  Row J of either matrix
// is then ONE register (rr = J / 4) of ONE 16-lane row (q = J % 4), indexed by the lane -- exactly the shape of an MFMA operand
// in k-slot q.  Pivot J:
//   u = row J of A (= column J: the block stays bitwise symmetric, l_x l_y and l_y l_x are the same product)
//   d = u[J] (DPP row_newbcast inside the 16-lane row), rinv = 1/sqrt(d) (v_rsq_f64 + two Newton steps)
//   l = u rinv below the diagonal, 0 elsewhere and in the other three lane rows;  m = -l rinv
//   A -= l l^T      one MFMA: both operands are l, k-slot J % 4, the other slots zero
//   V += m v_J^T    one MFMA: v_J = row J of V (V starts as I, so column J receives m and V[J][J] stays 1)
// The dependent chain of a column is DPP mov -> rsqrt -> one multiply -> one MFMA (~200 cycles: tools/microbench/dp_latency.hip has
// the instruction latencies), all 64 lanes work, and a pivot is ~35 instructions.  (An f64 MFMA runs on the SIMD's double-precision
// lanes: no VALU instruction of the wave issues beside it, so the second MFMA and the column's stores add to the chain rather than
// hide behind it -- ~400 cycles per pivot measured.)  (Round 2's sweep kept a
// row per lane quadruple and exchanged six operands per pivot by ds_bpermute: ~85 instructions and ~600 cycles per pivot, 3.7 us per
// sweep; a row-per-lane form with v_fmac_f64_dpp -- DPP on 64-bit operands issues at ~13 cycles -- reached 3.3 us.)
// No exec-masked branch and no store inside the sweep: a column's entries of L stay in a register of the lane row that computed
// them, a non-positive pivot is only noted for one atomic after the sweep.
template <int SRC>
__device__ __forceinline__ double row_bcast(double v) {   // lane SRC's v, in every lane of the same 16-lane row
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + SRC, 0xf, 0xf, true);   // DPP_ROW_NEWBCAST0 + SRC
}

template <int J>
__device__ __forceinline__ void diag16_pivot(double4_t& A, double4_t& V, double (&rinvs)[4], double (&lcol)[4], unsigned& bad, int q, int r) {
  constexpr int QJ = J & 3, RJ = J >> 2;
  const bool inrow = q == QJ;
  const bool below = inrow && r > J;
  // ---- the chain (a wave issues in order: the sched_barriers make the program order the schedule)
  const double u = A[RJ];                             // row J of the working block (lane row QJ)
  const double d = row_bcast<J>(u);                   // the pivot
  double rinv = __builtin_amdgcn_rsq(d);              // sqrt_rsqrt's rinv: v_rsq_f64 + two Newton steps
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double e = fma(-(d * rinv), rinv, 1.0);
    rinv = fma(0.5 * rinv, e, rinv);
  }
  const double t = u * rinv;
  const double lv = below ? t : 0.0;                  // column J of L below the diagonal; zero in the other k-slots
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (J < 15) A = __builtin_amdgcn_mfma_f64_16x16x4f64(lv, lv, A, 0, 0, 1);   // A -= l l^T
  __builtin_amdgcn_sched_barrier(0);
  // ---- behind the chain's MFMA
  if constexpr (J < 15) {
    const double mv = below ? -(t * rinv) : 0.0;
    const double vrow = inrow ? V[RJ] : 0.0;                                   // row J of V
    V = __builtin_amdgcn_mfma_f64_16x16x4f64(mv, vrow, V, 0, 0, 0);            // V += m v_J^T
  }
  lcol[RJ] = inrow ? lv : lcol[RJ];                   // column J of L below the diagonal: stored after the sweep by lane row QJ
  rinvs[RJ] = inrow ? rinv : rinvs[RJ];               // rows x = q + 4 rr of this lane: their pivots are seen by lane row q
  bad |= ((__builtin_amdgcn_fcmp(d, 0.0, 2 /* ogt */) >> (16 * QJ)) & 1ull) ? 0u : (1u << J);
  __builtin_amdgcn_sched_barrier(0);
}

template <int... J>
__device__ __forceinline__ void diag16_sweep(double4_t& A, double4_t& V, double (&rinvs)[4], double (&lcol)[4], unsigned& bad, int q, int r,
                                             std::integer_sequence<int, J...>) {
  (diag16_pivot<J>(A, V, rinvs, lcol, bad, q, r), ...);
}

__device__ __forceinline__ void diag16(double* D, double* Wd, double* wdiag, int* info, int col) {
  const int l = threadIdx.x & 63, r = l & 15, q = l >> 4;
  double4_t A, V;
  double rinvs[4] = {1.0, 1.0, 1.0, 1.0}, lcol[4] = {0.0, 0.0, 0.0, 0.0};
  unsigned bad = 0;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int x = q + 4 * rr;
    A[rr] = (x >= r) ? D[x + r * BLD] : D[r + x * BLD];   // the upper triangle mirrors the lower one
    V[rr] = (x == r) ? 1.0 : 0.0;
  }
  diag16_sweep(A, V, rinvs, lcol, bad, q, r, std::make_integer_sequence<int, 16>{});
  if (bad != 0 && l == 0) atomicCAS(info, 0, col + __builtin_ctz(bad) + 1);  // LAPACK info: first non-PD leading minor
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int x = q + 4 * rr;                        // W[x][r] = rinv_x V[x][r]
    if (r == x) {                                    // the pivot of row x is still where it was read (later updates add l_x l_y with l_x = 0)
      double ljj, rinv;
      sqrt_rsqrt(A[rr], ljj, rinv);
      D[x + x * BLD] = ljj;
    }
    if (r > x) D[r + x * BLD] = lcol[rr];            // column x of L: lane row q = x % 4 kept it
    const double w = (r <= x) ? V[rr] * rinvs[rr] : 0.0;
    Wd[x + r * 16] = w;
    if (r < x) D[r + x * BLD] = w;                    // strict lower part of the inverse, transposed into the upper triangle
    if (r == x) wdiag[x] = w;
  }
}

// 16x16x16 block product on one wave: acc (+/-)= Aop * Bop with Aop[x][k] = pa[x + k*lda_], Bop[k][y] = pb[y + k*ldb_]
// (both operands are addressed "row index contiguous"), acc lane layout D[x = (lane>>4) + 4r][y = lane&15].
template <int NEG>
__device__ __forceinline__ double4_t block_mma(const double* pa, int lda_, const double* pb, int ldb_, double4_t acc) {
  const int l = threadIdx.x & 63, q = l >> 4, r = l & 15;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[r + (4 * kk + q) * lda_], pb[r + (4 * kk + q) * ldb_], acc, 0, 0, NEG);
  return acc;
}

// one 16x16 block of phase C: b < nchol -> Cholesky trailing block, else inverse block (see the kernel)
__device__ __forceinline__ void potf2_phase_c_block(double* S, const double* Wd, int s, int c0, int m, int b, int lane, int q, int r) {
  const int nchol = m * (m + 1) / 2;
  if (b < nchol) {
    int ii = 0;
    while ((ii + 1) * (ii + 2) / 2 <= b) ++ii;
    const int I = s + 1 + ii, J = s + 1 + (b - ii * (ii + 1) / 2);
    // D'[x][y] = C_IJ[y][x]: lanes run down the rows of C_IJ (contiguous in LDS)
    double4_t acc;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) acc[rr] = S[(16 * I + r) + (16 * J + q + 4 * rr) * BLD];
    acc = block_mma<1>(S + 16 * J + c0 * BLD, BLD, S + 16 * I + c0 * BLD, BLD, acc);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) S[(16 * I + r) + (16 * J + q + 4 * rr) * BLD] = acc[rr];
  } else {
    const int e = b - nchol;
    const int I = s + 1 + e / (s + 1), J = e % (s + 1);
    // Y_IJ[a][b'] -= sum_k L_Is[a][k] * X_sJ[k][b'];  Y_IJ[a][b'] sits transposed at S[(16J + b') + (16I + a) * BLD]
    double4_t acc;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) acc[rr] = S[(16 * J + r) + (16 * I + q + 4 * rr) * BLD];
    if (J < s) acc = block_mma<1>(S + 16 * I + c0 * BLD, BLD, S + 16 * J + c0 * BLD, BLD, acc);
    else {  // X_ss = Wd: Bop[k][b'] = Wd[k][b'] = Wd[k + b' * 16]  -> "row index contiguous" means pb[b' + k * ld] with the transposed view
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(S[(16 * I + (lane & 15)) + (c0 + 4 * kk + (lane >> 4)) * BLD],
                                                   Wd[(4 * kk + (lane >> 4)) + (lane & 15) * 16], acc, 0, 0, 1);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) S[(16 * J + r) + (16 * I + q + 4 * rr) * BLD] = acc[rr];
  }
}

// the body: NW waves (16: the stand-alone kernel; 8: the factor role of panel_fused_kernel), sm = PB_SMEM_DOUBLES doubles
// of LDS.  Which wave computes a 16x16 block has no influence on the block's arithmetic: same bits for any NW.
// ptr (may be null; GPRC_POTF2_TRACE): s_memrealtime stamps of thread 0 -- [0] entry, [1] block loaded, [2] first 16x16 sweep done,
// [3 + 2 s] / [4 + 2 s] after the two barriers of step s, [19] exit (stores issued).  Measurement only.
#define POTF2_STAMP(k) do { if (ptr && threadIdx.x == 0) ptr[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
// preloaded: S already holds the block (gemm_tile_128<.., LDSOUT>), the loads from A are skipped.
template <int NW>
__device__ __attribute__((noinline)) void potf2_blocked_body(double* sm, double* A, int64_t lda, double* winv, int* info, int col0,
                                                             unsigned long long* ptr = nullptr, bool preloaded = false) {
  static_assert(NW >= 8, "phase B needs one wave per task: 7 tasks per step");
  constexpr int TYS = NW / 2;          // column groups of the 128-row load / store loops (NW * 64 threads / 128 rows)
  double* S = sm;                      // PB x BLD
  double* Wd2 = sm + PB * BLD;         // 2 x (16 x 16): the 16x16 inverse of step s lives in buffer s & 1
  double* Wdiag = Wd2 + 512;           // PB
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int q = lane >> 4, r = lane & 15;
  POTF2_STAMP(0);
  if (!preloaded) {
    // all the loads of a thread's 128 / TYS columns are issued before the first LDS store (the loop with one conditional load
    // per iteration took 4 us of a 46-us block); entries above the diagonal are read too -- allocated storage -- and dropped
    const int i = t & 127, ty = t >> 7;
    constexpr int BATCH = (NW == 8) ? 32 : 16;      // loads in flight per thread (the 16-wave kernel is capped at 128 VGPRs)
    for (int k0 = 0; k0 < PB / TYS; k0 += BATCH) {
      double v[BATCH];
#pragma unroll
      for (int k = 0; k < BATCH; ++k) v[k] = A[i + (int64_t)(ty + TYS * (k0 + k)) * lda];
#pragma unroll
      for (int k = 0; k < BATCH; ++k) S[i + (ty + TYS * (k0 + k)) * BLD] = (i >= ty + TYS * (k0 + k)) ? v[k] : 0.0;
    }
  }
  __syncthreads();
  POTF2_STAMP(1);
  if (wave == 0) diag16(S, Wd2, Wdiag, info, col0);  // phase A of step 0
  __syncthreads();
  POTF2_STAMP(2);
  for (int s = 0; s < 8; ++s) {
    const int c0 = 16 * s, m = 7 - s;
    const double* Wd = Wd2 + (s & 1) * 256;
    // ---- phase B: m blocks of the panel below and s blocks of inverse row s -- always 7 tasks, one wave each
    if (wave < m) {
      const int I = s + 1 + wave;
      // D'[x][y] = sum_k Wd[x][k] * A_I[y][k] = X_I[y][x]
      double4_t acc = block_mma<0>(Wd, 16, S + 16 * I + c0 * BLD, BLD, (double4_t){0.0, 0.0, 0.0, 0.0});
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) S[(16 * I + r) + (c0 + q + 4 * rr) * BLD] = acc[rr];
    } else if (wave - m < s) {
      const int J = wave - m;
      // X_sJ[a][b] = sum_k Wd[a][k] * Y_sJ[k][b]; Y_sJ[k][b] sits transposed at S[(16J + b) + (c0 + k) * BLD]
      double4_t acc = block_mma<0>(Wd, 16, S + 16 * J + c0 * BLD, BLD, (double4_t){0.0, 0.0, 0.0, 0.0});
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) S[(16 * J + r) + (c0 + q + 4 * rr) * BLD] = acc[rr];
    }
    __syncthreads();
    POTF2_STAMP(3 + 2 * s);
    // ---- phase C with look-ahead: m(m+1)/2 Cholesky blocks then m*(s+1) inverse blocks.  Wave 0 takes block 0 -- the
    // next diagonal block (s+1, s+1) -- and goes straight on to phase A of step s+1 (the sequential 16-pivot sweep,
    // the longest single piece of the kernel) while the other waves work through the other blocks; nothing they touch
    // overlaps that block, and its 16x16 inverse goes to the other Wd buffer.
    const int total = m * (m + 1) / 2 + m * (s + 1);
    if (wave == 0) {
      if (s < 7) {
        potf2_phase_c_block(S, Wd, s, c0, m, 0, lane, q, r);
        diag16(S + (c0 + 16) + (c0 + 16) * BLD, Wd2 + ((s + 1) & 1) * 256, Wdiag + c0 + 16, info, col0 + c0 + 16);
      }
    } else {
      for (int b = wave; b < total; b += NW - 1) potf2_phase_c_block(S, Wd, s, c0, m, b, lane, q, r);
    }
    // (a preloaded block's stores to memory -- gemm_tile_128<.., LDSOUT> did not wait for them -- are complete in every wave before
    // this barrier, hence before any wave stores the factored block over them below; free after the first step)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    POTF2_STAMP(4 + 2 * s);
  }
  {
    const int i = t & 127, ty = t >> 7;
    const double wd = Wdiag[i];
    constexpr int BATCH = 8;
    for (int k0 = 0; k0 < PB / TYS; k0 += BATCH) {  // LDS reads of a batch first (S[c + i BLD]: a row walk, stride BLD), then its stores back to back
      double lv[BATCH], wv[BATCH];
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int c = ty + TYS * (k0 + k);
        lv[k] = S[i + c * BLD];
        wv[k] = (i > c) ? S[c + i * BLD] : (i == c ? wd : 0.0);
      }
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int c = ty + TYS * (k0 + k);
        if (i >= c) A[i + (int64_t)c * lda] = lv[k];
        winv[i + c * PB] = wv[k];
      }
    }
  }
  POTF2_STAMP(19);
}

__global__ __launch_bounds__(1024) void potf2_inv_blocked_kernel(double* A, int64_t lda, double* winv, int* info, int col0) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  potf2_blocked_body<16>(sm, A, lda, winv, info, col0);
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

// operands of one k-step (4 consecutive k) for this wave's 64x64 sub-tile: 4 A + 4 B doubles per lane
__device__ __forceinline__ void read_ops(const double* Ac, const double* Bc, int kk, double (&a)[4], double (&b)[4]) {
#pragma unroll
  for (int m = 0; m < 4; ++m) a[m] = Ac[kk * 4 * G_LDT + m * 16];
#pragma unroll
  for (int n = 0; n < 4; ++n) b[n] = Bc[kk * 4 * G_LDT + n * 16];
}
// NEG = 1 sets the f64 MFMA's negate-A bit (the BLGP field is the NEG set on f64 MFMA; verified on gfx950 by
// tools/microbench/mfma_neg.hip): acc = acc - op_a * op_b, exactly.
template <int NEG>
__device__ __forceinline__ void mma_step(const double (&a)[4], const double (&b)[4], double4_t (&acc)[4][4]) {
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[n], a[m], acc[m][n], 0, 0, NEG);
}

// LDS-DMA of one k-tile of both strips: wave w moves k-slices w, w+4, w+8, w+12 of A and of B; one
// global_load_lds_dwordx4 per slice = 64 lanes x 16 B = the slice's 128 rows, landing contiguously at a
// wave-uniform LDS row (the 128-byte row pad survives because a row is exactly one instruction).
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void dma_ktile(const double* Ag, int64_t lda, const double* Bg, int64_t ldb, double* Asb, double* Bsb) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    __builtin_amdgcn_global_load_lds((gptr_t)(Ag + (int64_t)(4 * i) * lda), (lptr_t)(Asb + 4 * i * G_LDT), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(Bg + (int64_t)(4 * i) * ldb), (lptr_t)(Bsb + 4 * i * G_LDT), 16, 0, 0);
  }
}

// Where k-tile kt of an operand strip lives (this lane's 16 bytes of k-slice `wave`).  Plain: a column-major strip with
// one leading dimension.  SEG: the strip is rows [brow, brow + 128) of the PACKED factor across its first K columns
// (B = packed base, ldb = n_pad): every NB columns it moves to the next panel, with that panel's own leading dimension.
template <bool SEG>
__device__ __forceinline__ const double* strip_ktile(const double* B, int64_t ldb, int64_t brow, int kt, int lane, int wave, int64_t& ld) {
  if constexpr (SEG) {
    const int pp = kt / (NB / 16);                    // panel that holds k-tile kt
    ld = panel_ld(ldb, pp);
    return B + panel_offset(ldb, pp) + (brow - (int64_t)pp * NB) + (int64_t)((kt % (NB / 16)) * 16 + wave) * ld + 2 * lane;
  } else {
    ld = ldb;
    return B + 2 * lane + (int64_t)(kt * 16 + wave) * ldb;
  }
}

// The same place WITHOUT the lane's part: the wave-uniform address of k-slice `wave` of k-tile kt (the lane adds 16 bytes x lane as a
// 32-bit VGPR offset of the LDS-DMA instruction, whose base is then an SGPR pair: no VALU address arithmetic in the loop).
template <bool SEG>
__device__ __forceinline__ const char* strip_ktile_s(const double* B, int64_t ldb, int64_t brow, int kt, int wave, int64_t& ld) {
  if constexpr (SEG) {
    const int pp = kt / (NB / 16);
    ld = panel_ld(ldb, pp);
    return reinterpret_cast<const char*>(B + panel_offset(ldb, pp) + (brow - (int64_t)pp * NB) + (int64_t)((kt % (NB / 16)) * 16 + wave) * ld);
  } else {
    ld = ldb;
    return reinterpret_cast<const char*>(B + (int64_t)(kt * 16 + wave) * ldb);
  }
}

// SSQ (the predict's panel solve only): besides storing the tile, leave in ssq[0..127] the sum of squares of each of the
// tile's 128 rows over its 128 columns -- these columns of v^T are final after this tile, so colSums(v * v)
// (R/GPRclass.R:164) is assembled from these per-block partials and the pass that re-read the whole solved chunk is gone.
// Fixed order: a lane's 16 columns (n, r ascending), the four lanes of a row (xor 16, xor 32), the two column waves.
// LDSOUT: the finished tile ALSO goes to lds_out as the 128 x 128 LDS image potf2_blocked_body works on (leading dimension 144,
// zero above the diagonal) -- the factor role hands the updated diagonal block to the factorisation without the round trip
// through memory (one more workgroup barrier than without: every wave must be past its last operand read, the image overlaps
// the staging buffers).
// ILV: the main loop with every non-MFMA instruction in the shadow of an MFMA (below) -- the throughput kernels; false keeps the
// block-structured loop for the roles of the fused panel / service kernels, which inline this function several times and spill
// with the larger body (their tiles are short -- K = 128..384 -- and paced by flags, not by the loop).
// CORE (interleaved loops only): 2 = the loop without VALU instructions (the plain throughput kernels), 1 = the first interleaved loop
// (kept for the tiles that run beside the factor service -- sweep kernel, trailing_service_kernel: measured, see the loops' comments).
// WT: the tile is stored WRITE-THROUGH (sc1: global_store ... sc1, the agent-scope relaxed atomic store), leaving no dirty line in the XCD's L2.
template <bool SET, bool SEG = false, bool SEGA = false, bool SSQ = false, bool LDSOUT = false, bool ILV = true, int CORE = 2, bool WT = false>
__device__ __forceinline__ void gemm_tile_128(double* C, int64_t ldc, const double* A, int64_t lda, const double* B,
                                              int64_t ldb, int K, double* smem, int64_t brow = 0, int64_t arow = 0, int kt0 = 0,
                                              double* ssq = nullptr, int tid = -1, double* lds_out = nullptr) {
  const int t = tid < 0 ? (int)threadIdx.x : tid, lane = t & 63;   // tid: a 256-thread team inside a larger workgroup
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int fk = lane >> 4, fr = lane & 15;
  double* As = smem;
  double* Bs = smem + 2 * G_BUF;

  // C -= A*B^T: the accumulators START as the C tile (its loads fly with the first DMA) and every MFMA
  // subtracts, so the epilogue is stores only.  SET: accumulators start at zero, plain products.
  constexpr int NEG = SET ? 0 : 1;
  double* Cw = C + (wr * 64 + fr) + (int64_t)(wc * 64 + fk) * ldc;
  double4_t acc[4][4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m][n][r] = SET ? 0.0 : Cw[m * 16 + (int64_t)(n * 16 + 4 * r) * ldc];

  int64_t ldak, ldbk;
  // kt0 (first k-tile of the pass) addresses the PACKED operands only: a plain strip is handed over already pointing at
  // its first k-tile (the running pointer below restarts from it)
  const double* Ag = strip_ktile<SEGA>(A, lda, arow, SEGA ? kt0 : 0, lane, wave, ldak);  // this lane's 16 bytes of k-slice `wave`
  const double* Bg = strip_ktile<SEG>(B, ldb, brow, SEG ? kt0 : 0, lane, wave, ldbk);
  const int srow = wave * G_LDT;                          // LDS row of that slice (wave-uniform)
  const int foff = fr + fk * G_LDT;                       // this lane's MFMA operand element

  // Software pipeline, operand reads TWO k-steps ahead.  Four named operand sets (one per k-step of a tile).
  // Every k-step is:  s_waitcnt lgkmcnt(0)  ->  issue the reads of step +2  ->  16 MFMAs of this step.
  // The wait therefore only ever covers reads issued one whole MFMA block (>= 1000 cycles) earlier; with the
  // reads issued right in front of the compiler's own lgkmcnt(0) they were waited for on the spot.
  //   step (t,0): reads (t,2)          step (t,1): reads (t,3)   <- last LDS reads of tile t's buffer
  //   step (t,2): vmcnt(0) + barrier [tile t+1 landed, tile t's buffer drained]; DMA tile t+2; reads (t+1,0)
  //   step (t,3): reads (t+1,1)
  // so a DMA has 64 MFMAs (4096 cycles) to land, as before.
  constexpr int LGKM0 = 0xC07F;  // s_waitcnt lgkmcnt(0), vmcnt/expcnt untouched
  const int KT = K / G_KB;
  dma_ktile(Ag, ldak, Bg, ldbk, As + srow, Bs + srow);
  // vmcnt(0) through the BUILTIN, not inline asm, so that the compiler's waitcnt pass knows the C-tile loads
  // above have completed and does not re-wait vmcnt(0) (draining fresh DMAs) inside the loop.  0x0F70 = vmcnt(0).
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  if (KT > 1) {
    Ag = strip_ktile<SEGA>(A, lda, arow, (SEGA ? kt0 : 0) + 1, lane, wave, ldak);
    Bg = strip_ktile<SEG>(B, ldb, brow, (SEG ? kt0 : 0) + 1, lane, wave, ldbk);
    dma_ktile(Ag, ldak, Bg, ldbk, As + G_BUF + srow, Bs + G_BUF + srow);
  }
  if constexpr (!SEGA) Ag = A + 2 * lane + (int64_t)(2 * G_KB + wave) * lda;  // next tile to request: kt + 2
  if constexpr (!SEG) Bg = B + 2 * lane + (int64_t)(2 * G_KB + wave) * ldb;
  double a0[4], b0[4], a1[4], b1[4], a2[4], b2[4], a3[4], b3[4];
  read_ops(As + wr * 64 + foff, Bs + wc * 64 + foff, 0, a0, b0);
  read_ops(As + wr * 64 + foff, Bs + wc * 64 + foff, 1, a1, b1);
#ifndef GPRC_CORE_CLUMPED
  constexpr bool interleaved = ILV;
#else
  constexpr bool interleaved = false;
#endif
  if constexpr (interleaved && CORE == 2) {
  // ---- Main loop (round 3, second form): no VALU instruction but the MFMAs ------------------------------------------------------
  // An f64 MFMA executes on the SIMD's double-precision lanes, and a VALU instruction issued behind it -- a 32-bit address add as
  // much as an FMA -- takes the pipe away from the next MFMA: tools/microbench/mfma_valu_mix.hip measures 64 cycles per MFMA for a
  // pure stream, +14 with one v_add_u32 behind each MFMA, +18 with two; SALU and LDS instructions cost nothing.  The first
  // interleaved loop still carried ~20 VALU instructions per k-tile and wave (ISA: 8 v_lshl_add_u64 for the LDS-DMA addresses, 12
  // v_add_u32 / v_subrev_u32 for the ds_read2 bases) -- ~5 % of the pipe.  Here the loop has none:
  //   * LDS-DMA with an SGPR base: global_load_lds_dwordx4 v_off, s[base:base+1] -- the k-slice's address is wave-uniform, the lane
  //     contributes a constant 32-bit offset (16 B x lane); the compiler has no such selection for the builtin, hence inline assembly
  //     (m0 = the slice's LDS row, set by s_mov in the same statement);
  //   * operand reads as ds_read_b64 with 16-bit immediate offsets from TWO loop-invariant base registers (one per operand): every
  //     buffer / k-step / block offset is a constant of the instruction once the loop is unrolled by two k-tiles (ds_read2_b64's 8-bit
  //     offsets reach 2 KB only, and the compiler paid a v_add_u32 per pair for them);
  //   * the waits for those reads counted by hand (the compiler does not see assembly loads): the reads of a k-step are issued two
  //     blocks ahead, eight per block, in order: s_waitcnt lgkmcnt(8) in front of a block leaves exactly the next block's in flight.
  // Same products, same k order, same accumulators as before: identical bits.
  const unsigned lane_off = 16u * (unsigned)lane;
  const unsigned ldsA = (unsigned)(uintptr_t)(lptr_t)(As + srow), ldsB = (unsigned)(uintptr_t)(lptr_t)(Bs + srow);                    // wave-uniform
  const unsigned aBase = (unsigned)(uintptr_t)(lptr_t)(As + wr * 64 + foff), bBase = (unsigned)(uintptr_t)(lptr_t)(Bs + wc * 64 + foff);  // per lane
  // next tile to request (k-tile 2): wave-uniform addresses
  int64_t ldas, ldbs;
  const char* Asg = strip_ktile_s<SEGA>(A, lda, arow, (SEGA ? kt0 : 0) + 2, wave, ldas);
  const char* Bsg = strip_ktile_s<SEG>(B, ldb, brow, (SEG ? kt0 : 0) + 2, wave, ldbs);
#define GPRC_SB __builtin_amdgcn_sched_barrier(0);
#define GPRC_M(A_, B_, i) acc[(i) >> 2][(i) & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(B_[(i) & 3], A_[(i) >> 2], acc[(i) >> 2][(i) & 3], 0, 0, NEG); GPRC_SB
#define GPRC_RD(dst, base, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off));
  // two reads (blocks 2r, 2r+1) of k-step kk in buffer buf
#define GPRC_RA(buf, kk, r, A_) GPRC_RD(A_[2 * (r)], aBase, ((buf) * G_BUF + (kk) * 4 * G_LDT + (2 * (r)) * 16) * 8) GPRC_RD(A_[2 * (r) + 1], aBase, ((buf) * G_BUF + (kk) * 4 * G_LDT + (2 * (r) + 1) * 16) * 8) GPRC_SB
#define GPRC_RB(buf, kk, r, B_) GPRC_RD(B_[2 * (r)], bBase, ((buf) * G_BUF + (kk) * 4 * G_LDT + (2 * (r)) * 16) * 8) GPRC_RD(B_[2 * (r) + 1], bBase, ((buf) * G_BUF + (kk) * 4 * G_LDT + (2 * (r) + 1) * 16) * 8) GPRC_SB
#define GPRC_READY(cnt, A_, B_) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(A_[0]), "+v"(A_[1]), "+v"(A_[2]), "+v"(A_[3]), "+v"(B_[0]), "+v"(B_[1]), "+v"(B_[2]), "+v"(B_[3])); GPRC_SB
#define GPRC_BLOCK_R(A_, B_, buf, kk, RA_, RB_, cond)                                                               \
  GPRC_M(A_, B_, 0) GPRC_M(A_, B_, 1) if (cond) { GPRC_RA(buf, kk, 0, RA_) }                                        \
  GPRC_M(A_, B_, 2) GPRC_M(A_, B_, 3) if (cond) { GPRC_RA(buf, kk, 1, RA_) }                                        \
  GPRC_M(A_, B_, 4) GPRC_M(A_, B_, 5) if (cond) { GPRC_RB(buf, kk, 0, RB_) }                                        \
  GPRC_M(A_, B_, 6) GPRC_M(A_, B_, 7) if (cond) { GPRC_RB(buf, kk, 1, RB_) }                                        \
  GPRC_M(A_, B_, 8) GPRC_M(A_, B_, 9) GPRC_M(A_, B_, 10) GPRC_M(A_, B_, 11) GPRC_M(A_, B_, 12) GPRC_M(A_, B_, 13) GPRC_M(A_, B_, 14) GPRC_M(A_, B_, 15)
  // one LDS-DMA: k-slice wave + 4 i of the tile at Asg / Bsg into buffer buf (m0 <- the slice's LDS row; one wait state before its use)
#define GPRC_DMA(i, buf, more2)                                                                                      \
  if (more2) {                                                                                                      \
    if ((i) < 4) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(Asg + (int64_t)(4 * (i)) * ldas * 8), "s"(ldsA + (unsigned)(((buf) * G_BUF + 4 * (i) * G_LDT) * 8)) : "memory"); \
    else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(Bsg + (int64_t)(4 * ((i) - 4)) * ldbs * 8), "s"(ldsB + (unsigned)(((buf) * G_BUF + 4 * ((i) - 4) * G_LDT) * 8)) : "memory"); \
    GPRC_SB                                                                                                         \
  }
  // one k-tile in buffer buf (the next one in 1 - buf); more1 / more2: a tile kt+1 / kt+2 exists; kt: this tile's index
#define GPRC_KTILE(buf, more1, more2)                                                                                \
  {                                                                                                                 \
    GPRC_SB                                                                                                         \
    GPRC_READY(8, a0, b0)                                                                                           \
    GPRC_BLOCK_R(a0, b0, buf, 2, a2, b2, true)                                                                      \
    GPRC_READY(8, a1, b1)                                                                                           \
    GPRC_BLOCK_R(a1, b1, buf, 3, a3, b3, true)                                                                      \
    GPRC_READY(8, a2, b2)                                                                                           \
    GPRC_M(a2, b2, 0) GPRC_M(a2, b2, 1) GPRC_M(a2, b2, 2) GPRC_M(a2, b2, 3)                                         \
    if (more1) {                                                                                                    \
      /* this wave's reads of buffer buf have returned (lgkmcnt) and its share of tile kt+1 has landed (vmcnt); after the */ \
      /* barrier that holds for every wave: tile kt+1 may be read and buffer buf overwritten */                     \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                   \
      __builtin_amdgcn_s_barrier();                                                                                 \
      GPRC_SB                                                                                                       \
    }                                                                                                               \
    GPRC_M(a2, b2, 4) if (more1) { GPRC_RA(1 - (buf), 0, 0, a0) }                                                   \
    GPRC_M(a2, b2, 5) if (more1) { GPRC_RA(1 - (buf), 0, 1, a0) }                                                   \
    GPRC_M(a2, b2, 6) if (more1) { GPRC_RB(1 - (buf), 0, 0, b0) }                                                   \
    GPRC_M(a2, b2, 7) if (more1) { GPRC_RB(1 - (buf), 0, 1, b0) }                                                   \
    GPRC_M(a2, b2, 8) GPRC_DMA(0, buf, more2) GPRC_M(a2, b2, 9) GPRC_DMA(4, buf, more2) GPRC_M(a2, b2, 10) GPRC_DMA(1, buf, more2) GPRC_M(a2, b2, 11) GPRC_DMA(5, buf, more2) \
    GPRC_M(a2, b2, 12) GPRC_DMA(2, buf, more2) GPRC_M(a2, b2, 13) GPRC_DMA(6, buf, more2) GPRC_M(a2, b2, 14) GPRC_DMA(3, buf, more2) GPRC_M(a2, b2, 15) GPRC_DMA(7, buf, more2) \
    if (more2) {   /* the tile after the one just requested: 16 columns on, or the next panel of a packed strip (scalar arithmetic) */ \
      if constexpr (SEGA) { if (((kt0 + kt + 3) % (NB / 16)) == 0) Asg = strip_ktile_s<true>(A, lda, arow, kt0 + kt + 3, wave, ldas); else Asg += (int64_t)G_KB * ldas * 8; } \
      else Asg += (int64_t)G_KB * ldas * 8;                                                                         \
      if constexpr (SEG) { if (((kt0 + kt + 3) % (NB / 16)) == 0) Bsg = strip_ktile_s<true>(B, ldb, brow, kt0 + kt + 3, wave, ldbs); else Bsg += (int64_t)G_KB * ldbs * 8; } \
      else Bsg += (int64_t)G_KB * ldbs * 8;                                                                         \
      GPRC_SB                                                                                                       \
    }                                                                                                               \
    if (more1) { GPRC_READY(8, a3, b3) } else { GPRC_READY(0, a3, b3) }                                             \
    GPRC_BLOCK_R(a3, b3, 1 - (buf), 1, a1, b1, more1)                                                               \
  }
  {   // KT is even and >= 4: every caller's K is a multiple of 128 (eight k-tiles); launch_gemm_nt, the one launcher with a free K, checks it
    int kt = 0;
    for (; kt + 2 < KT; kt += 2) {
      GPRC_KTILE(0, true, true)
      ++kt;
      GPRC_KTILE(1, true, true)
      --kt;
    }
    GPRC_KTILE(0, true, false)
    ++kt;
    GPRC_KTILE(1, false, false)
  }
#undef GPRC_KTILE
#undef GPRC_DMA
#undef GPRC_BLOCK_R
#undef GPRC_READY
#undef GPRC_RA
#undef GPRC_RB
#undef GPRC_RD
#undef GPRC_M
#undef GPRC_SB
  } else if constexpr (interleaved) {
  // ---- Main loop, every non-MFMA instruction in the shadow of an MFMA (round 3, first form) -----------------------------------
  // (Still ~20 VALU instructions per k-tile and wave.  It stays for the K = 512 tiles of the kernels that run BESIDE the factor
  //  service: with the VALU-free loop their tiles are a third faster, the memory system correspondingly busier, and the service's
  //  latency-bound roles -- and with them the whole mid-size factorisation -- slower: n = 12288 13.55 -> 14.66 ms, 16384 27.44 -> 28.27,
  //  same box, profiles/r03_factor_schedules.txt.)
  // Counters on the shipped loop (profiles/r03_c4_core_counters.txt): the MFMA pipes were busy 91.7 % of the kernel's cycles at
  // 2.37 GHz, and per 64 MFMAs a wave issues 73 other instructions -- 16 ds_read2, 8 LDS-DMA with their m0 / address set-up, the
  // s_waitcnt / s_barrier, loop arithmetic.  In the block-structured loop they sat in CLUMPS between the 16-MFMA blocks: four
  // operand-read groups and, once per k-tile, vmcnt(0) + barrier + 8 DMA issues + 4 reads with nothing but the block's last MFMA
  // in flight.  A wave is in order: while it works through a clump it issues no MFMA, and its SIMD's pipe runs dry unless the
  // partner wave happens to be inside a block (a wave alone on its SIMD reached 75 %).  Here every MFMA is followed by at most
  // one other operation (a ds_read2, or one DMA with its set-up), the barrier sits BETWEEN two MFMAs of block 2 with four MFMAs
  // of the same wave still queued on the pipe, and the order is pinned by a sched_barrier after every statement.  The products,
  // their k order and the accumulator each one lands in are unchanged: identical bits.
  //   block 0 (a0,b0): reads (t,2) after MFMAs 1,3,5,7          block 1 (a1,b1): reads (t,3) after 1,3,5,7
  //   block 2 (a2,b2): MFMAs 0-3 | lgkmcnt(0) vmcnt(0) s_barrier | reads (t+1,0) after 4,5,6,7 | DMA slice i of tile t+2 after 8+i
  //   block 3 (a3,b3): reads (t+1,1) after 1,3,5,7
#define GPRC_SB __builtin_amdgcn_sched_barrier(0);
  // (timing experiments only -- wrong results: GPRC_EXP_NOVM drops the wait for the landed DMA, GPRC_EXP_NOBAR the workgroup barrier)
#ifdef GPRC_EXP_NOVM
#define GPRC_TILE_WAIT asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
#define GPRC_TILE_WAIT asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
#ifdef GPRC_EXP_NOBAR
#define GPRC_TILE_BARRIER
#else
#define GPRC_TILE_BARRIER __builtin_amdgcn_s_barrier();
#endif
#define GPRC_M(A_, B_, i) acc[(i) >> 2][(i) & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(B_[(i) & 3], A_[(i) >> 2], acc[(i) >> 2][(i) & 3], 0, 0, NEG); GPRC_SB
  // one ds_read2_b64 each: RA r = 0, 1 -> a[2r], a[2r+1]; RB r = 0, 1 -> b[2r], b[2r+1]
#define GPRC_RA(Ap_, kk, r, A_) A_[2 * (r)] = (Ap_)[(kk) * 4 * G_LDT + (2 * (r)) * 16]; A_[2 * (r) + 1] = (Ap_)[(kk) * 4 * G_LDT + (2 * (r) + 1) * 16]; GPRC_SB
#define GPRC_RB(Bp_, kk, r, B_) B_[2 * (r)] = (Bp_)[(kk) * 4 * G_LDT + (2 * (r)) * 16]; B_[2 * (r) + 1] = (Bp_)[(kk) * 4 * G_LDT + (2 * (r) + 1) * 16]; GPRC_SB
#define GPRC_BLOCK_R(A_, B_, Ap_, Bp_, kk, RA_, RB_, cond)                                                          \
  GPRC_M(A_, B_, 0) GPRC_M(A_, B_, 1) if (cond) { GPRC_RA(Ap_, kk, 0, RA_) }                                        \
  GPRC_M(A_, B_, 2) GPRC_M(A_, B_, 3) if (cond) { GPRC_RA(Ap_, kk, 1, RA_) }                                        \
  GPRC_M(A_, B_, 4) GPRC_M(A_, B_, 5) if (cond) { GPRC_RB(Bp_, kk, 0, RB_) }                                        \
  GPRC_M(A_, B_, 6) GPRC_M(A_, B_, 7) if (cond) { GPRC_RB(Bp_, kk, 1, RB_) }                                        \
  GPRC_M(A_, B_, 8) GPRC_M(A_, B_, 9) GPRC_M(A_, B_, 10) GPRC_M(A_, B_, 11) GPRC_M(A_, B_, 12) GPRC_M(A_, B_, 13) GPRC_M(A_, B_, 14) GPRC_M(A_, B_, 15)
#define GPRC_DMA(i, more2)                                                                                           \
  if (more2) {                                                                                                      \
    if ((i) < 4) __builtin_amdgcn_global_load_lds((gptr_t)(Ag + (int64_t)(4 * (i)) * ldak), (lptr_t)(As + cur + srow + 4 * (i) * G_LDT), 16, 0, 0); \
    else __builtin_amdgcn_global_load_lds((gptr_t)(Bg + (int64_t)(4 * ((i) - 4)) * ldbk), (lptr_t)(Bs + cur + srow + 4 * ((i) - 4) * G_LDT), 16, 0, 0); \
    GPRC_SB                                                                                                         \
  }
  // one k-tile; more1 / more2: a tile kt+1 / kt+2 exists (compile-time true in the steady-state loop, so that it has no branches)
#define GPRC_KTILE(more1, more2)                                                                                     \
  {                                                                                                                 \
    const int cur = (kt & 1) * G_BUF, nxt = G_BUF - cur;                                                            \
    const double* Ac = As + cur + wr * 64 + foff;                                                                   \
    const double* Bc = Bs + cur + wc * 64 + foff;                                                                   \
    const double* An = As + nxt + wr * 64 + foff;                                                                   \
    const double* Bn = Bs + nxt + wc * 64 + foff;                                                                   \
    GPRC_SB                                                                                                         \
    GPRC_BLOCK_R(a0, b0, Ac, Bc, 2, a2, b2, true)                                                                   \
    GPRC_BLOCK_R(a1, b1, Ac, Bc, 3, a3, b3, true)                                                                   \
    GPRC_M(a2, b2, 0) GPRC_M(a2, b2, 1) GPRC_M(a2, b2, 2) GPRC_M(a2, b2, 3)                                         \
    if (more1) {                                                                                                    \
      /* this wave's reads of buffer `cur` have returned (lgkmcnt) and its share of tile kt+1 has landed (vmcnt); after the */ \
      /* barrier that holds for every wave: tile kt+1 may be read and buffer `cur` overwritten */                   \
      GPRC_TILE_WAIT                                                                                                \
      GPRC_TILE_BARRIER                                                                                             \
      GPRC_SB                                                                                                       \
      if (more2) {                                                                                                  \
        if constexpr (SEGA || SEG) {                                                                                \
          const bool boundary = ((kt0 + kt + 2) % (NB / 16)) == 0;                                                  \
          if constexpr (SEGA) { if (boundary) Ag = strip_ktile<true>(A, lda, arow, kt0 + kt + 2, lane, wave, ldak); else Ag += (int64_t)G_KB * ldak; } \
          if constexpr (SEG) { if (boundary) Bg = strip_ktile<true>(B, ldb, brow, kt0 + kt + 2, lane, wave, ldbk); else Bg += (int64_t)G_KB * ldbk; } \
        }                                                                                                           \
        GPRC_SB                                                                                                     \
      }                                                                                                             \
    }                                                                                                               \
    GPRC_M(a2, b2, 4) if (more1) { GPRC_RA(An, 0, 0, a0) }                                                          \
    GPRC_M(a2, b2, 5) if (more1) { GPRC_RA(An, 0, 1, a0) }                                                          \
    GPRC_M(a2, b2, 6) if (more1) { GPRC_RB(Bn, 0, 0, b0) }                                                          \
    GPRC_M(a2, b2, 7) if (more1) { GPRC_RB(Bn, 0, 1, b0) }                                                          \
    GPRC_M(a2, b2, 8) GPRC_DMA(0, more2) GPRC_M(a2, b2, 9) GPRC_DMA(4, more2) GPRC_M(a2, b2, 10) GPRC_DMA(1, more2) GPRC_M(a2, b2, 11) GPRC_DMA(5, more2) \
    GPRC_M(a2, b2, 12) GPRC_DMA(2, more2) GPRC_M(a2, b2, 13) GPRC_DMA(6, more2) GPRC_M(a2, b2, 14) GPRC_DMA(3, more2) GPRC_M(a2, b2, 15) GPRC_DMA(7, more2) \
    if (more2) {                                                                                                    \
      if constexpr (!SEGA) Ag += (int64_t)G_KB * lda;                                                               \
      if constexpr (!SEG) Bg += (int64_t)G_KB * ldb;                                                                \
    }                                                                                                               \
    GPRC_BLOCK_R(a3, b3, An, Bn, 1, a1, b1, more1)                                                                  \
  }
  {
    int kt = 0;
    for (; kt + 2 < KT; ++kt) GPRC_KTILE(true, true)
    for (; kt < KT; ++kt) {
      const bool more1 = kt + 1 < KT;
      GPRC_KTILE(more1, false)
    }
  }
#undef GPRC_KTILE
#undef GPRC_TILE_WAIT
#undef GPRC_TILE_BARRIER
#undef GPRC_DMA
#undef GPRC_BLOCK_R
#undef GPRC_RA
#undef GPRC_RB
#undef GPRC_M
#undef GPRC_SB
  } else {
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = (kt & 1) * G_BUF, nxt = G_BUF - cur;
    const double* Ac = As + cur + wr * 64 + foff;
    const double* Bc = Bs + cur + wc * 64 + foff;
    const double* An = As + nxt + wr * 64 + foff;
    const double* Bn = Bs + nxt + wc * 64 + foff;
    // sched_barrier(0) after every MFMA block: register-only MFMAs otherwise drift across the explicit waits and
    // the compiler merges blocks, putting the reads back in front of a wait
    __builtin_amdgcn_s_waitcnt(LGKM0);
    read_ops(Ac, Bc, 2, a2, b2);
    mma_step<NEG>(a0, b0, acc);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(LGKM0);
    read_ops(Ac, Bc, 3, a3, b3);
    mma_step<NEG>(a1, b1, acc);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(LGKM0);
    if (kt + 1 < KT) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile kt+1 has landed (this wave's share)
      __syncthreads();                                   // ... everyone's share; and buffer `cur` is drained
      if (kt + 2 < KT) {
        // packed operands: inside a panel the next k-tile is 16 columns further on (same leading dimension); only at a
        // panel boundary (every NB / 16 k-tiles) is the address rebuilt from the panel geometry -- the full index
        // arithmetic (64-bit multiplies) on every k-tile cost ~2 % of the long-K passes
        if constexpr (SEGA || SEG) {
          const bool boundary = ((kt0 + kt + 2) % (NB / 16)) == 0;
          if constexpr (SEGA) { if (boundary) Ag = strip_ktile<true>(A, lda, arow, kt0 + kt + 2, lane, wave, ldak); else Ag += (int64_t)G_KB * ldak; }
          if constexpr (SEG) { if (boundary) Bg = strip_ktile<true>(B, ldb, brow, kt0 + kt + 2, lane, wave, ldbk); else Bg += (int64_t)G_KB * ldbk; }
        }
        dma_ktile(Ag, ldak, Bg, ldbk, As + cur + srow, Bs + cur + srow);
        if constexpr (!SEGA) Ag += (int64_t)G_KB * lda;
        if constexpr (!SEG) Bg += (int64_t)G_KB * ldb;
      }
      read_ops(An, Bn, 0, a0, b0);
    }
    mma_step<NEG>(a2, b2, acc);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(LGKM0);
    if (kt + 1 < KT) read_ops(An, Bn, 1, a1, b1);
    mma_step<NEG>(a3, b3, acc);
    __builtin_amdgcn_sched_barrier(0);
  }

  }

#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if constexpr (WT) __hip_atomic_store(&Cw[m * 16 + (int64_t)(n * 16 + 4 * r) * ldc], acc[m][n][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else Cw[m * 16 + (int64_t)(n * 16 + 4 * r) * ldc] = acc[m][n][r];
      }

  if constexpr (LDSOUT) {
    __syncthreads();  // every wave is past its last operand read: the staging buffers are free
    constexpr int LDO = 144;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int row = wr * 64 + fr + m * 16, col = wc * 64 + fk + n * 16 + 4 * r;
          lds_out[row + col * LDO] = (row >= col) ? acc[m][n][r] : 0.0;
        }
  }

  if constexpr (SSQ) {
    double rs[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      double q = 0.0;
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) q = fma(acc[m][n][r], acc[m][n][r], q);
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      rs[m] = q;
    }
    __syncthreads();  // every wave is past its last operand read: the staging buffers are free
    if (fk == 0) {
#pragma unroll
      for (int m = 0; m < 4; ++m) smem[wc * 128 + wr * 64 + m * 16 + fr] = rs[m];
    }
    __syncthreads();
    if (t < 128) ssq[t] = smem[t] + smem[128 + t];
  }
}

// ------------------------------------------------------------------------------------------------
// 256 x 128 macro-tile on 8 waves (512 threads, one workgroup per CU): the same wave-level 64 x 64 MFMA core as
// gemm_tile_128, but the four row-waves share ONE B strip and the A strip is twice as tall -- 384 strip rows staged per k-tile
// for 256 x 128 outputs instead of 2 x 256 for two independent 128 x 128 workgroups: -25 % LDS-DMA traffic per flop.  Every
// output element still accumulates k ascending in the same v_mfma_f64_16x16x4 steps from the loaded C value: bit-identical.
// A: plain column-major strip (256 rows), B: rows [brow, brow + 128) of the PACKED factor starting at k-tile kt0 (the predict's
// left-looking pass).  LDS: A [2][16][272] + B [2][16][144] doubles = 106,496 B.
// ------------------------------------------------------------------------------------------------
constexpr int T2_LDA = 272, T2_LDB = 144;
constexpr int T2_BUFA = G_KB * T2_LDA, T2_BUFB = G_KB * T2_LDB;
constexpr int T2_SMEM_DOUBLES = 2 * T2_BUFA + 2 * T2_BUFB;

__device__ __forceinline__ void read_ops2(const double* Ac, const double* Bc, int kk, double (&a)[4], double (&b)[4]) {
#pragma unroll
  for (int m = 0; m < 4; ++m) a[m] = Ac[kk * 4 * T2_LDA + m * 16];
#pragma unroll
  for (int n = 0; n < 4; ++n) b[n] = Bc[kk * 4 * T2_LDB + n * 16];
}

__device__ __forceinline__ void gemm_tile_256x128_seg(double* C, int64_t ldc, const double* A, int64_t lda, const double* Bpk, int64_t n_pad,
                                                      int K, double* smem, int64_t brow, int kt0) {
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // 0..7
  const int wr = wave >> 1, wc = wave & 1;
  const int fk = lane >> 4, fr = lane & 15;
  double* As = smem;
  double* Bs = smem + 2 * T2_BUFA;
  double* Cw = C + (wr * 64 + fr) + (int64_t)(wc * 64 + fk) * ldc;
  double4_t acc[4][4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m][n][r] = Cw[m * 16 + (int64_t)(n * 16 + 4 * r) * ldc];

  // wave w stages k-slices w and w + 8 of a k-tile: A rows 0..127, A rows 128..255, B rows 0..127 -- three 1-KiB wave-instructions each
  int64_t ldbk;
  const double* Ag = A + 2 * lane + (int64_t)wave * lda;                                   // k-slice `wave` of k-tile 0
  const double* Bg = strip_ktile<true>(Bpk, n_pad, brow, kt0, lane, wave, ldbk);
  auto dma2 = [&](const double* ag, const double* bg, int64_t ldb_, double* Asb, double* Bsb) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)(ag + (int64_t)(8 * i) * lda), (lptr_t)(Asb + (wave + 8 * i) * T2_LDA), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(ag + 128 + (int64_t)(8 * i) * lda), (lptr_t)(Asb + (wave + 8 * i) * T2_LDA + 128), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(bg + (int64_t)(8 * i) * ldb_), (lptr_t)(Bsb + (wave + 8 * i) * T2_LDB), 16, 0, 0);
    }
  };
  constexpr int LGKM0 = 0xC07F;
  const int KT = K / G_KB;
  dma2(Ag, Bg, ldbk, As, Bs);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  if (KT > 1) {
    Ag += (int64_t)G_KB * lda;
    Bg = strip_ktile<true>(Bpk, n_pad, brow, kt0 + 1, lane, wave, ldbk);
    dma2(Ag, Bg, ldbk, As + T2_BUFA, Bs + T2_BUFB);
  }
  const int foffA = fr + fk * T2_LDA, foffB = fr + fk * T2_LDB;
  double a0[4], b0[4], a1[4], b1[4], a2[4], b2[4], a3[4], b3[4];
  read_ops2(As + wr * 64 + foffA, Bs + wc * 64 + foffB, 0, a0, b0);
  read_ops2(As + wr * 64 + foffA, Bs + wc * 64 + foffB, 1, a1, b1);
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    const double* Ac = As + cur * T2_BUFA + wr * 64 + foffA;
    const double* Bc = Bs + cur * T2_BUFB + wc * 64 + foffB;
    const double* An = As + (1 - cur) * T2_BUFA + wr * 64 + foffA;
    const double* Bn = Bs + (1 - cur) * T2_BUFB + wc * 64 + foffB;
    __builtin_amdgcn_s_waitcnt(LGKM0);
    read_ops2(Ac, Bc, 2, a2, b2);
    mma_step<1>(a0, b0, acc);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(LGKM0);
    read_ops2(Ac, Bc, 3, a3, b3);
    mma_step<1>(a1, b1, acc);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(LGKM0);
    if (kt + 1 < KT) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 2 < KT) {
        Ag += (int64_t)G_KB * lda;
        if (((kt0 + kt + 2) % (NB / 16)) == 0) Bg = strip_ktile<true>(Bpk, n_pad, brow, kt0 + kt + 2, lane, wave, ldbk);
        else Bg += (int64_t)G_KB * ldbk;
        dma2(Ag, Bg, ldbk, As + cur * T2_BUFA, Bs + cur * T2_BUFB);
      }
      read_ops2(An, Bn, 0, a0, b0);
    }
    mma_step<1>(a2, b2, acc);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(LGKM0);
    if (kt + 1 < KT) read_ops2(An, Bn, 1, a1, b1);
    mma_step<1>(a3, b3, acc);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int m = 0; m < 4; ++m) Cw[m * 16 + (int64_t)(n * 16 + 4 * r) * ldc] = acc[m][n][r];
}

// blockIdx -> logical id so that each XCD (blocks b, b+8, ... share one) owns a contiguous id range
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// C[M x N] -= A * B^T, 1-D grid of (M/128)*(N/128) tiles visited in 8-row groups.  ROLE only gives
// each use its own symbol (rocprof / event profiler tell them apart): in-panel update (K = 128),
// predict-side right update (K = 512), posterior-covariance SYRK (K = n).
template <int ROLE>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(double* C, int64_t ldc, const double* A, int64_t lda,
                                                         const double* B, int64_t ldb, int tiles_m, int tiles_n, int K,
                                                         int lower, int group) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // Normally one tile per workgroup (gridDim.x == ntiles).  GPRC_PERSIST=N launches N workgroups that walk the tile list
  // with stride N instead (N a multiple of 8 keeps the XCD-contiguous id ranges); measured 4% slower at N=512.
  const unsigned ntiles = (unsigned)tiles_m * (unsigned)tiles_n;
  const int width = group * tiles_n;
  for (unsigned t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const unsigned id = xcd_remap(t, ntiles);
    const int g = id / width, first_m = g * group;
    const int gsize = (tiles_m - first_m < group) ? (tiles_m - first_m) : group;
    const int tr = first_m + (int)(id % width) % gsize;
    const int tc = (int)(id % width) / gsize;
    if (lower && tc > tr) continue;
    gemm_tile_128<false>(C + (int64_t)tr * 128 + (int64_t)tc * 128 * ldc, ldc, A + (int64_t)tr * 128, lda,
                         B + (int64_t)tc * 128, ldb, K, smem);
    __syncthreads();  // every wave has left the tile (LDS reads done) before the next tile's first DMA lands
  }
}

// X[M x 128] := X * W^T (W = inverse of the diagonal block, lower triangular), in place: a workgroup
// owns a full 128-row strip, and every load of it precedes the epilogue stores.
template <bool SSQ>
__global__ __launch_bounds__(256, 2) void trsm_panel_kernel(double* X, int64_t ldx, const double* winv, double* ssq) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Xs = X + (int64_t)blockIdx.x * 128;
  gemm_tile_128<true, false, false, SSQ>(Xs, ldx, Xs, ldx, winv, 128, 128, smem, 0, 0, 0, SSQ ? ssq + (int64_t)blockIdx.x * 128 : nullptr);
}

// Trailing update over the packed layout: for every target panel q in {q_begin, q_begin+stride, ..}
// C_q -= L_p[rows of q] * L_p[rows of q's diagonal block]^T, lower tiles only.
__global__ __launch_bounds__(256, 2) void trailing_kernel(double* packed, int64_t n_pad, int p, int q_begin, int q_stride,
                                                          int n_targets, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int P = (int)(n_pad / NB);
  constexpr int DIAG_TILES = TPP * (TPP + 1) / 2;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {  // persistent, as gemm_nt_kernel
    int id = (int)xcd_remap((unsigned)t, (unsigned)ntiles);
    // locate the target panel: panel q holds TPP*TPP*(P-q) - TPP*(TPP-1)/2 lower tiles
    int q = q_begin, s = 0;
    for (; s < n_targets; ++s, q += q_stride) {
      const int tq = TPP * TPP * (P - q) - TPP * (TPP - 1) / 2;
      if (id < tq) break;
      id -= tq;
    }
    if (s >= n_targets) continue;
    int tr, tc;
    if (id < DIAG_TILES) {  // the diagonal NB x NB block: lower tiles (0,0) (1,0) (1,1) (2,0) ...
      tr = 0;
      while ((tr + 1) * (tr + 2) / 2 <= id) ++tr;
      tc = id - tr * (tr + 1) / 2;
    } else {
      tr = TPP + (id - DIAG_TILES) / TPP;
      tc = (id - DIAG_TILES) % TPP;
    }
    const int64_t ldp = panel_ld(n_pad, p), ldq = panel_ld(n_pad, q);
    const double* Lp = packed + panel_offset(n_pad, p) + (int64_t)(q - p) * NB;  // row q*NB of panel p
    double* Cq = packed + panel_offset(n_pad, q);
    gemm_tile_128<false>(Cq + (int64_t)tr * 128 + (int64_t)tc * 128 * ldq, ldq, Lp + (int64_t)tr * 128, ldp,
                         Lp + (int64_t)tc * 128, ldp, NB, smem);
    __syncthreads();
  }
}


// Left-looking step of the predict solve: the columns of panels [j, j + G) of vt receive, in ONE pass with the C tile
// held in the accumulators, everything the right-looking form would have subtracted panel by panel:
//   vt[:, j NB : (j+G) NB] -= vt[:, 0 : j NB] * L[j NB : (j+G) NB, 0 : j NB]^T          (K = j NB).
// Same products in the same order (k ascending from the loaded C value), so the result is bit-identical; what
// changes is that a C tile is loaded and stored once instead of j times -- the per-tile prologue (C preload + first
// DMA, ~7 % of a K = 512 tile during which the tile's waves issue no MFMA) is paid once per j NB of K.
// [kp0, kp1): the source panels of this launch (kp0 = 0, kp1 = j: the whole pass; GPRC_KCHUNK splits it into K-chunks)
// tri_row0 >= 0 (fit()'s gradient: the rows of vt are rows tri_row0, tri_row0 + 1, ... of the IDENTITY, so row i is zero left of
// column tri_row0 + i and stays zero there): a tile's pass starts at its first row's column instead of column 0 -- the skipped
// products are exact zeros, so the bits are those of the full pass -- and a tile whose rows start right of the pass has nothing to do.
__global__ __launch_bounds__(256, 2) void solve_left_kernel(double* vt, int64_t ldv, const double* packed, int64_t n_pad, int j,
                                                            int tiles_m, int tiles_n, int group, int kp0, int kp1, int64_t tri_row0) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const unsigned ntiles = (unsigned)tiles_m * (unsigned)tiles_n;
  // (the triangular form keeps block id -> tile: a tile's length falls with its row, and XCD-contiguous id ranges would hand one XCD
  //  all the long tiles and another none -- measured 27 TFLOP/s; round-robin over the XCDs every one gets the same mix)
  const unsigned id = tri_row0 >= 0 ? blockIdx.x : xcd_remap(blockIdx.x, ntiles);
  const int width = group * tiles_n;
  const int g = id / width, first_m = g * group;
  const int gsize = (tiles_m - first_m < group) ? (tiles_m - first_m) : group;
  const int tr = first_m + (int)(id % width) % gsize;
  const int tc = (int)(id % width) / gsize;
  const int64_t col = (int64_t)j * NB + (int64_t)tc * 128;
  int kt_first = kp0 * (NB / 16);
  const int kt_end = kp1 * (NB / 16);
  if (tri_row0 >= 0) {
    const int64_t first_col = tri_row0 + (int64_t)tr * 128;      // a multiple of 128: whole k-tiles
    if (first_col / 16 > kt_first) kt_first = (int)(first_col / 16);
    if (kt_first >= kt_end) return;
  }
  gemm_tile_128<false, true>(vt + (int64_t)tr * 128 + col * ldv, ldv, vt + (int64_t)tr * 128 + (int64_t)kt_first * 16 * ldv, ldv, packed, n_pad,
                             (kt_end - kt_first) * 16, smem, col, 0, kt_first);
}


// The predict's in-panel solve in ONE launch: panel p of vt := vt L^-T once everything left of the panel has been applied.
// Per 128-row strip of vt the four 128-column sub-steps are C(.,j) -= vt(., panel columns < j) L(j, < j)^T  (K = 128 j), then
// C(.,j) := C(.,j) Winv_j^T [+ the per-row sums of squares of the finished block].  L and Winv are final, so strips are
// independent: what used to be seven dependent launches per panel (each draining the GPU, each latency-bound for the 64-tile
// slices of an 8-rank run) is one workgroup per strip running its seven tiles back to back.  The same gemm_tile_128 calls in the
// same order per strip: bit-identical.
template <bool SSQ>
__global__ __launch_bounds__(256, 2) void solve_panel_fused_kernel(double* vt, int64_t ldv, const double* packed, int64_t n_pad, int p,
                                                                   const double* winv, double* sspart, int64_t m_pad) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int64_t ld = panel_ld(n_pad, p);
  const double* pan = packed + panel_offset(n_pad, p);
  double* strip = vt + (int64_t)blockIdx.x * 128 + (int64_t)p * NB * ldv;   // my 128 rows, first column of the panel
  for (int j = 0; j < TPP; ++j) {
    double* C = strip + (int64_t)j * NBI * ldv;
    if (j > 0) {
      gemm_tile_128<false>(C, ldv, strip, ldv, pan + (int64_t)j * NBI, ld, j * NBI, smem);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the updated block is re-read (LDS-DMA) by all four waves
      __syncthreads();
    }
    const double* wblk = winv + ((int64_t)p * TPP + j) * NBI * NBI;
    double* ssq = SSQ ? sspart + ((int64_t)p * TPP + j) * m_pad + (int64_t)blockIdx.x * 128 : nullptr;
    gemm_tile_128<true, false, false, SSQ>(C, ldv, C, ldv, wblk, 128, 128, smem, 0, 0, 0, ssq);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // block j is an operand of the next sub-step's update
    __syncthreads();
  }
}

// the same pass on 256 x 128 macro-tiles (GPRC_TILE256=1; m_pad a multiple of 256)
__global__ __launch_bounds__(512) void solve_left_kernel256(double* vt, int64_t ldv, const double* packed, int64_t n_pad, int j,
                                                            int tiles_m, int tiles_n, int group, int kp0, int kp1) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const unsigned ntiles = (unsigned)tiles_m * (unsigned)tiles_n;
  const unsigned id = xcd_remap(blockIdx.x, ntiles);
  const int width = group * tiles_n;
  const int g = id / width, first_m = g * group;
  const int gsize = (tiles_m - first_m < group) ? (tiles_m - first_m) : group;
  const int tr = first_m + (int)(id % width) % gsize;
  const int tc = (int)(id % width) / gsize;
  const int64_t col = (int64_t)j * NB + (int64_t)tc * 128;
  gemm_tile_256x128_seg(vt + (int64_t)tr * 256 + col * ldv, ldv, vt + (int64_t)tr * 256 + (int64_t)kp0 * NB * ldv, ldv, packed, n_pad,
                        (kp1 - kp0) * NB, smem, col, kp0 * (NB / 16));
}

// Trailing update by a RANGE of source panels [p_begin, p_end) in one pass: every lower tile of the target panels
// q_begin, q_begin + q_stride, ... (n_targets of them) receives
//   A[R.., C..] -= L[R.., p_begin NB : p_end NB] * L[C.., p_begin NB : p_end NB]^T,
// both operand strips walking through the packed panels.  Same products, same order as the single-panel passes
// p = p_begin .. p_end - 1 (k ascending from the loaded C value): bit-identical, one C load/store and one tile prologue
// instead of p_end - p_begin.  p_begin = 0 is the left-looking sweep of one GPU; the multi-rank driver uses it to apply
// the panels it has received in batches.
__global__ __launch_bounds__(256, 2) void trailing_range_kernel(double* packed, int64_t n_pad, int p_begin, int p_end, int q_begin,
                                                                int q_stride, int n_targets, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int P = (int)(n_pad / NB);
  constexpr int DIAG_TILES = TPP * (TPP + 1) / 2;
  int id = (int)xcd_remap(blockIdx.x, (unsigned)ntiles);
  int q = q_begin, s = 0;
  for (; s < n_targets; ++s, q += q_stride) {
    const int tq = TPP * TPP * (P - q) - TPP * (TPP - 1) / 2;
    if (id < tq) break;
    id -= tq;
  }
  if (s >= n_targets) return;
  int tr, tc;
  if (id < DIAG_TILES) {
    tr = 0;
    while ((tr + 1) * (tr + 2) / 2 <= id) ++tr;
    tc = id - tr * (tr + 1) / 2;
  } else {
    tr = TPP + (id - DIAG_TILES) / TPP;
    tc = (id - DIAG_TILES) % TPP;
  }
  const int64_t ldq = panel_ld(n_pad, q);
  double* Cq = packed + panel_offset(n_pad, q);
  const int64_t row = (int64_t)q * NB + (int64_t)tr * 128, col = (int64_t)q * NB + (int64_t)tc * 128;
  gemm_tile_128<false, true, true>(Cq + (int64_t)tr * 128 + (int64_t)tc * 128 * ldq, ldq, packed, n_pad, packed, n_pad,
                                   (p_end - p_begin) * NB, smem, col, row, p_begin * (NB / 16));
}


// ------------------------------------------------------------------------------------------------
// One launch per 512-column panel: diagonal blocks, panel solves and in-panel updates of all four 128-column sub-steps.
//
// The launch-per-stage form (factor_subpanel: update K = 128 j -> diagonal block -> panel solve, 4 x 3 dependent launches
// per panel) leaves the GPU almost empty while each stage drains: ~570 us per panel, which is most of the fit at
// n = 8192..16384 and most of one rank's share of an 8-rank sweep.  Here the stages of different 128-row strips overlap
// and only the true dependencies remain, carried by agent-scope flags inside one grid of 512-thread workgroups:
//   factor workgroup (ticket 0, 8 waves)   the whole critical chain, with no hand-off on it:
//                                           for j = 0..3: factor + invert block (j, j) in LDS (potf2_blocked_body<8>),
//                                           publish W_j; then, itself, the two tiles the NEXT diagonal block waits for:
//                                           L(j+1, j) = C'(j+1, j) Winv_j^T (publish R_{j+1}) and
//                                           C(j+1, j+1) -= L(j+1, j) L(j+1, j)^T
//   diagonal strips s = 1..3 (tickets 2..4, one 4-wave team): blocks (s, 0..s-2) like any strip; then everything of
//                                           blocks (s, s-1) and (s, s) that does not need L(s-1, s-1) yet -- the updates
//                                           with the columns left of block s-1 -- and publish E_s
//   other strips, two per workgroup (two 4-wave teams with their own LDS halves, in lockstep: the same flags, the same
//                                           barrier count): for j = 0..3: wait R_j (j > 0), C(s,j) -= L(s,<j) L(j,<j)^T,
//                                           wait W_j, C(s,j) := C(s,j) Winv_j^T
// Every tile is a gemm_tile_128 call with the same operands and the same K order as in the launch-per-stage form (an update
// split in two calls continues the same accumulator chain from the stored value), and the diagonal block goes through the
// same 16x16-block arithmetic: the results are bit-identical.
// No deadlock, whatever the dispatch order, the number of resident workgroups (149 KB of LDS: one per CU) or what else runs
// on the GPU: roles are dealt by a ticket counter in START order; every flag is published by one of the first five tickets
// (factor role and diagonal strips); those five wait only on each other, in an order without cycles (the factor role
// publishes W_0, R_1, W_1, R_2 ... before it waits for the E_s that needs them); and a workgroup with a later ticket can
// only be running -- and spinning -- when all five earlier ones have started.
// Hand-off protocol: MI355X_MICROARCH.md "inter-workgroup visibility" (plain stores, every wave's vmcnt(0), workgroup
// barrier, one lane's agent release fence + vmcnt(0), relaxed agent store of the flag; one lane polls with relaxed agent
// loads, agent acquire fence + vmcnt(0), workgroup barrier, plain / LDS-DMA loads).
// ------------------------------------------------------------------------------------------------
struct PanelSync { int ticket; int failed; int W[4]; int E[4]; int R[4]; int LA; int SU; };   // 16 ints, zeroed before the launch
static_assert(TPP <= 4, "PanelSync holds four flags of each kind and LA four 8-bit fields: the factor service is written for NB = 4 x 128");
// Bound of every device-side dependency wait: WALL time (s_memrealtime, 100 MHz), not a poll count -- a busy, shared GPU slows the
// polls down but must not shorten the patience.  Legitimate waits are below 10 ms (one trailing update at n <= 24576).
constexpr unsigned long long WAIT_LIMIT_TICKS = 400000000ULL;   // 4 s
// (factor service only -- LA: the look-ahead strips' finished blocks, one 8-bit count per 128-column sub-step j (a strip adds
//  1 << 8 j after block (s, j)): field j = TPP when all four strips have finished sub-step j.  A single sum over the sub-steps --
//  the round-2 form -- reads "4 (j + 1)" also when one strip is a sub-step ahead and another one behind, which happens as soon as
//  the service's workgroups do not start together (another context's kernels on the GPU): profiles/r03_la_counter_race.txt.
//  A strip goes through its blocks in order, so field TPP - 1 = TPP -- LA >= TPP << 24 -- means rows [NB, 2 NB) are final;
//  E[0], E[1]: blocks (s, j <= s-2) the diagonal strips s = 2, 3 have finished;
//  SU: the split chain's counters, eight 4-bit fields with 3-bit counts -- field j: 32-row slices of L(j, j-1) that are final (the chain
//  helpers' solve phase), field 4 + j: slices of block (j, j) that have received L(j, j-1) (their update phase); each reaches
//  CHAIN_HELPERS.  j = 0 stands for the step BEHIND the panel: slices of L(4, 3), and of tile (0, 0) of the next diagonal block; bit 3 of
//  field 0: that tile has received the k-chunks 0..2 (role SERVICE_D0).  R[0]: block (4, 3) is ready for its solve (look-ahead strip 4))

// Who gave up, and on what: every wait of a factorisation that runs out its bound -- and every wait that was still unsatisfied when it
// saw that somebody else had (site + 10) -- leaves a record for the host (wait_timeout_report, gprc_prof_wait_timeout): up to
// WAIT_DIAG_RECORDS records of 8 ints behind a counter -- [0] site (1 flag, 2 count, 3 field, 4 chain, 5 sweep, 6 gate), [1] blockIdx.x,
// [2] gridDim.x, [3] the value needed, [4] the value seen, [5] the awaited word's index inside its panel's PanelSync (or its distance
// from it), [6] threads per workgroup, [7] low 32 bits of the PanelSync's address (which panel)
constexpr int WAIT_DIAG_RECORDS = 48;
__device__ int g_wait_diag[8 * (WAIT_DIAG_RECORDS + 1)];
__device__ __attribute__((noinline)) void wait_diag(int site, int need, int seen, const int* word, const void* sy) {
  const int k = atomicAdd(&g_wait_diag[0], 1);
  if (k >= WAIT_DIAG_RECORDS) return;
  int* r = g_wait_diag + 8 * (k + 1);
  r[0] = site; r[1] = (int)blockIdx.x; r[2] = (int)gridDim.x; r[3] = need; r[4] = seen;
  r[5] = sy ? (int)(word - static_cast<const int*>(sy)) : -1; r[6] = (int)blockDim.x; r[7] = (int)(uintptr_t)sy;
}

// relaxed: the caller is throughput work (a strip riding in the sweep kernel), not a role of the chain: it looks at the flag once per
// microsecond instead of every ~50 ns.  Hundreds of workgroups polling flags and counters at full rate slow every device-scope access
// down -- the chain's own publishes and polls included (measured with the round-3 tile core, whose faster tiles left the sweep's
// workgroups waiting longer: ticket atomics 6 -> 16 us, n = 8192 factorisation 6.27 -> 6.71 ms; tools/sweep_prof.py).
__device__ __forceinline__ void poll_pause(bool relaxed) {
  if (relaxed) __builtin_amdgcn_s_sleep(32);      // 32 x 64 clocks ~ 0.9 us
  else __builtin_amdgcn_s_sleep(2);
}

__device__ __forceinline__ void panel_flag_wait(int* flag, PanelSync* sy, int* info, bool relaxed = false) {   // the whole workgroup calls it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's own stores: its team-mates re-read them after the barrier
  if (threadIdx.x == 0) {
    int spins = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      poll_pause(relaxed);
      if ((++spins & 255) != 0) continue;
      // somebody has already given up (e.g. a profiler that serialises dispatches keeps producer and consumer kernels apart): every
      // later wait of the factorisation returns at once instead of running out its own bound
      if (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == GPRC_INFO_WAIT_TIMEOUT) { wait_diag(11, 1, 0, flag, sy); break; }
      // exit condition every wave reaches (a producer that never publishes must not leave this workgroup spinning on the GPU
      // for ever): after WAIT_LIMIT_TICKS of wall time give up, let the grid drain, and tell the host through the ONE word it
      // always reads after a factorisation -- info = GPRC_INFO_WAIT_TIMEOUT (< 0; LAPACK infos are > 0)
      if (__builtin_amdgcn_s_memrealtime() - t0 > WAIT_LIMIT_TICKS) {
        wait_diag(1, 1, 0, flag, sy);
        __hip_atomic_store(&sy->failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicExch(info, GPRC_INFO_WAIT_TIMEOUT);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

__device__ __forceinline__ void panel_flag_publish(int* flag) {               // the whole workgroup calls it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ __forceinline__ void panel_ready_wait(int* ctr, int need, PanelSync* sy, int* info, bool relaxed = false) {   // the whole workgroup calls it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) {
    int spins = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
      poll_pause(relaxed);
      if ((++spins & 255) != 0) continue;
      if (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == GPRC_INFO_WAIT_TIMEOUT) {
        wait_diag(12, need, __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), ctr, sy);
        break;
      }
      if (__builtin_amdgcn_s_memrealtime() - t0 > WAIT_LIMIT_TICKS) {   // bounded in wall time (see panel_flag_wait)
        wait_diag(2, need, __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), ctr, sy);
        __hip_atomic_store(&sy->failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicExch(info, GPRC_INFO_WAIT_TIMEOUT);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

// waits until the 8-bit field at `shift` of *ctr has reached `need` (the look-ahead strips' per-sub-step counts in LA)
__device__ __forceinline__ void panel_field_wait(int* ctr, int shift, int need, PanelSync* sy, int* info) {   // the whole workgroup calls it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) {
    int spins = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (((__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> shift) & 0xff) < need) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 255) != 0) continue;
      if (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == GPRC_INFO_WAIT_TIMEOUT) {
        wait_diag(13, need << shift, __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), ctr, sy);
        break;
      }
      if (__builtin_amdgcn_s_memrealtime() - t0 > WAIT_LIMIT_TICKS) {   // bounded in wall time (see panel_flag_wait)
        wait_diag(3, need << shift, __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), ctr, sy);
        __hip_atomic_store(&sy->failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicExch(info, GPRC_INFO_WAIT_TIMEOUT);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

__device__ __forceinline__ void panel_count_publish(int* ctr, int add) {      // the whole workgroup calls it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(ctr, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Workgroup barriers gemm_tile_128 executes for a K-deep tile WITHOUT the SSQ epilogue: one after the first DMA and one
// per further k-tile.  Waves of a workgroup that sit a tile out call this so that the barrier counts match.
__device__ __forceinline__ void gemm_tile_shadow_barriers(int K) {
  const int KT = K / G_KB;
  for (int b = 0; b < KT; ++b) __builtin_amdgcn_s_barrier();
}

// ------------------------------------------------------------------------------------------------
// The SPLIT chain (factor service only).  Between two diagonal blocks of a panel the chain runs two 128 x 128 x 128 tiles -- the solve
// L(j+1, j) = C'(j+1, j) Winv_j^T and the update C(j+1, j+1) -= L(j+1, j) L(j+1, j)^T -- and on ONE CU each is bound by that CU's MFMA
// rate (13.7 us of MFMAs + 3.5 us of prologue: 17.4 and 20.5-24 us measured, profiles/r03_chain_traces_after_diag16.txt; a second team
// on the same CU changes nothing).  Here CHAIN_HELPERS = 4 resident workgroups (8 waves each, a CU each) take 32 ROWS of both tiles
// each: a quarter of the MFMAs (3.4 us), operands loaded in one go (the slice's 32 x 128 A rows through LDS, every wave's 16 B rows
// straight into registers), results stored write-through (sc1) and handed on by counters -- solve slices to each other (the update
// needs all of L(j+1, j)), update slices to the factor role, which then loads the block itself (potf2_blocked_body, not preloaded).
// Every output element is still acc = C (or 0), then for the k-steps 0..31 ascending acc = v_mfma_f64_16x16x4(B strip value, A strip
// value, acc) with the update's negate-A bit: the same operand roles, k order and accumulator start as gemm_tile_128 -- identical bits.
// ------------------------------------------------------------------------------------------------
#ifdef GPRC_CHAIN_PROF
constexpr int GPRC_CHAIN_PROF_PANEL = GPRC_CHAIN_PROF;
#else
constexpr int GPRC_CHAIN_PROF_PANEL = -1;
#endif
constexpr int CHAIN_HELPERS = 4;                  // 32-row slices of a 128-row block
constexpr int PANEL_DIAG_TILES = TPP * (TPP + 1) / 2;   // lower 128 x 128 tiles of a panel's diagonal block
constexpr int PANEL_LA_TILES = TPP * TPP;             // tiles of a panel's rows [NB, 2 NB)
constexpr int AUX_STRIDE = 2 * TPP + PANEL_LA_TILES;  // ints per panel in the sync block's last array: 2 TPP early-chunk flags, then the slice counts of the sweep's head tiles
constexpr int CHAIN_LDS_LD = 48;                  // doubles per k-slice of the A image: 32 rows + 16 pad (consecutive k-slices 32 banks apart)
static_assert(CHAIN_HELPERS * 32 == 128 && CHAIN_HELPERS < 8, "SU holds 3-bit counts of 32-row slices (bit 3 of field 0 is a flag)");

#ifdef GPRC_CHAIN_PROF
// measurement build: s_memrealtime stamps (100 MHz) of panel GPRC_CHAIN_PROF's chain -- [0..31] helper 0 (8 per sub-step: W seen,
// operands in, MFMAs done, stored + counted, S seen, operands in, MFMAs done, stored + counted), [32 + 4 j ..] the factor role
// (U seen, potf2 done, W published)
__device__ unsigned long long g_chain_prof[64];
#define CHAIN_STAMP(on, k) do { if ((on) && threadIdx.x == 0) g_chain_prof[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CHAIN_STAMP(on, k)
#endif

// waits until ((*a >> sa) & ma) >= va and (b == null or ((*b >> sb) & mb) >= vb); the whole workgroup calls it; the polls fly together
__device__ __forceinline__ void chain_wait2(int* a, int sa, int ma, int va, int* b, int sb, int mb, int vb, PanelSync* sy, int* info) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) {
    int spins = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
      const int xa = (__hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> sa) & ma;
      const int xb = b ? (__hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> sb) & mb : vb;
      if (xa >= va && xb >= vb) break;
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255) != 0) continue;
      if (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == GPRC_INFO_WAIT_TIMEOUT) {
        if (xa < va) wait_diag(14, va << sa, __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a, sy);
        else wait_diag(14, vb << sb, __hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b, sy);
        break;
      }
      if (__builtin_amdgcn_s_memrealtime() - t0 > WAIT_LIMIT_TICKS) {   // bounded in wall time (see panel_flag_wait)
        if (xa < va) wait_diag(4, va << sa, __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a, sy);
        else wait_diag(4, vb << sb, __hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b, sy);
        __hip_atomic_store(&sy->failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicExch(info, GPRC_INFO_WAIT_TIMEOUT);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

// C (32 rows x 128 columns, column-major) = (SET ? 0 : C) -/+ A (32 x 128) B (128 x 128)^T on 512 threads: wave w the columns
// [16 w, 16 w + 16).  lds: 128 x CHAIN_LDS_LD doubles.  C may be A (the solve in place: every A element is in LDS before the first
// store).  Stores are write-through; the caller publishes (every wave's vmcnt(0), barrier, counter).
template <bool SET>
__device__ __forceinline__ void chain_slice_32(double* C_, int64_t ldc, const double* A_, int64_t lda, const double* B_, int64_t ldb, double* lds_,
                                               bool prof = false, int stamp0 = 0) {
  // explicit address spaces: inside a non-inlined role the pointers are generic, and FLAT loads would tie the LDS waits to the
  // outstanding global loads (a flat instruction counts in vmcnt and lgkmcnt)
  typedef __attribute__((address_space(1))) double gdouble;
  typedef __attribute__((address_space(3))) double ldouble;
  gdouble* C = (gdouble*)C_;
  const gdouble* A = (const gdouble*)A_;
  const gdouble* B = (const gdouble*)B_;
  ldouble* lds = (ldouble*)lds_;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fk = lane >> 4, fr = lane & 15;
  constexpr int NEG = SET ? 0 : 1;
  // the A slice: thread t brings rows [8 (t & 3), +8) of k-slice t >> 2
  const gdouble* asrc = A + 8 * (t & 3) + (int64_t)(t >> 2) * lda;
  double av[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) av[i] = asrc[i];
  double4_t acc[2];
  gdouble* Cw = C + fr + (int64_t)(16 * wave + fk) * ldc;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[m][r] = SET ? 0.0 : Cw[16 * m + (int64_t)(4 * r) * ldc];
  // this wave's 16 B rows, all 32 k-steps: one double per lane and step
  const gdouble* bsrc = B + (16 * wave + fr) + (int64_t)fk * ldb;
  double b[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) b[s] = bsrc[(int64_t)(4 * s) * ldb];
  ldouble* adst = lds + (t >> 2) * CHAIN_LDS_LD + 8 * (t & 3);
#pragma unroll
  for (int i = 0; i < 8; ++i) adst[i] = av[i];
  __syncthreads();
  const ldouble* ap = lds + fk * CHAIN_LDS_LD + fr;
  double a0[2], a1[2];   // operand reads two k-steps ahead
  a0[0] = ap[0]; a1[0] = ap[16];
  a0[1] = ap[4 * CHAIN_LDS_LD]; a1[1] = ap[4 * CHAIN_LDS_LD + 16];
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    const double x0 = a0[s & 1], x1 = a1[s & 1];
    if (s + 2 < 32) { a0[s & 1] = ap[4 * (s + 2) * CHAIN_LDS_LD]; a1[s & 1] = ap[4 * (s + 2) * CHAIN_LDS_LD + 16]; }
    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[s], x0, acc[0], 0, 0, NEG);
    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[s], x1, acc[1], 0, 0, NEG);
#ifdef GPRC_CHAIN_PROF
    if (s == 0) CHAIN_STAMP(prof, stamp0);       // the first MFMAs are issued: A image and b[0] are in
#endif
  }
#ifdef GPRC_CHAIN_PROF
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  if (prof && threadIdx.x == 0) { double sink = acc[0][0] + acc[1][3]; asm volatile("" :: "v"(sink)); g_chain_prof[stamp0 + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) __hip_atomic_store(&Cw[16 * m + (int64_t)(4 * r) * ldc], acc[m][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the slice is stored: every wave's vmcnt(0), barrier, then one lane adds `add` to *ctr (write-through payload: no write-back).
// Returns (to thread 0 only) the counter's previous value.
__device__ __forceinline__ int chain_publish(int* ctr, int add) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  return threadIdx.x == 0 ? __hip_atomic_fetch_add(ctr, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
}

// Chain helper q of one panel (512 threads): rows [32 q, 32 q + 32) of the solve tile of every sub-step and of EVERY update of the
// diagonal blocks (1,1), (2,2), (3,3) -- the one the next potf2 waits for, and, in the 35 us that potf2 then runs, the updates of the
// later diagonal blocks with the column just finished (the diagonal strips' own share in the unsplit form: with the two chain tiles
// down from 38 to 17 us the chain waited 29 us for strip 3's K = 256 tiles, tools/chain_prof.py).  A helper touches only its own 32
// rows of those blocks, k-chunk after k-chunk in order: no flag between them, and the same accumulator chain through memory.
// strip_progress: E[0] / E[1], the finished blocks (s, j <= s-2) of the diagonal strips s = 2, 3.
// dnext (null: no next panel in the launch's group): tile (0, 0) of the NEXT panel's diagonal block, leading dimension ldn.  The path from
// W_3 to the next chain is the same pattern once more -- L(4, 3) = C'(4, 3) Winv_3^T, D(0, 0) -= L(4, 3) L(4, 3)^T, potf2 -- and was a
// 17-us solve on the look-ahead strip's CU plus a 20-us tile on a next-diagonal-block role's; the helpers do both in slices: the
// look-ahead strip TPP hands block (4, 3) over when it has received the panel's earlier columns (R[0]), role SERVICE_D0 hands tile
// (0, 0) over when it has received the k-chunks 0..2 (bit 3 of SU), the last solve slice counts the
// block into LA for that strip, the last update slice counts the tile into ready_next, and the next factor role starts on the
// update slices' count (field 4 of SU) instead of waiting for the other nine tiles, which it does not read.
// ready_mine (null: first panel of the launch): the count of finished tiles of THIS panel's diagonal block -- the helpers' first solve
// reads tile (1, 0), which the factor role no longer waits for.
__device__ __attribute__((noinline)) void panel_chain_helper_role(double* sm, double* pan, int64_t ld, double* wp, int* info, PanelSync* sy, int q,
                                                                  bool prof, double* dnext, int64_t ldn, int* ready_next, int* ready_mine) {
  for (int j = 0; j + 1 < TPP; ++j) {
    double* Cn = pan + (int64_t)(j + 1) * NBI + (int64_t)j * NBI * ld;            // block (j+1, j)
    double* Dn = pan + (int64_t)(j + 1) * NBI + (int64_t)(j + 1) * NBI * ld;      // block (j+1, j+1)
    const int fs = 4 * (j + 1), fu = 16 + 4 * (j + 1);
    // solve: W_j published; j > 0: block (j+1, j) has received the columns left of block j (E = 1, the diagonal strip j+1)
    chain_wait2(&sy->W[j], 0, 1, 1, j > 0 ? &sy->E[j + 1] : ready_mine, 0, 0xffff, j > 0 ? 1 : PANEL_DIAG_TILES, sy, info);
    CHAIN_STAMP(prof, 8 * j);
    chain_slice_32<true>(Cn + 32 * q, ld, Cn + 32 * q, ld, wp + (int64_t)j * NBI * NBI, 128, sm, prof, 8 * j + 1);
    const int before = chain_publish(&sy->SU, 1 << fs);
    CHAIN_STAMP(prof, 8 * j + 3);
    // the last slice completes L(j+1, j): R_{j+1} for the strips (every slice was written through and had landed before its count)
    if (threadIdx.x == 0 && ((before >> fs) & 7) == CHAIN_HELPERS - 1) __hip_atomic_store(&sy->R[j + 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // update: all four slices of L(j+1, j)
    chain_wait2(&sy->SU, fs, 7, CHAIN_HELPERS, nullptr, 0, 0, 0, sy, info);
    CHAIN_STAMP(prof, 8 * j + 4);
    chain_slice_32<false>(Dn + 32 * q, ld, Cn + 32 * q, ld, Cn, ld, sm, prof, 8 * j + 5);
    (void)chain_publish(&sy->SU, 1 << fu);
    CHAIN_STAMP(prof, 8 * j + 7);
    // while the factor role is busy with block (j+1, j+1): column j into the later diagonal blocks
    for (int s = j + 2; s < TPP; ++s) {
      double* Ls = pan + (int64_t)s * NBI + (int64_t)j * NBI * ld;                // L(s, j), solved by diagonal strip s
      chain_wait2(&sy->E[s - 2], 0, 0xffff, j + 1, nullptr, 0, 0, 0, sy, info);
      chain_slice_32<false>(pan + (int64_t)s * NBI + (int64_t)s * NBI * ld + 32 * q, ld, Ls + 32 * q, ld, Ls, ld, sm);
    }
  }
  if (dnext) {
    double* Cn = pan + (int64_t)TPP * NBI + (int64_t)(TPP - 1) * NBI * ld;       // block (4, 3)
    chain_wait2(&sy->W[TPP - 1], 0, 1, 1, &sy->R[0], 0, 1, 1, sy, info);
    CHAIN_STAMP(prof, 24);
    chain_slice_32<true>(Cn + 32 * q, ld, Cn + 32 * q, ld, wp + (int64_t)(TPP - 1) * NBI * NBI, 128, sm, prof, 25);
    const int b0 = chain_publish(&sy->SU, 1);
    CHAIN_STAMP(prof, 27);
    if (threadIdx.x == 0 && (b0 & 7) == CHAIN_HELPERS - 1) __hip_atomic_fetch_add(&sy->LA, 1 << (8 * (TPP - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    chain_wait2(&sy->SU, 0, 7, CHAIN_HELPERS, &sy->SU, 3, 1, 1, sy, info);   // all slices of L(4, 3); tile (0, 0) has its k-chunks 0..2 (bit 3)
    CHAIN_STAMP(prof, 28);
    chain_slice_32<false>(dnext + 32 * q, ldn, Cn + 32 * q, ld, Cn, ld, sm, prof, 29);
    const int b1 = chain_publish(&sy->SU, 1 << 16);
    CHAIN_STAMP(prof, 31);
    if (threadIdx.x == 0 && ((b1 >> 16) & 7) == CHAIN_HELPERS - 1) __hip_atomic_fetch_add(ready_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The factor role beside the chain helpers: the four diagonal blocks, each loaded when its update slices are complete.
__device__ __forceinline__ void panel_factor_role_split(double* sm, double* pan, int64_t ld, double* wp, int* info, int p, PanelSync* sy) {
  for (int j = 0; j < TPP; ++j) {
    if (j > 0) chain_wait2(&sy->SU, 16 + 4 * j, 7, CHAIN_HELPERS, nullptr, 0, 0, 0, sy, info);
    CHAIN_STAMP(p == GPRC_CHAIN_PROF_PANEL, 32 + 4 * j);
    potf2_blocked_body<8>(sm, pan + (int64_t)j * NBI + (int64_t)j * NBI * ld, ld, wp + (int64_t)j * NBI * NBI, info, p * NB + j * NBI, nullptr, false);
    CHAIN_STAMP(p == GPRC_CHAIN_PROF_PANEL, 33 + 4 * j);
    panel_flag_publish(&sy->W[j]);
    CHAIN_STAMP(p == GPRC_CHAIN_PROF_PANEL, 34 + 4 * j);
  }
}

// trace (may be null): the factor role's lane 0 leaves s_memrealtime stamps (100 MHz) of its stages there -- measurement only
#define PANEL_STAMP(k) do { if (trace && t == 0) trace[k] = __builtin_amdgcn_s_memrealtime(); } while (0)

// The factor role of one panel: 512 threads (8 waves), sm = PB_SMEM_DOUBLES doubles of LDS.
__device__ __forceinline__ void panel_factor_role(double* sm, double* pan, int64_t ld, double* wp, int* info, int p, PanelSync* sy,
                                                  unsigned long long* trace, unsigned long long* ptrace = nullptr) {
  const int t = threadIdx.x, team = t >> 8, tid = t & 255;
  PANEL_STAMP(0);
  for (int j = 0; j < TPP; ++j) {
    potf2_blocked_body<8>(sm, pan + (int64_t)j * NBI + (int64_t)j * NBI * ld, ld, wp + (int64_t)j * NBI * NBI, info, p * NB + j * NBI,
                          (ptrace && j == 1) ? ptrace : nullptr, j > 0);   // blocks 1..3 arrive in LDS from the update tile below
    PANEL_STAMP(1 + 6 * j);
    panel_flag_publish(&sy->W[j]);               // (its vmcnt(0) + barrier also make L(j,j) / Winv_j visible to this workgroup's own DMA)
    PANEL_STAMP(2 + 6 * j);
    if (j + 1 == TPP) break;
    // the two tiles the next diagonal block is waiting for, by team 0 (team 1 shadows the barriers)
    double* Cn = pan + (int64_t)(j + 1) * NBI + (int64_t)j * NBI * ld;            // block (j+1, j)
    double* Dn = pan + (int64_t)(j + 1) * NBI + (int64_t)(j + 1) * NBI * ld;      // block (j+1, j+1)
    if (j > 0) panel_ready_wait(&sy->E[j + 1], 1, sy, info);                      // block (j+1, j): its update with the columns left of block j
    PANEL_STAMP(3 + 6 * j);
    if (team == 0) gemm_tile_128<true, false, false, false, false, false>(Cn, ld, Cn, ld, wp + (int64_t)j * NBI * NBI, 128, 128, sm, 0, 0, 0, nullptr, tid);
    else gemm_tile_shadow_barriers(128);
    PANEL_STAMP(4 + 6 * j);
    panel_flag_publish(&sy->R[j + 1]);           // rows of strip j+1 left of its diagonal block are final
    PANEL_STAMP(5 + 6 * j);
    if (j > 0) panel_ready_wait(&sy->E[j + 1], 2, sy, info);                      // block (j+1, j+1): likewise
    // the updated block goes to memory (its upper triangle is part of the packed matrix's bits) AND, as potf2's LDS image, straight
    // to the factorisation: no wait for the stores, no reload (5 + 1.5 us per diagonal block)
    if (team == 0) gemm_tile_128<false, false, false, false, true, false>(Dn, ld, Cn, ld, Cn, ld, 128, sm, 0, 0, 0, nullptr, tid, sm);
    else { gemm_tile_shadow_barriers(128); __builtin_amdgcn_s_barrier(); }
    __syncthreads();                             // the image is complete for all 8 waves
    PANEL_STAMP(6 + 6 * j);
  }
}

// The role of strip s (128 rows of the panel) for ONE 4-wave team: tid = thread within the team, smem = the team's
// G_SMEM_DOUBLES of LDS.  s >= TPP: an ordinary strip; s = 2, 3: a diagonal strip (blocks (s, 0..s-2), then the early part of
// blocks (s, s-1) and (s, s), then E_s); s = 0, 1 have nothing to do.  Its workgroup barriers and flag waits are workgroup-wide:
// teams sharing a workgroup must run strips of the same kind.
// progress (may be null; factor service): after every finished block (s, j) of the j loop the workgroup adds `teams` to it behind a
// release.  early (may be null; factor service: the progress counters of the diagonal strips 2 and 3, early[0] / early[1]): an
// ordinary strip then applies the k-chunks 0..j-2 of the update of block (s, j) as soon as L(j, 0..j-2) is final -- long before R_j --
// and only the last chunk (columns of block j-1) after R_j: the same products in the same order, continued through memory.
// progress_shift: the count of block (s, j) goes to the 8-bit field j of *progress (teams << progress_shift j; 0: one plain sum).
__device__ __forceinline__ void panel_strip_role(double* smem, double* pan, int64_t ld, double* wp, int* info, PanelSync* sy, int s, int tid,
                                                 int* progress = nullptr, int teams = 1, int* early = nullptr, int progress_shift = 0, bool relaxed = false,
                                                 bool skip_diag = false, bool hand_last = false, int* early_done = nullptr) {
  if (s < 2) return;
  const double* Arow = pan + (int64_t)s * 128;     // my 128 rows of the panel
  const int jlast = s < TPP ? s - 2 : TPP - 1;     // a diagonal strip solves blocks (s, 0..s-2) itself
  for (int j = 0; j <= jlast; ++j) {
    const int64_t cj = (int64_t)j * NBI;
    double* C = pan + (int64_t)s * 128 + cj * ld;
    __syncthreads();                               // the previous tile's LDS reads are over before this one's first DMA lands
    if (j > 0) {
      if (early && j >= 2) {
        const int64_t ke = (int64_t)(j - 1) * NBI;
        if (early_done) {                                   // split chain: a next-diagonal-block role has applied the early chunks (panel_next_diag_role)
          panel_flag_wait(&early_done[(j - 2) * TPP], sy, info, relaxed);
        } else {
          panel_ready_wait(&early[j - 2], j - 1, sy, info, relaxed);   // L(j, 0..j-2) final
          gemm_tile_128<false, false, false, false, false, false>(C, ld, Arow, ld, pan + cj, ld, (int)ke, smem, 0, 0, 0, nullptr, tid);
        }
        panel_flag_wait(&sy->R[j], sy, info, relaxed);      // (its vmcnt(0) + barrier: the block is reloaded as the next call's C)
        gemm_tile_128<false, false, false, false, false, false>(C, ld, Arow + ke * ld, ld, pan + cj + ke * ld, ld, 128, smem, 0, 0, 0, nullptr, tid);
      } else {
        panel_flag_wait(&sy->R[j], sy, info, relaxed);      // rows of strip j left of its diagonal block are final
        gemm_tile_128<false, false, false, false, false, false>(C, ld, Arow, ld, pan + cj, ld, (int)cj, smem, 0, 0, 0, nullptr, tid);
      }
    }
    if (hand_last && j == TPP - 1) {               // split chain, look-ahead strip TPP: the chain helpers solve this block (R[0]: it is ready for them)
      panel_flag_publish(&sy->R[0]);
      break;
    }
    panel_flag_wait(&sy->W[j], sy, info, relaxed);
    gemm_tile_128<true, false, false, false, false, false>(C, ld, C, ld, wp + (int64_t)j * NBI * NBI, 128, 128, smem, 0, 0, 0, nullptr, tid);
    if (progress) panel_count_publish(progress, teams << (progress_shift * j));
  }
  if (s < TPP) {                                   // diagonal strip: the early part of blocks (s, s-1) and (s, s): K = 128 (s-1)
    const int64_t K = (int64_t)(s - 1) * NBI;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my own L(s, 0..s-2): every wave's stores, then the barrier
    __syncthreads();
    // block (s, s-1) first: the factor role's solve tile is the next thing on the chain that needs this strip (E_s = 1); block (s, s),
    // which its update tile preloads one tile later, second (E_s = 2).  (The other order -- (s, s) needs my own rows only and can run
    // before R_{s-1} -- left the factor role waiting 13 us for E_3 in every panel.)
    panel_flag_wait(&sy->R[s - 1], sy, info, relaxed);
    gemm_tile_128<false, false, false, false, false, false>(pan + (int64_t)s * 128 + K * ld, ld, Arow, ld, pan + K, ld, (int)K, smem, 0, 0, 0, nullptr, tid);
    panel_count_publish(&sy->E[s], 1);
    if (skip_diag) return;                         // split chain: the chain helpers apply every update of the diagonal blocks themselves
    gemm_tile_128<false, false, false, false, false, false>(pan + (int64_t)s * 128 + (int64_t)s * NBI * ld, ld, Arow, ld, Arow, ld, (int)K, smem, 0, 0, 0, nullptr, tid);
    panel_count_publish(&sy->E[s], 1);
  }
}

__global__ __launch_bounds__(512) void panel_fused_kernel(double* packed, int64_t n_pad, int p, double* winv, int* info, PanelSync* sy,
                                                          unsigned long long* trace, int potf2_trace) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int sh_id;
  const int t = threadIdx.x;
  if (t == 0) sh_id = atomicAdd(&sy->ticket, 1);
  __syncthreads();
  const int id = sh_id;
  const int64_t ld = panel_ld(n_pad, p);
  double* pan = packed + panel_offset(n_pad, p);
  double* wp = winv + (int64_t)p * TPP * NBI * NBI;
  const int S = (int)(ld / 128);
  const int team = t >> 8, tid = t & 255;
  if (id == 0) {                                   // ---- factor role: the critical chain
    panel_factor_role(sm, pan, ld, wp, info, p, sy, potf2_trace ? nullptr : trace, potf2_trace ? trace : nullptr);
    return;
  }
  int s;
  if (id <= TPP) {                                 // a diagonal strip: one team
    if (team == 1) return;
    s = id - 1;
  } else {
    s = TPP + 2 * (id - TPP - 1) + team;
    if (s >= S) return;                            // odd strip count: the last workgroup runs one team
  }
  panel_strip_role(sm + team * G_SMEM_DOUBLES, pan, ld, wp, info, sy, s, tid);
}

// ------------------------------------------------------------------------------------------------
// The explicit inverse of a panel's NB x NB diagonal block, for the triangular solves with vectors (kernels_vec.hip).
//
// x_p = L_pp^-1 b_p through the four 128-block inverses is a chain of eight dependent small products -- 35 us in one workgroup,
// half of every panel step of a solve.  With inv(L_pp) explicit it is ONE 512 x 512 product, spread over 16 workgroups.
// Stored TRANSPOSED: T = inv(L_pp)^T, column-major with ld = NB, so that row r of the inverse is the NB contiguous doubles at
// T + r NB.  Block row j of T (128-blocks) from the W_i the factorisation leaves in winv:
//   T(j, j) = W_j^T
//   T(j, i) = -( sum_{k = j}^{i-1} T(j, k) L(i, k)^T ) W_i^T        i > j      [inv(i, j) = -W_i sum_k L(i, k) inv(k, j), transposed]
// i.e. per block one accumulating gemm_tile_128 call from a zeroed tile (K = 128 (i - j): the blocks T(j, j..i-1) and L(i, j..i-1)
// are contiguous strips) and one X := X W^T call.  Blocks below the diagonal of T are never written and never read.
// One 4-wave team per block row.  sy != null (factor service): block (j, i) waits for W_i of the panel being factored.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void inv512_row_role(double* smem, const double* pan, int64_t ld, const double* wp, double* T, int j, int tid,
                                                PanelSync* sy, int* info) {
  if (sy) panel_flag_wait(&sy->W[j], sy, info);
  {
    const double* W = wp + (int64_t)j * NBI * NBI;
    double* Tjj = T + (int64_t)j * 128 + (int64_t)j * 128 * NB;
    for (int e = tid; e < 128 * 128; e += 256) Tjj[(e & 127) + (int64_t)(e >> 7) * NB] = W[(e >> 7) + (e & 127) * 128];   // T[c, r] = W[r, c]
  }
  for (int i = j + 1; i < TPP; ++i) {
    double* C = T + (int64_t)j * 128 + (int64_t)i * 128 * NB;
    for (int e = tid; e < 128 * 128; e += 256) C[(e & 127) + (int64_t)(e >> 7) * NB] = 0.0;
    if (sy) panel_flag_wait(&sy->W[i], sy, info);        // (its vmcnt(0) + barrier also settle the stores above for the team's loads)
    else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    gemm_tile_128<false, false, false, false, false, false>(C, NB, T + (int64_t)j * 128 + (int64_t)j * 128 * NB, NB, pan + (int64_t)i * 128 + (int64_t)j * 128 * ld, ld, 128 * (i - j), smem,
                         0, 0, 0, nullptr, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gemm_tile_128<true, false, false, false, false, false>(C, NB, C, NB, wp + (int64_t)i * NBI * NBI, 128, 128, smem, 0, 0, 0, nullptr, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

// panels [p0, p0 + gridDim.x / TPP): one workgroup per block row
__global__ __launch_bounds__(256, 2) void inv512_kernel(const double* packed, int64_t n_pad, const double* winv, double* inv, int p0) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int p = p0 + (int)blockIdx.x / TPP, j = (int)blockIdx.x % TPP;
  inv512_row_role(smem, packed + panel_offset(n_pad, p), panel_ld(n_pad, p), winv + (int64_t)p * TPP * NBI * NBI, inv + (int64_t)p * NB * NB, j,
                  (int)threadIdx.x, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------
// The factor SERVICE: the whole dependent chain of a factorisation in one persistent launch (one-GPU right-looking sweep,
// n <= 24576).
//
// With one fused launch per panel the chain of panel p + 1 (330 us) starts only when the trailing update of panel p has drained,
// and run beside that update on a second stream its workgroups queue for whole CUs behind the update's GEMM tiles.  Here 21
// workgroups are launched ONCE per factorisation on a side stream and stay resident (21 of 256 CUs); they walk through the panels:
//   role 0        the factor role of the fused kernel (diagonal blocks + the two tiles each next one waits for)
//   roles 1, 2    the diagonal strips 2 and 3 (they also count their finished blocks (s, j <= s-2) into E[0] / E[1])
//   roles 3..6    LOOK-AHEAD strips 4..7: the rows of panel p that are the rows of the NEXT diagonal block; after every finished
//                 block (s, j) they count themselves into LA
//   roles 7..16   the ten lower tiles of the next diagonal block: as soon as the look-ahead strips have finished sub-step j
//                 (field j of LA = 4), D(a, b) -= L(4+a, j) L(4+b, j)^T -- the K = 512 update of that tile in its four k-chunks, in
//                 order, continuing one accumulator chain through memory -- and after chunk 3 they count themselves into
//                 ready[p + 1], on which roles 0..2 start panel p + 1.
//   roles 17..20  (when the caller wants it) the explicit inverse of the panel's diagonal block for the vector solves, one block
//                 row each, paced by the W flags (inv512_row_role) -- off the chain
// (One 4-wave team per CU: two teams sharing a CU ran a K = 128 tile in ~39 us instead of ~20, and those tiles are the path
// between two panels' chains.)  So the distance between two chains is one 128-column solve + one K = 128 tile, with no kernel
// launch, no drained GPU and no contended CU on it.  The caller's stream carries only throughput work, one launch per panel
// (trailing_service_kernel): the trailing update of panel p WITHOUT the next diagonal block, and the ordinary strips (>= 8) of panel
// p + 1, which wait on the service's W / R flags.  That kernel is tied in by counters: its tiles of the next panel's rows 4..7 count
// into ready_la[p + 1] (the look-ahead strips of panel p + 1 wait for 16), its tiles of the diagonal block after the next count into
// ready_d2[p + 2] (roles 7..16 wait for 10 before they add panel p + 1's part: k ascending, as in the launch-per-panel form), and its
// tiles of the next panel's column wait for the last field of LA of panel p (their B operand is the look-ahead strips' result).  Same tiles, same
// k order: bit-identical.
// No deadlock: main-stream kernels of panel p wait only on service flags of panel p; the service waits, for panel p, only on the
// update of panel p - 1, which waits only on service flags of panel p - 1; and nothing that waits on the service is launched before
// the service is resident (service_gate_kernel, the first kernel on the caller's stream).
// ------------------------------------------------------------------------------------------------
constexpr int SERVICE_LA0 = 3;                                   // first look-ahead strip role (one 4-wave team per workgroup)
constexpr int SERVICE_D0 = SERVICE_LA0 + TPP;                    // first next-diagonal-block role (one tile per workgroup)
constexpr int SERVICE_INV0 = SERVICE_D0 + PANEL_DIAG_TILES;     // first explicit-inverse role (one block row of inv(L_pp)^T each)
constexpr int SERVICE_H0 = SERVICE_INV0 + TPP;                   // first chain helper (split chain only; 8 waves each)
constexpr int SERVICE_WGS = SERVICE_H0 + CHAIN_HELPERS;

// One team's share of the next diagonal block: lower tile `idx` (0..9: (0,0) (1,0) (1,1) (2,0) ...) of the block, in four k-chunks.
__device__ __forceinline__ void panel_next_diag_role(double* smem, const double* pan, int64_t ld, double* Dn, int64_t ldn, int* info,
                                                     PanelSync* sy, int idx, int tid, unsigned long long* stamp, bool hand_last = false,
                                                     int* aux = nullptr) {
  int tr = 0;
  while ((tr + 1) * (tr + 2) / 2 <= idx) ++tr;
  const int tc = idx - tr * (tr + 1) / 2;
  double* C = Dn + (int64_t)tr * 128 + (int64_t)tc * 128 * ldn;
  // aux (split chain; null otherwise): this role also applies, between its own k-chunks, the EARLY k-chunks of one look-ahead strip's
  // block -- with the chain down to ~200 us a look-ahead strip's ten tiles (~220 us on its one CU) had become the bound
  // (tools/chain_prof.py), while these roles are busy 80 us per panel.  idx 1..4: block (s, 2), s = 3 + idx, chunk 0 after the role's own
  // chunk 0; idx 5..8: block (s, 3), s = idx - 1, chunk 0 there and chunk 1 after the role's chunk 1.  aux[s - TPP] / aux[TPP + s - TPP]
  // = 1: the block has its early chunks (the strip applies the last one).  Same products in the same order, continued through memory.
  const int es = !aux ? -1 : (idx >= 1 && idx <= TPP) ? TPP - 1 + idx : (idx > TPP && idx <= 2 * TPP) ? idx - 1 : -1;   // my look-ahead strip
  const int eb = idx <= TPP ? 2 : 3;                                                                                    // ... and its block
  for (int j = 0; j < TPP; ++j) {
    if (hand_last && j == TPP - 1) {                   // split chain, tile (0, 0): the chain helpers apply the last k-chunk (bit 3 of SU: the tile is ready for them)
      panel_count_publish(&sy->SU, 8);
      break;
    }
    panel_field_wait(&sy->LA, 8 * j, TPP, sy, info);   // all four look-ahead strips have finished sub-step j
    if (stamp && j == TPP - 1 && threadIdx.x == 0) *stamp = __builtin_amdgcn_s_memrealtime();
    gemm_tile_128<false, false, false, false, false, false>(C, ldn, pan + (int64_t)(TPP + tr) * 128 + (int64_t)j * NBI * ld, ld, pan + (int64_t)(TPP + tc) * 128 + (int64_t)j * NBI * ld, ld,
                         128, smem, 0, 0, 0, nullptr, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile is reloaded as the next chunk's C
    __syncthreads();
    if (es >= 0 && j <= eb - 2) {                      // L(es, j) is final (LA field j); L(eb, j): the diagonal strip eb's progress
      panel_ready_wait(&sy->E[eb - 2], j + 1, sy, info);
      double* Ce = const_cast<double*>(pan) + (int64_t)es * 128 + (int64_t)eb * NBI * ld;
      gemm_tile_128<false, false, false, false, false, false>(Ce, ld, pan + (int64_t)es * 128 + (int64_t)j * NBI * ld, ld, pan + (int64_t)eb * NBI + (int64_t)j * NBI * ld, ld,
                           128, smem, 0, 0, 0, nullptr, tid);
      if (j == eb - 2) panel_flag_publish(&aux[(eb - 2) * TPP + es - TPP]);
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
}

// ready: three counters per panel -- ready[p]: lower tiles of diagonal block p complete (10); ready[P + p]: tiles of panel p's rows
// [NB, 2 NB) that have received panel p - 1 (16); ready[2 P + p]: lower tiles of diagonal block p that have received panel p - 2 (10)
// trace (may be null; GPRC_SERVICE_TRACE): 16 s_memrealtime stamps per panel, see gprc_prof_service_trace -- measurement only
#define SERVICE_STAMP(p, k) do { if (trace && (threadIdx.x & 255) == 0) trace[16 * (p) + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)

// The launch serves the panels [p_begin, p_end) -- a group of the grouped left-looking schedule, or all of them: its first panel is
// complete when the launch starts; the next diagonal block is updated only inside the group (the panel behind the group receives
// everything in its left-looking pass, k ascending), the look-ahead strips are solved for every panel that has rows below it.
__global__ __launch_bounds__(512) void panel_service_kernel(double* packed, int64_t n_pad, double* winv, int* info, PanelSync* sy_base,
                                                            int* ready, int P, unsigned long long* trace, double* inv, int p_begin, int p_end,
                                                            int split, int part) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int t = threadIdx.x, team = t >> 8, tid = t & 255;
  if (t == 0) __hip_atomic_fetch_add(&ready[3 * P], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // resident: see service_gate_kernel
  // (Putting the factor role and its four helpers behind ONE L2 -- a grid of 8 x 5 with the five of them on workgroups 0, 8, 16, 24, 32,
  //  the dealing to XCDs being round robin -- was measured and is SLOWER: n = 8192 5.43 -> 5.67 ms, 16384 27.3 -> 27.5; the other twenty
  //  roles then crowd four XCDs and the look-ahead strips fall 70 us behind.  profiles/r03_chain_split.txt.)
  // part 0: the whole service in one launch (role = workgroup).  SHARED service (service_shared): two launches -- part 1 the factor role and
  // the chain helpers (8 waves each: a CU of their own), part 2 the twenty 4-wave roles with only a GEMM team's LDS, so that ONE workgroup of
  // the sweep kernel fits beside each of them and has the CU's matrix cores while the role waits for its flags.
  const int role = part == 0 ? (int)blockIdx.x : part == 1 ? (blockIdx.x == 0 ? 0 : SERVICE_H0 + (int)blockIdx.x - 1) : 1 + (int)blockIdx.x;
  if (part == 2) __builtin_amdgcn_s_setprio(3);   // beside throughput work on the same SIMDs: the role's instructions go first
#ifdef GPRC_CHAIN_PROF
  if (t == 0 && (role == 0 || role >= SERVICE_H0)) g_chain_prof[56 + (role == 0 ? 0 : 1 + role - SERVICE_H0)] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;
#endif
  if (role >= SERVICE_H0) { if (!split) return; }    // the chain helpers: all 8 waves
  else if (role >= 1 && team == 1) return;           // every other role but the factor role is one 4-wave team
  if (role >= SERVICE_INV0 && role < SERVICE_H0 && !inv) return;
  for (int p = p_begin; p < p_end; ++p) {
    PanelSync* sy = sy_base + p;
    const int64_t ld = panel_ld(n_pad, p);
    double* pan = packed + panel_offset(n_pad, p);
    double* wp = winv + (int64_t)p * TPP * NBI * NBI;
    const bool hand = split && p + 1 < p_end;        // the path W_3 -> next chain goes through the chain helpers
    // aux[p][2 TPP]: flags "block (TPP + i, 2) / (TPP + i, 3) of panel p has its early k-chunks", behind the sweep kernel's flags in the sync block
    int* aux = ready + 3 * P + 16 + 2 * P * TPP * P + 8 * P + P * TPP * P * TPP + (int64_t)p * AUX_STRIDE;
    if (role <= 2) {
      if (role == 0) SERVICE_STAMP(p, 14);
      if (p > p_begin) {
        if (role == 0 && split) chain_wait2(&sy_base[p - 1].SU, 16, 7, CHAIN_HELPERS, nullptr, 0, 0, 0, sy, info);   // tile (0, 0) is all the first potf2 reads
        else panel_ready_wait(&ready[p], PANEL_DIAG_TILES, sy, info);
      }
      if (role == 0) SERVICE_STAMP(p, 0);
      if (role == 0) {
        if (split) panel_factor_role_split(sm, pan, ld, wp, info, p, sy);
        else panel_factor_role(sm, pan, ld, wp, info, p, sy, nullptr);
      } else panel_strip_role(sm, pan, ld, wp, info, sy, role + 1, tid, &sy->E[role - 1], 1, nullptr, 0, false, split != 0);
      if (role == 0) SERVICE_STAMP(p, 1);
    } else if (role >= SERVICE_H0) {
      panel_chain_helper_role(sm, pan, ld, wp, info, sy, role - SERVICE_H0, role == SERVICE_H0 && p == GPRC_CHAIN_PROF_PANEL,   // paced by this panel's W flags
                              hand ? packed + panel_offset(n_pad, p + 1) : nullptr, hand ? panel_ld(n_pad, p + 1) : 0, &ready[p + 1],
                              p > p_begin ? &ready[p] : nullptr);
    } else if (role >= SERVICE_INV0) {
      inv512_row_role(sm, pan, ld, wp, inv + (int64_t)p * NB * NB, role - SERVICE_INV0, tid, sy, info);
    } else if (role < SERVICE_D0 ? p + 1 < P : p + 1 < p_end) {
      if (role < SERVICE_D0) {
        if (p > p_begin) panel_ready_wait(&ready[P + p], PANEL_LA_TILES, sy, info);
        if (role == SERVICE_LA0) SERVICE_STAMP(p, 2);
        panel_strip_role(sm, pan, ld, wp, info, sy, TPP + (role - SERVICE_LA0), tid, &sy->LA, 1, sy->E, 8, false, false, hand && role == SERVICE_LA0,
                         hand ? aux + (role - SERVICE_LA0) : nullptr);
        if (role == SERVICE_LA0) SERVICE_STAMP(p, 3);
      } else {
        if (p > p_begin) panel_ready_wait(&ready[2 * P + p + 1], PANEL_DIAG_TILES, sy, info);
        if (role == SERVICE_D0) SERVICE_STAMP(p, 4);
        panel_next_diag_role(sm, pan, ld, packed + panel_offset(n_pad, p + 1), panel_ld(n_pad, p + 1), info, sy, role - SERVICE_D0, tid,
                             (role == SERVICE_D0 && trace) ? trace + 16 * p + 5 : nullptr, hand && role == SERVICE_D0, hand ? aux : nullptr);
        if (!(hand && role == SERVICE_D0)) panel_count_publish(&ready[p + 1], 1);
        if (role == SERVICE_INV0 - 1) SERVICE_STAMP(p, 6);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                               // LDS and this panel's stores are settled before the next panel's first DMA
  }
}

// First kernel of the caller's stream: one wave that returns once every service workgroup is resident.  Whatever waits on the
// service is ordered behind it, so a GPU full of waiting workgroups can never keep the service out.
// limit_ticks: patience in s_memrealtime ticks (WAIT_LIMIT_TICKS; the test hook GPRC_TEST_SERVICE_TIMEOUT passes 0 with an
// unreachable `need` once: the gate gives up at its first look, every wait behind it returns at once, info = GPRC_INFO_WAIT_TIMEOUT).
__global__ void service_gate_kernel(int* alive, int need, int* info, unsigned long long limit_ticks) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(alive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
      if (__builtin_amdgcn_s_memrealtime() - t0 >= limit_ticks) {
        wait_diag(6, need, __hip_atomic_load(alive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), alive, nullptr);
        atomicExch(info, GPRC_INFO_WAIT_TIMEOUT);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
}

// the ordinary strips (s >= 8) of panel p, one 4-wave workgroup each, waiting on the service's flags
__global__ __launch_bounds__(256, 2) void panel_strips_kernel(double* packed, int64_t n_pad, int p, double* winv, int* info, PanelSync* sy,
                                                              unsigned long long* trace) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int64_t ld = panel_ld(n_pad, p);
  if (blockIdx.x == 0) SERVICE_STAMP(p, 7);
  panel_strip_role(smem, packed + panel_offset(n_pad, p), ld, winv + (int64_t)p * TPP * NBI * NBI, info, sy, 2 * TPP + (int)blockIdx.x, (int)threadIdx.x,
                   nullptr, 1, sy->E);
  if (blockIdx.x == 0) SERVICE_STAMP(p, 8);
}

// The caller's-stream kernel of panel p under the factor service: the trailing update of panel p -- every lower tile of the panels
// behind p EXCEPT the next diagonal block (the service's roles 5..9 own it) -- AND the ordinary strips of panel p + 1, which so run
// beside the bulk of the update instead of as a phase of their own (190 us per panel with the GPU two thirds empty).
// Roles are dealt by a ticket counter in START order:
//   tickets [0, n_first)                    the tiles of panel p + 1's own column, rows [NB, 2 NB) first: those 16 count into
//                                           ready_la (the service's look-ahead strips of panel p + 1 wait for them), the others into
//                                           rowcnt[their 128-row strip]; all of them first wait until the look-ahead strips of
//                                           panel p are final (sy->LA >= TPP << 24: their B operand)
//   tickets [n_first, n_first + nstrips)    strip 8 + i of panel p + 1: waits for its four tiles (rowcnt = 4), then the strip role on
//                                           the service's W / R flags of panel p + 1
//   the rest                                all other tiles (XCD-contiguous ranges), the diagonal block of panel p + 2 first (-> ready_d2)
// A strip workgroup can only be running when every ticket before it has started, so the tiles it waits for are running or done,
// and those wait only on the service: no deadlock whatever the dispatch order.
// q_end: the targets are the panels (p, q_end) -- the rest of the group.
__global__ __launch_bounds__(256, 2) void trailing_service_kernel(double* packed, int64_t n_pad, int p, int ntiles, int nstrips, PanelSync* sy_base,
                                                                  int* ready, int* rowcnt, double* winv, int* info, unsigned long long* trace, int q_end) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ int sh_ticket;
  const int P = (int)(n_pad / NB);
  constexpr int DIAG = PANEL_DIAG_TILES;
  PanelSync* sy = sy_base + p;
  const int T0 = TPP * TPP * (P - p - 1) - TPP * (TPP - 1) / 2;   // tiles of panel p + 1 (>= DIAG + LAT: there are >= 2 targets)
  const int n_first = T0 - DIAG;
  // Only the first `base` workgroups (the waiting roles and what they wait for) take tickets; the rest map block id -> tile directly,
  // so that block id mod 8 -- the XCD -- still selects a contiguous range of tiles (ticket order would scatter an XCD's tiles and
  // with them the operand strips its L2 serves: measured -12 % on the update).
  const int base = (n_first + nstrips + 7) & ~7;
  int t;
  if ((int)blockIdx.x < base) {
    if (threadIdx.x == 0) sh_ticket = atomicAdd(&sy->ticket, 1);
    __syncthreads();
    t = sh_ticket;
    if (t >= n_first + nstrips) return;
  } else {
    t = n_first + nstrips + ((int)blockIdx.x - base);
  }
  if (t == 0) SERVICE_STAMP(p, 9);
  int s, local;                                                     // target index behind p + 1, tile index in that panel's list
  bool sig_d2 = false;
  if (t < n_first) { s = 0; local = DIAG + t; }
  else if (t < n_first + nstrips) {                                 // ---- an ordinary strip of panel p + 1
    const int q = p + 1, strip = 2 * TPP + (t - n_first);
    if (strip == 2 * TPP) SERVICE_STAMP(q, 7);
    panel_ready_wait(&rowcnt[(int64_t)q * TPP * P + strip], TPP, sy_base + q, info);
    panel_strip_role(smem, packed + panel_offset(n_pad, q), panel_ld(n_pad, q), winv + (int64_t)q * TPP * NBI * NBI, info, sy_base + q, strip, (int)threadIdx.x,
                     nullptr, 1, sy_base[q].E);
    if (strip == 2 * TPP) SERVICE_STAMP(q, 8);
    return;
  } else {
    const int nrest = ntiles - n_first;
    int id = (int)xcd_remap((unsigned)(t - n_first - nstrips), (unsigned)nrest);
    if (id == nrest - 1) SERVICE_STAMP(p, 13);
    const int T1 = T0 - TPP * TPP;
    if (id < DIAG) { s = 1; local = id; sig_d2 = true; }
    else if (id < T1) { s = 1; local = id; }
    else {
      id -= T1;
      s = 2;
      for (;; ++s) {
        if (p + 1 + s >= q_end) return;
        const int tq = TPP * TPP * (P - p - 1 - s) - TPP * (TPP - 1) / 2;
        if (id < tq) break;
        id -= tq;
      }
      local = id;
    }
  }
  const int q = p + 1 + s;
  int tr, tc;
  if (local < DIAG) {
    tr = 0;
    while ((tr + 1) * (tr + 2) / 2 <= local) ++tr;
    tc = local - tr * (tr + 1) / 2;
  } else {
    tr = TPP + (local - DIAG) / TPP;
    tc = (local - DIAG) % TPP;
  }
  if (s == 0) panel_ready_wait(&sy->LA, TPP << (8 * (TPP - 1)), sy, info);
  if (t == 0) SERVICE_STAMP(p, 10);
  const int64_t ldp = panel_ld(n_pad, p), ldq = panel_ld(n_pad, q);
  const double* Lp = packed + panel_offset(n_pad, p) + (int64_t)(q - p) * NB;  // row q*NB of panel p
  double* Cq = packed + panel_offset(n_pad, q);
  gemm_tile_128<false, false, false, false, false, true, 1>(Cq + (int64_t)tr * 128 + (int64_t)tc * 128 * ldq, ldq, Lp + (int64_t)tr * 128, ldp, Lp + (int64_t)tc * 128, ldp, NB, smem);
  if (s == 0) panel_count_publish(tr < 2 * TPP ? &ready[P + q] : &rowcnt[(int64_t)q * TPP * P + tr], 1);
  else if (sig_d2) panel_count_publish(&ready[2 * P + q], 1);
  if (t == 0) SERVICE_STAMP(p, 11);
  if (sig_d2 && local == 0) SERVICE_STAMP(p, 12);
}


// ------------------------------------------------------------------------------------------------
// The caller's-stream work of a whole GROUP of panels under the factor service in ONE persistent launch (round 3).
//
// One trailing_service_kernel per panel leaves the GPU partly empty at every launch boundary: the last generation of a launch's
// tiles is only partly filled (at n = 8192 the updates have 3.9, 3.4, 2.9 ... generations of ~470 co-resident tiles: 25 generations
// are paid for 20.6), the next launch cannot start one tile before the last one of this launch has finished, and an in-order stream
// has no way to let them overlap (hipExtAnyOrderLaunch is not honoured on gfx950: tools/microbench/anyorder.hip).  Here the same
// work items -- panel p's tiles of panel p + 1's column, the ordinary strips of panel p + 1, every other tile of panel p's update,
// p = p_begin .. p_last - 1 -- are dealt to 2 x (CUs - service workgroups) persistent workgroups by ticket counters, and what the
// kernel boundary used to order is carried by three kinds of flags:
//   ver[q][tr][tc]        number of the group's panels tile (tr, tc) of panel q has received: the update with panel p waits for
//                         p - p_begin and leaves p - p_begin + 1 (the same k order as the launch-per-panel form: identical bits);
//   stripdone[p][strip]   the ordinary strip (>= 2 TPP) of panel p is final -- an operand of panel p's update (the strips of the
//                         group's first panel ran in a launch of their own in front of this kernel);
//   the service's counters and flags as before (LA of panel p for the tiles of panel p + 1's column, rowcnt / ready_la / ready_d2).
// Order of the tickets: panel by panel; inside a panel the `head` (the column of panel p + 1, then that panel's strips: one counter,
// sy[p].ticket) before the `rest`, which is split into eight contiguous ranges with a counter each -- a workgroup takes from the
// range of ITS XCD (HW_REG_XCC_ID) so that an XCD's L2 keeps serving neighbouring tiles' operand strips, and from the other ranges
// only when its own is used up.  A workgroup looks at panel p + 1 only when every ticket of panel p has been taken.
// No deadlock: a workgroup holding a ticket waits only for tickets that come earlier in this order -- all of them taken by
// workgroups that are running -- or for the service, which is resident (service_gate_kernel) and itself waits only for such tickets.
// Every wait is bounded in wall time; after a timeout (info = GPRC_INFO_WAIT_TIMEOUT) every workgroup leaves at its next look.
// ------------------------------------------------------------------------------------------------
#ifdef GPRC_SWEEP_PROF
__device__ unsigned long long g_sweep_prof[8];   // ticks (100 MHz) summed over workgroups: take, wait, gemm, publish, strip; [5] tiles, [6] strips, [7] kernel
#define SWEEP_T(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#define SWEEP_ADD(k, a, b) do { if (threadIdx.x == 0) atomicAdd(&g_sweep_prof[k], (b) - (a)); } while (0)
#else
#define SWEEP_T(var)
#define SWEEP_ADD(k, a, b)
#endif
struct SweepSync {          // views into the sync block behind PanelSync[P], ready[3 P + 16] and rowcnt[P TPP P]
  int* stripdone;           // [P][TPP P]
  int* rest_ticket;         // [P][8]
  int* ver;                 // [P][TPP P][TPP]
  int* aux;                 // [P][AUX_STRIDE]: from 2 TPP on, the finished 32-row slices of the head tiles of the update INTO panel q (sweep_slice_32)
};

// a ticket of counter ctr, or `limit` when it is used up (one lane; one device-scope round trip -- a counter past its limit is harmless)
__device__ __forceinline__ int sweep_take(int* ctr, int limit) {
  const int t = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return t < limit ? t : limit;
}

// waits until *a >= va, *b >= vb, *c >= vc (null pointers are skipped; the three polls fly together), then one acquire for all.
// false: somebody's wait has timed out (info = GPRC_INFO_WAIT_TIMEOUT) -- the caller leaves.
__device__ __forceinline__ bool sweep_wait3(int* a, int va, int* b, int vb, int* c, int vc, int* failed, int* info, int* sh_dead) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) {
    int spins = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
      const int xa = a ? __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : va;
      const int xb = b ? __hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : vb;
      const int xc = c ? __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : vc;
      if (xa >= va && xb >= vb && xc >= vc) break;
      poll_pause(true);
      if ((++spins & 63) != 1) continue;              // at the first miss and every 64th from there
      if (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == GPRC_INFO_WAIT_TIMEOUT) {
        if (xa < va) wait_diag(15, va, xa, a, failed - 1); else if (xb < vb) wait_diag(15, vb, xb, b, failed - 1); else wait_diag(15, vc, xc, c, failed - 1);
        *sh_dead = 1;
        break;
      }
      if (__builtin_amdgcn_s_memrealtime() - t0 > WAIT_LIMIT_TICKS) {   // bounded in wall time (see panel_flag_wait)
        if (xa < va) wait_diag(5, va, xa, a, failed - 1); else if (xb < vb) wait_diag(5, vb, xb, b, failed - 1); else wait_diag(5, vc, xc, c, failed - 1);
        __hip_atomic_store(failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicExch(info, GPRC_INFO_WAIT_TIMEOUT);
        *sh_dead = 1;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  return *sh_dead == 0;
}

// the item is complete: its flag (a version or a strip's "done"), and (may be null) one count for the service / the strips
// written_through: every byte of the item was stored sc1 (gemm_tile_128<.., WT>): nothing of it is dirty in L2, no write-back needed
// (the microarchitecture guide's form "sc1 payload -> every wave's vmcnt(0) -> barrier -> sc1 flag"; the consumers acquire as always)
__device__ __forceinline__ void sweep_publish(int* flag, int value, int* ctr, bool written_through = false) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!written_through) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ctr) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// 32 rows x 128 columns of C -= A B^T over K columns (a multiple of 128) on ONE 4-wave workgroup: chain_slice_32's scheme (the A rows of a
// 128-column chunk through LDS, each wave's B rows straight into registers, write-through stores), wave w the columns [32 w, 32 w + 32).
// Every element: acc = C, then the k-steps ascending with the negate-A bit -- gemm_tile_128's arithmetic, identical bits.
__device__ __forceinline__ void sweep_slice_32(double* C_, int64_t ldc, const double* A_, int64_t lda, const double* B_, int64_t ldb, int K, double* lds_) {
  typedef __attribute__((address_space(1))) double gdouble;
  typedef __attribute__((address_space(3))) double ldouble;
  gdouble* C = (gdouble*)C_;
  const gdouble* A = (const gdouble*)A_;
  const gdouble* B = (const gdouble*)B_;
  ldouble* lds = (ldouble*)lds_;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fk = lane >> 4, fr = lane & 15;
  double4_t acc[2][2];
  gdouble* Cw = C + fr + (int64_t)(32 * wave + fk) * ldc;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][n][r] = Cw[16 * m + (int64_t)(16 * n + 4 * r) * ldc];
  for (int k0 = 0; k0 < K; k0 += 128) {
    const gdouble* asrc = A + 16 * (t & 1) + (int64_t)(k0 + (t >> 1)) * lda;   // thread t: rows [16 (t & 1), +16) of k-slice t >> 1
    double av[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) av[i] = asrc[i];
    const gdouble* bsrc = B + (32 * wave + fr) + (int64_t)(k0 + fk) * ldb;
    double b0[32], b1[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) { b0[s] = bsrc[(int64_t)(4 * s) * ldb]; b1[s] = bsrc[16 + (int64_t)(4 * s) * ldb]; }
    __syncthreads();                                   // the previous chunk's LDS reads are over
    ldouble* adst = lds + (t >> 1) * CHAIN_LDS_LD + 16 * (t & 1);
#pragma unroll
    for (int i = 0; i < 16; ++i) adst[i] = av[i];
    __syncthreads();
    const ldouble* ap = lds + fk * CHAIN_LDS_LD + fr;
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const double a0 = ap[4 * s * CHAIN_LDS_LD], a1 = ap[4 * s * CHAIN_LDS_LD + 16];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0[s], a0, acc[0][0], 0, 0, 1);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0[s], a1, acc[1][0], 0, 0, 1);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1[s], a0, acc[0][1], 0, 0, 1);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1[s], a1, acc[1][1], 0, 0, 1);
    }
  }
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) __hip_atomic_store(&Cw[16 * m + (int64_t)(16 * n + 4 * r) * ldc], acc[m][n][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct SweepItem { int kind, p, s, local, strip, flags; };   // kind 0: tile (s, local) of panel p's update; 1: strip of panel p + 1; 2: rows [32 strip, +32) of head tile `local`; -1: nothing left
constexpr int SWEEP_SIG_D2 = 1, SWEEP_FIRST = 2, SWEEP_LAST = 4;

// One lane's view of the ticket order (see the kernel): which panel it is at, which counter, and that panel's item counts.
struct SweepCursor {
  int p, phase;                                    // phase 0: head; 1 + k: the rest range of XCD (xcc + k) & 7
  int n_first, nstrips, nrest, T1;
  __device__ void enter(int p_, int P, int64_t n_pad, int q_end) {
    p = p_; phase = 0;
    constexpr int DIAG = PANEL_DIAG_TILES;
    const int T0 = TPP * TPP * (P - p - 1) - TPP * (TPP - 1) / 2;   // lower tiles of panel p + 1
    n_first = T0 - DIAG;                                            // ... without its diagonal block (the service's)
    const int ld1 = (int)(panel_ld(n_pad, p + 1) / 128);
    nstrips = ld1 > 2 * TPP ? ld1 - 2 * TPP : 0;
    int ntiles = -DIAG;
    for (int q = p + 1; q < q_end; ++q) ntiles += TPP * TPP * (P - q) - TPP * (TPP - 1) / 2;
    nrest = ntiles - n_first;
    T1 = T0 - TPP * TPP;                                            // lower tiles of panel p + 2
  }
};

// panels p in [p_begin, p_last): the update of panel p over the targets (p, q_end) (without the next diagonal block) and the
// ordinary strips of panel p + 1.  The caller makes sure p_last - 1 still has something to do (p_last + 1 < P, p_last < q_end).
// CORE: the tile loop of gemm_tile_128 -- 1 (first interleaved loop) where two sweep workgroups share a CU, 2 (no VALU instruction in the
// loop) where ONE workgroup has the CU (n_pad < 10752): there the faster tile pays (n = 8192 5.47 -> 5.27 ms); beside a second workgroup
// it costs 5-8 % (12288 13.4 -> 14.3, 16384 26.6 -> 27.5, 32768 171.6 -> 173.7; same box, profiles/r03_chain_split.txt).
template <int CORE>
__global__ __launch_bounds__(256, 2) void trailing_sweep_kernel(double* packed, int64_t n_pad, int p_begin, int p_last, int q_end, PanelSync* sy_base,
                                                                int* ready, int* rowcnt, SweepSync sw, double* winv, int* info,
                                                                unsigned long long* trace, int head_slices) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ SweepItem sh_item;
  __shared__ int sh_dead;
  if (threadIdx.x == 0) sh_dead = 0;
  const int P = (int)(n_pad / NB);
  constexpr int DIAG = PANEL_DIAG_TILES;
  const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7);   // HW_REG_XCC_ID[3:0]: speed only (which L2 this CU sits behind)
  SWEEP_T(tkern0);

  // The next item in ticket order (lane 0; one device-scope atomic in the steady state).  Taking it AHEAD of time -- while the
  // current tile's stores drain -- was measured slower at every size (n = 8192 6.66 against 6.36 ms, 16384 27.75 against 27.52):
  // a ticket then sits behind its holder's publish while somebody may already be waiting for it.
  SweepCursor cur;
  cur.enter(p_begin, P, n_pad, q_end);
  auto find_next = [&](SweepItem& it) {
    it.kind = -1; it.flags = 0; it.s = 0; it.local = 0; it.strip = 0;
    while (cur.p < p_last) {
      it.p = cur.p;
      if (cur.phase == 0) {
        // (CORE == 2, one workgroup per CU: the chain is the bound, and the first 16 head tiles -- the next panel's rows [NB, 2 NB), which
        //  release that panel's look-ahead strips -- are dealt in four 32-row slices each: 64 items of ~25 us instead of 16 tiles of ~100)
        const int extra = (CORE == 2 && head_slices && cur.n_first >= PANEL_LA_TILES) ? 3 * PANEL_LA_TILES : 0;
        const int lim = cur.n_first + cur.nstrips + extra;
        int t = sweep_take(&sy_base[cur.p].ticket, lim);
        if (t >= lim) { cur.phase = 1; continue; }
        if (t == 0) it.flags |= SWEEP_FIRST;
        if (extra && t < 4 * PANEL_LA_TILES) { it.kind = 2; it.local = DIAG + (t >> 2); it.strip = t & 3; return; }
        t -= extra;
        if (t < cur.n_first) { it.kind = 0; it.local = DIAG + t; }
        else { it.kind = 1; it.strip = 2 * TPP + (t - cur.n_first); }
        return;
      }
      if (cur.phase <= 8) {
        const int x = (xcc + cur.phase - 1) & 7;
        const int rq = cur.nrest >> 3, rr = cur.nrest & 7;
        const int cnt = rq + (x < rr ? 1 : 0);
        const int start = x < rr ? x * (rq + 1) : rr * (rq + 1) + (x - rr) * rq;
        const int i = cnt > 0 ? sweep_take(sw.rest_ticket + (int64_t)cur.p * 8 + x, cnt) : cnt;
        if (i >= cnt) { ++cur.phase; continue; }
        int id = start + i;
        if (id == cur.nrest - 1) it.flags |= SWEEP_LAST;
        if (id < cur.T1) { it.kind = 0; it.s = 1; it.local = id; if (id < DIAG) it.flags |= SWEEP_SIG_D2; return; }
        id -= cur.T1;
        int s = 2;
        for (; cur.p + 1 + s < q_end; ++s) {
          const int tq = TPP * TPP * (P - cur.p - 1 - s) - TPP * (TPP - 1) / 2;
          if (id < tq) break;
          id -= tq;
        }
        if (cur.p + 1 + s < q_end) { it.kind = 0; it.s = s; it.local = id; return; }
        continue;
      }
      if (cur.p + 1 < p_last) cur.enter(cur.p + 1, P, n_pad, q_end);   // every ticket of this panel has been taken
      else cur.p = p_last;
    }
  };

  SweepItem nxt;
  for (;;) {
    SWEEP_T(tk0);
    if (threadIdx.x == 0) find_next(nxt);
    __syncthreads();                                  // every wave has left the previous item (its LDS, its view of sh_item)
    if (threadIdx.x == 0) sh_item = nxt;
    __syncthreads();
    const SweepItem it = sh_item;
    SWEEP_T(tk1);
    SWEEP_ADD(0, tk0, tk1);
    if (it.kind < 0) break;
    const int p = it.p;
    PanelSync* sy = sy_base + p;
    if (it.flags & SWEEP_FIRST) SERVICE_STAMP(p, 9);
    if (it.flags & SWEEP_LAST) SERVICE_STAMP(p, 13);
    if (it.kind == 0 || it.kind == 2) {
      // ---- one tile of the update with panel p: target q = p + 1 + s, tile (tr, tc) of that panel (kind 2: 32 rows of it)
      const int q = p + 1 + it.s, local = it.local;
      int tr, tc;
      if (local < DIAG) {
        tr = 0;
        while ((tr + 1) * (tr + 2) / 2 <= local) ++tr;
        tc = local - tr * (tr + 1) / 2;
      } else {
        tr = TPP + (local - DIAG) / TPP;
        tc = (local - DIAG) % TPP;
      }
      const int stage = p - p_begin;                                // versions the tiles of this panel's update wait for
      const int ra = (q - p) * TPP + tr, rb = (q - p) * TPP + tc;   // the operands' 128-row strips of panel p
      int* verp = sw.ver + ((int64_t)q * TPP * P + tr) * TPP + tc;
      // strips TPP .. 2 TPP - 1 are the service's look-ahead strips (LA >= TPP << 24 when all four are final); the others ride in this
      // kernel, except those of the group's first panel, which ran in a launch of their own in front of it
      int* sd_p = sw.stripdone + (int64_t)p * TPP * P;
      int* wa = ra >= 2 * TPP ? (stage > 0 ? sd_p + ra : nullptr) : &sy->LA;
      int* wb = rb >= 2 * TPP ? (stage > 0 ? sd_p + rb : nullptr) : &sy->LA;
      if (wb == wa) wb = nullptr;
      SWEEP_T(tw0);
      constexpr int LA_FINAL = TPP << (8 * (TPP - 1));
      if (!sweep_wait3(stage > 0 ? verp : nullptr, stage, wa, wa == &sy->LA ? LA_FINAL : 1, wb, wb == &sy->LA ? LA_FINAL : 1, &sy->failed, info, &sh_dead)) break;
      SWEEP_T(tw1);
      const int64_t ldp = panel_ld(n_pad, p), ldq = panel_ld(n_pad, q);
      const double* Lp = packed + panel_offset(n_pad, p) + (int64_t)(q - p) * NB;   // row q NB of panel p
      double* Cq = packed + panel_offset(n_pad, q);
      // (CORE = 1: see the loops' comments.  WT: the tile is stored write-through, so that its publication needs no write-back of the XCD's
      //  L2 -- a release fence per tile, by ~60 workgroups per XCD, each flushing what all of them have dirtied since the last one, was
      //  2 % of the mid-size factorisation: n = 16384 27.42 -> 26.84 ms, 8192 5.98 -> 5.81, same box)
      if constexpr (CORE == 2) if (it.kind == 2) {
        sweep_slice_32(Cq + (int64_t)tr * 128 + (int64_t)tc * 128 * ldq + 32 * it.strip, ldq, Lp + (int64_t)tr * 128 + 32 * it.strip, ldp, Lp + (int64_t)tc * 128, ldp, NB, smem);
        // the slice is stored (write-through): count it; the fourth one publishes the tile as a whole tile's holder would
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0 &&
            (__hip_atomic_fetch_add(sw.aux + (int64_t)q * AUX_STRIDE + 2 * TPP + (local - DIAG), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 3) == 3) {
          __hip_atomic_store(verp, stage + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(&ready[P + q], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        SWEEP_ADD(5, 0ull, 1ull);
        if (it.flags & SWEEP_FIRST) SERVICE_STAMP(p, 11);
        continue;
      }
      gemm_tile_128<false, false, false, false, false, true, CORE, true>(Cq + (int64_t)tr * 128 + (int64_t)tc * 128 * ldq, ldq, Lp + (int64_t)tr * 128, ldp, Lp + (int64_t)tc * 128, ldp, NB, smem);
      SWEEP_T(tw2);
      int* ctr = nullptr;
      if (it.s == 0) ctr = tr < 2 * TPP ? &ready[P + q] : &rowcnt[(int64_t)q * TPP * P + tr];
      else if (it.flags & SWEEP_SIG_D2) ctr = &ready[2 * P + q];
      sweep_publish(verp, stage + 1, ctr, true);
      SWEEP_T(tw3);
      SWEEP_ADD(1, tw0, tw1); SWEEP_ADD(2, tw1, tw2); SWEEP_ADD(3, tw2, tw3); SWEEP_ADD(5, 0ull, 1ull);
      if (it.flags & SWEEP_FIRST) SERVICE_STAMP(p, 11);
    } else {
      // ---- an ordinary strip of panel p + 1: its four tiles of this update first, then the strip role on the service's flags
      const int q = p + 1, strip = it.strip;
      if (strip == 2 * TPP) SERVICE_STAMP(q, 7);
      if (!sweep_wait3(&rowcnt[(int64_t)q * TPP * P + strip], TPP, nullptr, 0, nullptr, 0, &sy_base[q].failed, info, &sh_dead)) break;
      panel_strip_role(smem, packed + panel_offset(n_pad, q), panel_ld(n_pad, q), winv + (int64_t)q * TPP * NBI * NBI, info, sy_base + q, strip, (int)threadIdx.x,
                       nullptr, 1, sy_base[q].E, 0, true);
      sweep_publish(sw.stripdone + (int64_t)q * TPP * P + strip, 1, nullptr);
      SWEEP_T(ts1);
      SWEEP_ADD(4, tk1, ts1); SWEEP_ADD(6, 0ull, 1ull);
      if (strip == 2 * TPP) SERVICE_STAMP(q, 8);
    }
  }
  SWEEP_T(tkern1);
  SWEEP_ADD(7, tkern0, tkern1);
}
}  // namespace
}  // namespace gprc
// include/gprc_native.h: the raw records of the timed-out waits, see wait_diag; clears them
extern "C" __attribute__((visibility("default"))) int gprc_prof_wait_timeout(int* out, int ints) {   // out[0]: records written; 8 ints per record from out[8]
  if (!out || ints < 8) return -1;
  const size_t bytes = sizeof(int) * (size_t)std::min(ints, 8 * (gprc::WAIT_DIAG_RECORDS + 1));
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gprc::g_wait_diag), bytes) != hipSuccess) return -1;
  static const int z[8 * (gprc::WAIT_DIAG_RECORDS + 1)] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(gprc::g_wait_diag), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
namespace gprc {
namespace {
#ifdef GPRC_CHAIN_PROF
}  // namespace
}  // namespace gprc
extern "C" __attribute__((visibility("default"))) int gprc_debug_chain_prof(unsigned long long* out64) {
  return hipMemcpyFromSymbol(out64, HIP_SYMBOL(gprc::g_chain_prof), 512) == hipSuccess ? 0 : -1;
}
namespace gprc {
namespace {
#endif
#ifdef GPRC_SWEEP_PROF
}  // namespace
}  // namespace gprc
extern "C" __attribute__((visibility("default"))) int gprc_debug_sweep_prof(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(gprc::g_sweep_prof), 64) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(gprc::g_sweep_prof), z, 64) != hipSuccess) return -1; }
  return 0;
}
namespace gprc {
namespace {
#endif

}  // namespace

int launch_potf2_inv(hipStream_t s, double* A, int64_t lda, double* winv, int* info_dev, int col0) {
  ProfScope ps(s, PK_POTF2, 128.0 * 128 * 128 / 3 * 2, 8.0 * 3 * 128 * 128);
  static bool attr_set[MAX_DEVICES] = {};  // the attribute is per device: one process may hold contexts on several
  const size_t smem = PB_SMEM_DOUBLES * sizeof(double);
  int dev = 0;
  GPRC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= MAX_DEVICES || !attr_set[dev]) {
    GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(potf2_inv_blocked_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    if (dev >= 0 && dev < MAX_DEVICES) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(potf2_inv_blocked_kernel, dim3(1), dim3(1024), smem, s, A, lda, winv, info_dev, col0);
  GPRC_LAUNCH_CHECK();
  return 0;
}

int launch_panel_fused(hipStream_t s, double* packed, int64_t n_pad, int64_t p, double* winv, int* info_dev, void* sync16) {
  static bool attr_set[MAX_DEVICES] = {};
  const size_t smem = PB_SMEM_DOUBLES * sizeof(double);
  static_assert(PB_SMEM_DOUBLES >= 2 * G_SMEM_DOUBLES, "two GEMM teams must fit beside each other in the factor role's LDS");
  int dev = 0;
  GPRC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= MAX_DEVICES || !attr_set[dev]) {
    GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    if (dev >= 0 && dev < MAX_DEVICES) attr_set[dev] = true;
  }
  const int64_t ld = panel_ld(n_pad, p), S = ld / 128;
  const unsigned grid = (unsigned)(1 + TPP + (S - TPP + 1) / 2);
  GPRC_HIP(hipMemsetAsync(sync16, 0, sizeof(PanelSync), s));
  // algorithmic work of a whole panel factorisation: in-panel updates + panel solves + the four diagonal blocks
  double fl = 0.0;
  for (int j = 0; j < TPP; ++j) fl += 2.0 * (double)(ld - j * NBI) * NBI * (j * NBI) + (double)(ld - (j + 1) * NBI) * NBI * NBI + 2.0 * NBI * NBI * NBI / 3.0;
  ProfScope ps(s, PK_PANEL_FUSED, fl, 8.0 * 2.0 * (double)ld * NB);
  // GPRC_PANEL_TRACE=<p>: the factor role of panel p stamps its stages (24 x 8 bytes at sync + 64 .. sync + 256; read back with
  // gprc_prof_panel_trace).  Measurement only.
  static const long long trace_p = [] { const char* e = std::getenv("GPRC_PANEL_TRACE"); return e ? std::atoll(e) : -1LL; }();
  unsigned long long* trace = (trace_p == p) ? reinterpret_cast<unsigned long long*>(static_cast<char*>(sync16) + 64) : nullptr;
  static const int potf2_trace = std::getenv("GPRC_POTF2_TRACE") != nullptr;   // the stamps are those of the SECOND diagonal block's potf2 instead
  hipLaunchKernelGGL(panel_fused_kernel, dim3(grid), dim3(512), smem, s, packed, n_pad, (int)p, winv, info_dev, reinterpret_cast<PanelSync*>(sync16), trace,
                     potf2_trace);
  GPRC_LAUNCH_CHECK();
  return 0;
}

static int ensure_gemm_attrs();

// flags of every panel | ready, ready_la, ready_d2 (P ints each) + the service's "resident" counter | rowcnt (P x 4 P ints: per panel,
// per 128-row strip, the tiles of that strip which have received the previous panel)
// ... and behind them the persistent sweep's flags (trailing_sweep_kernel): stripdone[P][TPP P], rest_ticket[P][8], ver[P][TPP P][TPP]
size_t panel_service_sync_bytes(int64_t P) {
  return (size_t)P * sizeof(PanelSync) + (3 * (size_t)P + 16) * sizeof(int) + (size_t)P * TPP * P * sizeof(int) +
         ((size_t)P * TPP * P + 8 * (size_t)P + (size_t)P * TPP * P * TPP) * sizeof(int) + (size_t)P * AUX_STRIDE * sizeof(int);   // ... and aux[P][AUX_STRIDE]
}

// sync: panel_service_sync_bytes(P) bytes of device memory, zeroed by the caller (stream-ordered before this launch)
int launch_inv512(hipStream_t s, const double* packed, int64_t n_pad, const double* winv, double* inv, int64_t p_begin, int64_t p_end) {
  if (p_end <= p_begin) return 0;
  GPRC_TRY(ensure_gemm_attrs());
  hipLaunchKernelGGL(inv512_kernel, dim3((unsigned)((p_end - p_begin) * TPP)), dim3(256), G_SMEM_DOUBLES * sizeof(double), s, packed, n_pad, winv, inv,
                     (int)p_begin);
  GPRC_LAUNCH_CHECK();
  return 0;
}

// The split chain (four more resident CUs) where the panel chain weighs: below n_pad = 20480, i.e. wherever the whole matrix is ONE group
// of the service (measured, same box, profiles/r03_chain_split.txt; in the grouped schedule beyond it makes no difference and the four
// CUs stay with the update).  GPRC_CHAIN_SPLIT=0 / 1 forces it off / on at every size.
bool chain_split(int64_t n_pad) {
  static const int v = [] { const char* e = std::getenv("GPRC_CHAIN_SPLIT"); return e ? std::atoi(e) : -1; }();
  return v >= 0 ? v != 0 : n_pad < 20480;
}

// the records the timed-out waits left (wait_diag), as text for the host's error message; clears them
std::string wait_timeout_report() {
  static int v[8 * (WAIT_DIAG_RECORDS + 1)];
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_wait_diag), sizeof(v)) != hipSuccess || v[0] == 0) return "";
  static const int z[8 * (WAIT_DIAG_RECORDS + 1)] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wait_diag), z, sizeof(z));
  static const char* const site[] = {"?", "flag", "count", "field", "chain", "sweep", "gate"};
  std::string out = " [waits that gave up (+10: still waiting when somebody else had):";
  const int n = v[0] < WAIT_DIAG_RECORDS ? v[0] : WAIT_DIAG_RECORDS;
  for (int k = 0; k < n && k < 8; ++k) {
    const int* r = v + 8 * (k + 1);
    const int st = r[0] % 10;
    out += std::string(k ? ";" : "") + " " + site[st >= 0 && st <= 6 ? st : 0] + (r[0] >= 10 ? "+10" : "") + " wait of workgroup " + std::to_string(r[1]) + "/" +
           std::to_string(r[2]) + "x" + std::to_string(r[6]) + " needed " + std::to_string(r[3]) + " saw " + std::to_string(r[4]) + " word " + std::to_string(r[5]);
  }
  return out + (v[0] > 8 ? "; ... " + std::to_string(v[0]) + " in all]" : "]");
}

// The SHARED service.  A service workgroup asks for the factor role's 149 KB of LDS, so nothing else fits on its CU -- 21 to 25 CUs
// that, where the update is the bound, mostly sleep on their flags.  From n_pad = 13312 on, the 4-wave roles (all but the factor role and
// the chain helpers, whose 8 waves x 256 VGPRs fill a CU) are launched on their own, on a second side stream, with a GEMM team's LDS only
// (launch_panel_service, part 2): ONE sweep workgroup then fits beside each of them and has the CU's matrix cores while the role waits.
// Measured, same box (profiles/r03_chain_split.txt): n = 14336 19.45 -> 19.07 ms, 16384 27.3 -> 26.4, 18432 37.55 -> 35.97, 24576
// 79.7 -> 78.9, 32768 172.5 -> 170.4, 65536 unchanged; 12288 and below unchanged or slower (the chain's own tiles run at half rate
// beside a busy sweep workgroup), hence the threshold.  Raising the roles' wave priority (s_setprio 3) changes nothing measurable; it
// stays.  GPRC_SERVICE_SHARE=0 / 1 forces it off / on (from n_pad = 10752, where the sweep runs two workgroups per CU).
bool service_shared(int64_t n_pad) {
  static const int v = [] { const char* e = std::getenv("GPRC_SERVICE_SHARE"); return e ? std::atoi(e) : -1; }();
  return n_pad >= 10752 && (v >= 0 ? v != 0 : n_pad >= 13312);
}

// workgroups the service keeps resident (a CU each)
int service_workgroups(bool with_inverse, int64_t n_pad) { return SERVICE_INV0 + (with_inverse ? TPP : 0) + (chain_split(n_pad) ? CHAIN_HELPERS : 0); }

// part: 0 the whole service; 1 / 2 the two launches of the shared service (on two streams: they run side by side)
int launch_panel_service(hipStream_t s, double* packed, int64_t n_pad, double* winv, int* info_dev, void* sync, void* trace, double* inv,
                         int64_t p_begin, int64_t p_end, int part) {
  static bool attr_set[MAX_DEVICES] = {};
  const size_t smem = PB_SMEM_DOUBLES * sizeof(double);
  int dev = 0;
  GPRC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= MAX_DEVICES || !attr_set[dev]) {
    GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_service_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    if (dev >= 0 && dev < MAX_DEVICES) attr_set[dev] = true;
  }
  const int64_t P = n_pad / NB;
  PanelSync* sy = reinterpret_cast<PanelSync*>(sync);
  int* ready = reinterpret_cast<int*>(static_cast<char*>(sync) + (size_t)P * sizeof(PanelSync));
  ProfScope ps(s, PK_PANEL_FUSED, 0.0, 0.0);
  const unsigned grid = part == 0 ? SERVICE_WGS : part == 1 ? 1 + CHAIN_HELPERS : SERVICE_H0 - 1;
  hipLaunchKernelGGL(panel_service_kernel, dim3(grid), dim3(512), part == 2 ? G_SMEM_DOUBLES * sizeof(double) : smem, s, packed, n_pad, winv, info_dev, sy,
                     ready, (int)P, static_cast<unsigned long long*>(trace), inv, (int)p_begin, (int)p_end, chain_split(n_pad) ? 1 : 0, part);
  GPRC_LAUNCH_CHECK();
  return 0;
}

// launches: service launches so far on this sync buffer, this one included (the "resident" counter is cumulative)
int launch_service_gate(hipStream_t s, int64_t n_pad, int* info_dev, void* sync, int launches) {
  const int64_t P = n_pad / NB;
  int* ready = reinterpret_cast<int*>(static_cast<char*>(sync) + (size_t)P * sizeof(PanelSync));
  // GPRC_TEST_SERVICE_TIMEOUT=1 (test hook): the FIRST gate of the process waits for a residency count that cannot be reached and
  // gives up at once -- the timeout / refill / service-off path of the fit entry points without a profiler.
  static std::atomic<bool> fire{std::getenv("GPRC_TEST_SERVICE_TIMEOUT") != nullptr};
  const bool forced = fire.exchange(false);
  hipLaunchKernelGGL(service_gate_kernel, dim3(1), dim3(64), 0, s, ready + 3 * P, forced ? (1 << 30) : SERVICE_WGS * launches, info_dev,
                     forced ? 0ULL : WAIT_LIMIT_TICKS);
  GPRC_LAUNCH_CHECK();
  return 0;
}

// the ordinary strips of panel p (rows from 2 NB below the panel's top)
int launch_panel_strips(hipStream_t s, double* packed, int64_t n_pad, int64_t p, double* winv, int* info_dev, void* sync, void* trace) {
  const int64_t ld = panel_ld(n_pad, p), S = ld / 128;
  if (S <= 2 * TPP) return 0;
  GPRC_TRY(ensure_gemm_attrs());
  double fl = 0.0;
  for (int j = 0; j < TPP; ++j) fl += 2.0 * (double)(ld - 2 * NB) * NBI * (j * NBI) + (double)(ld - 2 * NB) * NBI * NBI;
  ProfScope ps(s, PK_GEMM_INNER, fl, 8.0 * 2.0 * (double)(ld - 2 * NB) * NB);
  hipLaunchKernelGGL(panel_strips_kernel, dim3((unsigned)(S - 2 * TPP)), dim3(256), G_SMEM_DOUBLES * sizeof(double), s, packed, n_pad, (int)p, winv,
                     info_dev, reinterpret_cast<PanelSync*>(sync) + p, static_cast<unsigned long long*>(trace));
  GPRC_LAUNCH_CHECK();
  return 0;
}

// the caller's-stream kernel of panel p under the service: the trailing update of panel p over the targets (p, q_end) (everything
// except the next diagonal block) + the ordinary strips of panel p + 1
int launch_trailing_service(hipStream_t s, double* packed, int64_t n_pad, int64_t p, double* winv, int* info_dev, void* sync, void* trace, int64_t q_end) {
  const int64_t P = n_pad / NB;
  if (q_end > P) q_end = P;
  if (q_end - p - 1 < 1 || p + 2 >= P) return 0;   // no target, or the only target is the last panel: its diagonal block is all there is
  GPRC_TRY(ensure_gemm_attrs());
  int64_t tiles = -(int64_t)PANEL_DIAG_TILES;
  double fl = 0.0, by = 0.0;
  for (int64_t q = p + 1; q < q_end; ++q) {
    tiles += (int64_t)TPP * TPP * (P - q) - TPP * (TPP - 1) / 2;
    const double rows = (double)(n_pad - q * NB);
    const double elems = rows * NB - 0.5 * NB * (double)(NB - 1) - (q == p + 1 ? 0.5 * NB * (double)(NB + 1) : 0.0);
    fl += 2.0 * elems * NB;
    by += 8.0 * (2.0 * elems + rows * NB);
  }
  const int64_t ld1 = panel_ld(n_pad, p + 1);
  const int64_t nstrips = std::max<int64_t>(0, ld1 / 128 - 2 * TPP);
  for (int j = 0; j < TPP; ++j) fl += nstrips * (2.0 * 128 * NBI * (j * NBI) + 128.0 * NBI * NBI);
  ProfScope ps(s, PK_TRAILING, fl, by);
  PanelSync* sy = reinterpret_cast<PanelSync*>(sync);
  int* ready = reinterpret_cast<int*>(static_cast<char*>(sync) + (size_t)P * sizeof(PanelSync));
  int* rowcnt = ready + 3 * P + 16;
  const int64_t n_first = (int64_t)TPP * TPP * (P - p - 1) - TPP * (TPP - 1) / 2 - PANEL_DIAG_TILES;
  const int64_t base = (n_first + nstrips + 7) & ~(int64_t)7;     // ticketed workgroups, padded to a multiple of the XCD count
  hipLaunchKernelGGL(trailing_service_kernel, dim3((unsigned)(base + (tiles - n_first))), dim3(256), G_SMEM_DOUBLES * sizeof(double), s, packed, n_pad, (int)p,
                     (int)tiles, (int)nstrips, sy, ready, rowcnt, winv, info_dev, static_cast<unsigned long long*>(trace), (int)q_end);
  GPRC_LAUNCH_CHECK();
  return 0;
}

// the caller's-stream work of the panels [g0, g1) of a group under the service in ONE persistent launch (trailing_sweep_kernel);
// service_wgs: workgroups the service keeps resident (they hold a CU each)
int launch_trailing_sweep(hipStream_t s, double* packed, int64_t n_pad, int64_t g0, int64_t g1, double* winv, int* info_dev, void* sync, void* trace,
                          int service_wgs) {
  const int64_t P = n_pad / NB;
  const int64_t q_end = std::min(g1, P);
  int64_t p_last = g0;                                              // one past the last panel with something to do (launch_trailing_service's rule)
  for (int64_t p = g0; p + 1 < g1; ++p)
    if (!(q_end - p - 1 < 1 || p + 2 >= P)) p_last = p + 1;
  if (p_last == g0) return 0;
  GPRC_TRY(ensure_gemm_attrs());
  double fl = 0.0, by = 0.0;
  for (int64_t p = g0; p < p_last; ++p) {
    for (int64_t q = p + 1; q < q_end; ++q) {
      const double rows = (double)(n_pad - q * NB);
      const double elems = rows * NB - 0.5 * NB * (double)(NB - 1) - (q == p + 1 ? 0.5 * NB * (double)(NB + 1) : 0.0);
      fl += 2.0 * elems * NB;
      by += 8.0 * (2.0 * elems + rows * NB);
    }
    const int64_t nstrips = std::max<int64_t>(0, panel_ld(n_pad, p + 1) / 128 - 2 * TPP);
    for (int j = 0; j < TPP; ++j) fl += nstrips * (2.0 * 128 * NBI * (j * NBI) + 128.0 * NBI * NBI);
  }
  ProfScope ps(s, PK_TRAILING, fl, by);
  PanelSync* sy = reinterpret_cast<PanelSync*>(sync);
  int* ready = reinterpret_cast<int*>(static_cast<char*>(sync) + (size_t)P * sizeof(PanelSync));
  int* rowcnt = ready + 3 * P + 16;
  SweepSync sw;
  sw.stripdone = rowcnt + P * TPP * P;
  sw.rest_ticket = sw.stripdone + P * TPP * P;
  sw.ver = sw.rest_ticket + 8 * P;
  sw.aux = sw.ver + P * TPP * P * TPP;
  int dev = 0, cus = 0;
  GPRC_HIP(hipGetDevice(&dev));
  GPRC_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  // Two workgroups per CU the service leaves free (all co-resident) -- except below n_pad = 10752, where the panel chain is the bound and
  // what counts is how quickly a tile the chain waits for is done: ONE workgroup per CU has the CU's MFMA pipes to itself
  // (measured, same box: n = 8192 6.28 -> 5.99 ms, 10240 9.44 -> 9.21; 12288 13.55 -> 14.60: profiles/r03_factor_schedules.txt).
  // GPRC_SWEEP_WGS=<n> overrides.
  static const int wgs_env = [] { const char* e = std::getenv("GPRC_SWEEP_WGS"); return e ? std::atoi(e) : 0; }();
  const int per_cu = n_pad < 10752 ? 1 : 2;
  // shared service: the resident 4-wave roles' CUs take ONE sweep workgroup each beside the role
  const int shared = service_shared(n_pad) ? service_wgs - 1 - (chain_split(n_pad) ? CHAIN_HELPERS : 0) : 0;
  const int wgs = wgs_env > 0 ? wgs_env : std::max(8, per_cu * (cus - service_wgs) + shared);
  static const int head_slices = [] { const char* e = std::getenv("GPRC_HEAD_SLICES"); return e ? std::atoi(e) : 1; }();   // 0: whole head tiles (A/B switch)
  if (per_cu == 1 && wgs_env <= 0)
    hipLaunchKernelGGL(trailing_sweep_kernel<2>, dim3((unsigned)wgs), dim3(256), G_SMEM_DOUBLES * sizeof(double), s, packed, n_pad, (int)g0, (int)p_last,
                       (int)q_end, sy, ready, rowcnt, sw, winv, info_dev, static_cast<unsigned long long*>(trace), head_slices);
  else
    hipLaunchKernelGGL(trailing_sweep_kernel<1>, dim3((unsigned)wgs), dim3(256), G_SMEM_DOUBLES * sizeof(double), s, packed, n_pad, (int)g0, (int)p_last,
                       (int)q_end, sy, ready, rowcnt, sw, winv, info_dev, static_cast<unsigned long long*>(trace), 0);
  GPRC_LAUNCH_CHECK();
  return 0;
}

static int ensure_gemm_attrs() {
  static bool done[MAX_DEVICES] = {};  // per device, as above
  int dev = 0;
  GPRC_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < MAX_DEVICES && done[dev]) return 0;
  const int smem = (int)(G_SMEM_DOUBLES * sizeof(double));
  GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<PK_GEMM_INNER>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<PK_SOLVE_UPDATE>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<PK_COV_SYRK>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trsm_panel_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trsm_panel_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(solve_panel_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(solve_panel_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trailing_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_strips_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trailing_service_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trailing_sweep_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trailing_sweep_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(inv512_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(solve_left_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trailing_range_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  if (dev >= 0 && dev < MAX_DEVICES) done[dev] = true;
  return 0;
}

int launch_solve_left(hipStream_t s, double* vt, int64_t ldv, int64_t m_pad, const double* packed, int64_t n_pad, int64_t j, int64_t G,
                      int64_t tri_row0) {
  if (j <= 0 || m_pad <= 0 || G <= 0) return 0;
  if (m_pad % 128) { set_error("solve_left: m_pad must be a multiple of 128"); return GPRC_ERR_ARG; }
  if ((j + G) * NB > n_pad) { set_error("solve_left: panel group beyond the factor"); return GPRC_ERR_ARG; }
  GPRC_TRY(ensure_gemm_attrs());
  const int64_t N = G * NB, tiles = (m_pad / 128) * (N / 128);
  // GPRC_KCHUNK=<panels>: the pass is cut into launches of at most that many source panels (K = 512 x panels): the tiles an
  // XCD runs concurrently are re-aligned at every launch boundary, so its L2 keeps serving the shared operand strips;
  // the price is one more C tile load/store per chunk.  Same products, same order: bit-identical.  Default: one launch.
  static const int64_t kchunk = [] { const char* e = std::getenv("GPRC_KCHUNK"); return e ? std::atoll(e) : 0; }();
  const int64_t step = kchunk > 0 ? kchunk : j;
  // GPRC_TILE256=1: 256 x 128 macro-tiles on 8 waves (experiment: -25 % operand staging per flop); needs m_pad % 256 == 0
  static const bool tile256 = [] { const char* e = std::getenv("GPRC_TILE256"); return e && std::atoi(e) == 1; }();
  const bool use256 = tile256 && m_pad % 256 == 0;
  if (use256) {
    static bool attr_set[MAX_DEVICES] = {};
    int dev = 0;
    GPRC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES || !attr_set[dev]) {
      GPRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(solve_left_kernel256), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(T2_SMEM_DOUBLES * sizeof(double))));
      if (dev >= 0 && dev < MAX_DEVICES) attr_set[dev] = true;
    }
  }
  for (int64_t kp0 = 0; kp0 < j; kp0 += step) {
    const int64_t kp1 = std::min(j, kp0 + step), K = (kp1 - kp0) * NB;
    double fl = 2.0 * (double)m_pad * N * (double)K;
    if (tri_row0 >= 0) {   // algorithmic work of the triangular form: per 128-row tile only the columns from its first row on
      fl = 0.0;
      for (int64_t tr = 0; tr < m_pad / 128; ++tr) fl += 2.0 * 128.0 * N * (double)std::max<int64_t>(0, kp1 * NB - std::max<int64_t>(kp0 * NB, tri_row0 + tr * 128));
    }
    ProfScope ps(s, PK_SOLVE_LEFT, fl, 8.0 * (2.0 * m_pad * N + (double)m_pad * K + (double)N * K));
    if (use256 && tri_row0 < 0)
      hipLaunchKernelGGL(solve_left_kernel256, dim3((unsigned)(tiles / 2)), dim3(512), T2_SMEM_DOUBLES * sizeof(double), s, vt, ldv, packed, n_pad,
                         (int)j, (int)(m_pad / 256), (int)(N / 128), 8, (int)kp0, (int)kp1);
    else
      hipLaunchKernelGGL(solve_left_kernel, dim3((unsigned)tiles), dim3(256), G_SMEM_DOUBLES * sizeof(double), s, vt, ldv, packed, n_pad,
                         (int)j, (int)(m_pad / 128), (int)(N / 128), 8, (int)kp0, (int)kp1, tri_row0);
    GPRC_LAUNCH_CHECK();
  }
  return 0;
}

// ssq != nullptr: also ssq[i] = sum_j X_new[i][j]^2 for the M rows (the predict's fused colSums(v * v) partial of this block column)
// vt[:, panel p] := vt[:, panel p] L_pp^-T (in place; everything left of panel p already applied); sspart (may be null):
// per-row sums of squares of the four finished 128-column blocks at sspart[(4 p + j) * m_pad + row]
int launch_solve_panel_fused(hipStream_t s, double* vt, int64_t ldv, int64_t m_pad, const double* packed, int64_t n_pad, int64_t p,
                             const double* winv, double* sspart, int64_t ss_stride) {
  if (m_pad <= 0) return 0;
  if (ss_stride <= 0) ss_stride = m_pad;
  if (m_pad % 128) { set_error("solve_panel_fused: m_pad must be a multiple of 128"); return GPRC_ERR_ARG; }
  GPRC_TRY(ensure_gemm_attrs());
  const double M = (double)m_pad;
  ProfScope ps(s, PK_SOLVE_PANEL, M * 128.0 * 128.0 * TPP + 2.0 * M * 128.0 * 128.0 * (TPP * (TPP - 1) / 2), 8.0 * 2.0 * M * NB);
  const size_t smem = G_SMEM_DOUBLES * sizeof(double);
  if (sspart) hipLaunchKernelGGL(solve_panel_fused_kernel<true>, dim3((unsigned)(m_pad / 128)), dim3(256), smem, s, vt, ldv, packed, n_pad, (int)p, winv, sspart, ss_stride);
  else hipLaunchKernelGGL(solve_panel_fused_kernel<false>, dim3((unsigned)(m_pad / 128)), dim3(256), smem, s, vt, ldv, packed, n_pad, (int)p, winv, sspart, ss_stride);
  GPRC_LAUNCH_CHECK();
  return 0;
}

int launch_trsm_panel(hipStream_t s, double* X, int64_t ldx, int64_t M, const double* winv, double* ssq) {
  if (M <= 0) return 0;
  if (M % 128) { set_error("trsm_panel: M must be a multiple of 128"); return GPRC_ERR_ARG; }
  GPRC_TRY(ensure_gemm_attrs());
  ProfScope ps(s, PK_TRSM_PANEL, 1.0 * M * 128 * 128, 8.0 * 2 * M * 128);
  if (ssq) hipLaunchKernelGGL(trsm_panel_kernel<true>, dim3((unsigned)(M / 128)), dim3(256), G_SMEM_DOUBLES * sizeof(double), s, X, ldx, winv, ssq);
  else hipLaunchKernelGGL(trsm_panel_kernel<false>, dim3((unsigned)(M / 128)), dim3(256), G_SMEM_DOUBLES * sizeof(double), s, X, ldx, winv, ssq);
  GPRC_LAUNCH_CHECK();
  return 0;
}

int launch_gemm_nt(hipStream_t s, double* C, int64_t ldc, const double* A, int64_t lda, const double* B, int64_t ldb,
                   int64_t M, int64_t N, int64_t K, int lower, int kind) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  // (K: the tile loop runs two k-tiles of 16 per iteration and needs at least four: every caller's K is a multiple of 128)
  if (M % 128 || N % 128 || K % (2 * G_KB) || K < 4 * G_KB || (lda & 1) || (ldb & 1)) { set_error("gemm_nt: bad shape"); return GPRC_ERR_ARG; }
  GPRC_TRY(ensure_gemm_attrs());
  const int64_t tiles = (M / 128) * (N / 128);
  if (tiles > 0x7fffffff) { set_error("gemm_nt: too many tiles"); return GPRC_ERR_ARG; }
  const double useful = lower ? 0.5 : 1.0;  // algorithmic: the lower triangle only
  ProfScope ps(s, kind, 2.0 * M * N * K * useful, 8.0 * (2.0 * M * N * useful + (M + N) * (double)K));
  constexpr int pg = 0;   // (a persistent grid of N workgroups striding over the tile list measured 4 % slower: DESIGN.md 3)
  const dim3 grid((unsigned)((pg > 0 && tiles > pg) ? pg : tiles)), block(256);
  const size_t smem = G_SMEM_DOUBLES * sizeof(double);
  const int tm = (int)(M / 128), tn = (int)(N / 128);
  constexpr int group = 8;  // 8 x 8 concurrent tiles per XCD share 16 strips; 4..32 measured within 0.5 %
  if (kind == PK_SOLVE_UPDATE) hipLaunchKernelGGL((gemm_nt_kernel<PK_SOLVE_UPDATE>), grid, block, smem, s, C, ldc, A, lda, B, ldb, tm, tn, (int)K, lower, group);
  else if (kind == PK_COV_SYRK) hipLaunchKernelGGL((gemm_nt_kernel<PK_COV_SYRK>), grid, block, smem, s, C, ldc, A, lda, B, ldb, tm, tn, (int)K, lower, group);
  else hipLaunchKernelGGL((gemm_nt_kernel<PK_GEMM_INNER>), grid, block, smem, s, C, ldc, A, lda, B, ldb, tm, tn, (int)K, lower, group);
  GPRC_LAUNCH_CHECK();
  return 0;
}

// part: 0 all tiles; 1 only the lower tiles of the first target's diagonal block; 2 everything but those
int launch_trailing_update(hipStream_t s, double* packed, int64_t n_pad, int64_t p, int64_t q_begin, int64_t q_end,
                           int64_t q_stride) {
  const int64_t P = n_pad / NB;
  if (q_begin <= p || q_stride <= 0) { set_error("trailing_update: bad panel range"); return GPRC_ERR_ARG; }
  if (q_end > P) q_end = P;
  int64_t tiles = 0, nt = 0;
  for (int64_t q = q_begin; q < q_end; q += q_stride) { tiles += (int64_t)TPP * TPP * (P - q) - TPP * (TPP - 1) / 2; ++nt; }
  if (tiles <= 0) return 0;
  GPRC_TRY(ensure_gemm_attrs());
  double fl = 0.0, by = 0.0;  // algorithmic: lower triangle of the 512-wide diagonal block + everything below it
  for (int64_t q = q_begin; q < q_end; q += q_stride) {
    const double rows = (double)(n_pad - q * NB);
    const double elems = rows * NB - 0.5 * NB * (double)(NB - 1);
    fl += 2.0 * elems * NB;
    by += 8.0 * (2.0 * elems + rows * NB);
  }
  ProfScope ps(s, PK_TRAILING, fl, by);

  constexpr int pg = 0;
  const unsigned grid = (unsigned)((pg > 0 && tiles > pg) ? pg : tiles);
  hipLaunchKernelGGL(trailing_kernel, dim3(grid), dim3(256), G_SMEM_DOUBLES * sizeof(double), s, packed, n_pad, (int)p,
                     (int)q_begin, (int)q_stride, (int)nt, (int)tiles);
  GPRC_LAUNCH_CHECK();
  return 0;
}

// target panels q_begin, q_begin + q_stride, ... < q_end updated with the source panels [p_begin, p_end) in one pass
int launch_trailing_range(hipStream_t s, double* packed, int64_t n_pad, int64_t p_begin, int64_t p_end, int64_t q_begin, int64_t q_end,
                          int64_t q_stride) {
  const int64_t P = n_pad / NB;
  if (q_end > P) q_end = P;
  if (p_begin < 0 || p_end <= p_begin || q_stride <= 0) return 0;
  if (q_begin < p_end) { set_error("trailing_range: a target panel is not behind the source range"); return GPRC_ERR_ARG; }
  if (q_begin >= q_end) return 0;
  GPRC_TRY(ensure_gemm_attrs());
  int64_t tiles = 0, nt = 0;
  double fl = 0.0, by = 0.0;
  const double K = (double)(p_end - p_begin) * NB;
  for (int64_t q = q_begin; q < q_end; q += q_stride) {
    tiles += (int64_t)TPP * TPP * (P - q) - TPP * (TPP - 1) / 2;
    ++nt;
    const double rows = (double)(n_pad - q * NB);
    const double elems = rows * NB - 0.5 * NB * (double)(NB - 1);
    fl += 2.0 * elems * K;
    by += 8.0 * (2.0 * elems + rows * K);
  }
  ProfScope ps(s, PK_TRAILING_LEFT, fl, by);
  hipLaunchKernelGGL(trailing_range_kernel, dim3((unsigned)tiles), dim3(256), G_SMEM_DOUBLES * sizeof(double), s, packed, n_pad,
                     (int)p_begin, (int)p_end, (int)q_begin, (int)q_stride, (int)nt, (int)tiles);
  GPRC_LAUNCH_CHECK();
  return 0;
}

// left-looking update of target panels [q_begin, q_end) with every panel before q_begin (GPRC_KCHUNK: in K-chunks, see solve_left)
int launch_trailing_left(hipStream_t s, double* packed, int64_t n_pad, int64_t q_begin, int64_t q_end) {
  static const int64_t kchunk = [] { const char* e = std::getenv("GPRC_KCHUNK"); return e ? std::atoll(e) : 0; }();
  if (kchunk <= 0) return launch_trailing_range(s, packed, n_pad, 0, q_begin, q_begin, q_end, 1);
  for (int64_t p0 = 0; p0 < q_begin; p0 += kchunk)
    GPRC_TRY(launch_trailing_range(s, packed, n_pad, p0, std::min(q_begin, p0 + kchunk), q_begin, q_end, 1));
  return 0;
}

}  // namespace gprc
