// Dependent-issue latency of the double-precision instructions the 16x16 pivot sweep (diag16, kernels_chol.hip) is made of, one wave
// alone on its SIMD: cycles per instruction of a 256-long dependent chain (s_memtime), and of 4 independent chains interleaved.
// Build: hipcc --offload-arch=gfx950 -O2 dp_latency.hip -o bin/dp_latency
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))

#define CHAIN_KERNEL(name, body)                                                       \
  __global__ void name(double* io, unsigned long long* cyc) {                          \
    double a = io[threadIdx.x], b = io[64 + threadIdx.x], c = io[128 + threadIdx.x], d = io[192 + threadIdx.x]; \
    const double m = io[256 + threadIdx.x];                                            \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                    \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                        \
    asm volatile("s_nop 0" ::: "memory");                                              \
    REP256(body)                                                                       \
    asm volatile("s_nop 0" ::: "memory");                                              \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                        \
    io[threadIdx.x] = a + b + c + d;                                                   \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                    \
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }                      \
  }

CHAIN_KERNEL(k_fma_dep, asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a) : "v"(m));)
CHAIN_KERNEL(k_fma_4, asm volatile("v_fma_f64 %0, %0, %4, %0\n\tv_fma_f64 %1, %1, %4, %1\n\tv_fma_f64 %2, %2, %4, %2\n\tv_fma_f64 %3, %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
CHAIN_KERNEL(k_mul_dep, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(m));)
CHAIN_KERNEL(k_rsq_dep, asm volatile("v_rsq_f64 %0, %0\n\ts_nop 0" : "+v"(a));)
CHAIN_KERNEL(k_movdpp_dep, asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a));)
CHAIN_KERNEL(k_fmacdpp_dep, asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(m));)
CHAIN_KERNEL(k_fmacdpp_4, asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %1, %1, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %2, %2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %3, %3, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
CHAIN_KERNEL(k_fmacdpp_indep_src, asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(m));)
CHAIN_KERNEL(k_cndmask_dep, { int& f = *reinterpret_cast<int*>(&a); asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(f)); })
CHAIN_KERNEL(k_fma32_dep, { float& f = *reinterpret_cast<float*>(&a); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f)); })
CHAIN_KERNEL(k_bperm_dep, { int& f = *reinterpret_cast<int*>(&a); asm volatile("ds_bpermute_b32 %0, %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(f)); })
CHAIN_KERNEL(k_readlane_dep, { int& f = *reinterpret_cast<int*>(&a); int s; asm volatile("v_readlane_b32 %1, %0, 3\n\ts_nop 3\n\tv_mov_b32 %0, %1" : "+v"(f), "=s"(s)); })


typedef double double4_t __attribute__((ext_vector_type(4)));
// acc -> acc: the pipe time of one v_mfma_f64_16x16x4 (back-to-back dependent accumulations are forwarded inside the pipe)
__global__ void k_mfma_acc(double* io, unsigned long long* cyc) {
  double a = io[threadIdx.x];
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  REP256(acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);)
  asm volatile("" : "+v"(acc));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  io[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}
// result -> operand: the MFMA's result is read by a VALU instruction whose result is the next MFMA's operand (the pivot chain's shape)
__global__ void k_mfma_operand(double* io, unsigned long long* cyc) {
  double a = io[threadIdx.x] * 1e-3;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  REP256(acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0); a = acc[0] * 1e-3;)
  asm volatile("" : "+v"(acc));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  io[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

// Does a VALU instruction cost MFMA-pipe time?  16 independent accumulators (no dependency stalls: the pipe is the bound, 64 cycles per MFMA),
// one extra instruction of the given kind behind every MFMA.
#define MIX_KERNEL(name, extra)                                                                                                     \
  __global__ void name(double* io, unsigned long long* cyc) {                                                                      \
    double a = io[threadIdx.x];                                                                                                     \
    double4_t acc[16];                                                                                                              \
    for (int i = 0; i < 16; ++i) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};                                                          \
    int v = threadIdx.x, w = threadIdx.x * 3; unsigned long long q = threadIdx.x; (void)v; (void)w; (void)q;                        \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                                                 \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                                     \
    for (int rep = 0; rep < 16; ++rep) {                                                                                            \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                                              \
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc[i], 0, 0, 0);                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                          \
        extra;                                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                          \
      }                                                                                                                             \
    }                                                                                                                               \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                                     \
    double sum = 0.0;                                                                                                               \
    for (int i = 0; i < 16; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];                                              \
    io[threadIdx.x] = sum + v + w + (double)q;                                                                                      \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                                                 \
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }                                                                   \
  }
MIX_KERNEL(k_mix_none, asm volatile("" ::: "memory"))
MIX_KERNEL(k_mix_valu32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(w)))
MIX_KERNEL(k_mix_valu32x2, asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %0" : "+v"(v), "+v"(w)))
MIX_KERNEL(k_mix_valu64, asm volatile("v_lshl_add_u64 %0, %0, 0, %0" : "+v"(q)))
MIX_KERNEL(k_mix_salu, { int sx; asm volatile("s_add_u32 %0, 1, 2" : "=s"(sx)); (void)sx; })
MIX_KERNEL(k_mix_fma64, asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a)))

int main() {
  double* io; unsigned long long* cyc;
  hipMalloc(&io, 512 * 8); hipMalloc(&cyc, 64);
  double h[512];
  for (int i = 0; i < 512; ++i) h[i] = 1.0 + 1e-9 * i;
  struct { const char* name; void (*k)(double*, unsigned long long*); int n; } ks[] = {
      {"v_fma_f64 dependent", k_fma_dep, 256}, {"v_fma_f64 4 independent chains", k_fma_4, 1024}, {"v_mul_f64 dependent", k_mul_dep, 256},
      {"v_rsq_f64 dependent (+s_nop 0)", k_rsq_dep, 256}, {"v_mov_b64_dpp dependent (+s_nop 1)", k_movdpp_dep, 256},
      {"v_fmac_f64_dpp dependent (+s_nop 1)", k_fmacdpp_dep, 256}, {"v_fmac_f64_dpp 4 independent (+s_nop 1 each)", k_fmacdpp_4, 1024},
      {"v_fmac_f64_dpp acc chain, constant dpp source (+s_nop 1)", k_fmacdpp_indep_src, 256},
      {"v_cndmask_b32 dependent", k_cndmask_dep, 256}, {"v_fma_f32 dependent", k_fma32_dep, 256},
      {"ds_bpermute_b32 dependent (+waitcnt)", k_bperm_dep, 256}, {"v_mfma_f64_16x16x4 acc -> acc", k_mfma_acc, 256},
      {"v_mfma_f64_16x16x4 result -> v_mul -> operand", k_mfma_operand, 256},
      {"MFMA stream (16 accumulators), nothing between", k_mix_none, 256}, {"MFMA stream + 1 v_add_u32 per MFMA", k_mix_valu32, 256},
      {"MFMA stream + 2 v_add_u32 per MFMA", k_mix_valu32x2, 256}, {"MFMA stream + 1 v_lshl_add_u64 per MFMA", k_mix_valu64, 256},
      {"MFMA stream + 1 s_add_u32 per MFMA", k_mix_salu, 256}, {"MFMA stream + 1 v_fma_f64 per MFMA", k_mix_fma64, 256}, {"v_readlane_b32 -> v_mov dependent (+s_nop 3)", k_readlane_dep, 256}};
  for (auto& e : ks) {
    unsigned long long c[2] = {0, 0}, best = ~0ull, bestr = 0;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemcpy(io, h, sizeof(h), hipMemcpyHostToDevice);
      hipLaunchKernelGGL(e.k, dim3(1), dim3(64), 0, 0, io, cyc);
      hipDeviceSynchronize();
      hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
      if (c[0] < best) { best = c[0]; bestr = c[1]; }
    }
    std::printf("%-62s %7.2f s_memtime ticks per instruction (%llu / %d; %.2f us by s_memrealtime = %.1f ns each)\n", e.name, (double)best / e.n, best,
                e.n, bestr / 100.0, bestr * 10.0 / e.n);
  }
  // s_memrealtime is the constant 100 MHz counter; ns each x shader clock (~2.1-2.4 GHz) = shader cycles per instruction
  return 0;
}
