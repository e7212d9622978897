// Microbenchmark + layout check for v_mfma_f64_16x16x4_f64 on gfx950 (MI355X).
// Test infrastructure only: measures the fp64 MFMA / FMA issue rates the roofline in DESIGN.md is
// priced against, and verifies the documented A/B/C lane maps with exact-integer asymmetric data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double double4_t __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2);} } while (0)

// one wave: D(16x16) = A(16x4) * B(4x16); row-major host arrays
__global__ void layout_kernel(const double* A, const double* B, double* D) {
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];      // A[i=l&15][k=l>>4]
  double b = B[(l >> 4) * 16 + (l & 15)];     // B[k=l>>4][j=l&15]
  double4_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) + 4 * r, col = l & 15;  // documented f64 C/D map
    D[row * 16 + col] = c[r];
  }
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_rate(double* out, int iters, double seed) {
  double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
  double4_t acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void fma_rate(double* out, int iters, double seed) {
  double a = 1.0 + seed * 1e-9, b = seed * 1e-9;
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  if (s == 12345.678) out[0] = s;
}

__global__ void copy_kernel(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}
__global__ void write_kernel(double2* __restrict__ out, size_t n, double v) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  double2 x; x.x = v; x.y = v;
  for (; i < n; i += stride) out[i] = x;
}

template <typename F>
double time_ms(F f, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s  CUs=%d  clock=%d kHz  mem=%.1f GiB  LDS/block=%zu\n", p.name, p.multiProcessorCount, p.clockRate, p.totalGlobalMem / 1073741824.0, p.sharedMemPerBlock);
  // ---- layout check
  std::vector<double> A(64), B(64), D(256), R(256, 0.0);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = (i + 1) * 10 + k;     // asymmetric
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (k + 1) * 100 + 3 * j * j + j;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
  double *dA, *dB, *dD; CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
  CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
  layout_kernel<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
  CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
  int bad = 0; for (int i = 0; i < 256; ++i) if (D[i] != R[i]) ++bad;
  printf("LAYOUT mfma_f64_16x16x4: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);

  double* out; CK(hipMalloc(&out, 64));
  const int iters = 4096;
  int cus = p.multiProcessorCount;
  for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
    int grid = cus * wgs_per_cu;
    auto report = [&](const char* name, double ms, double flops) { printf("%-28s wg/cu=%d  %.3f ms  %.2f TFLOP/s\n", name, wgs_per_cu, ms, flops / ms * 1e-9); };
    double fl = (double)grid * 4 /*waves*/ * iters * 2048.0;
    report("mfma_f64 16x16x4 nacc=1", time_ms([&] { mfma_rate<1><<<grid, 256>>>(out, iters, 1.0); }, 5), fl * 1);
    report("mfma_f64 16x16x4 nacc=2", time_ms([&] { mfma_rate<2><<<grid, 256>>>(out, iters, 1.0); }, 5), fl * 2);
    report("mfma_f64 16x16x4 nacc=4", time_ms([&] { mfma_rate<4><<<grid, 256>>>(out, iters, 1.0); }, 5), fl * 4);
    report("mfma_f64 16x16x4 nacc=16", time_ms([&] { mfma_rate<16><<<grid, 256>>>(out, iters, 1.0); }, 5), fl * 16);
    double ff = (double)grid * 256 * iters * 2.0;
    report("v_fma_f64 nacc=4", time_ms([&] { fma_rate<4><<<grid, 256>>>(out, iters, 1.0); }, 5), ff * 4);
    report("v_fma_f64 nacc=16", time_ms([&] { fma_rate<16><<<grid, 256>>>(out, iters, 1.0); }, 5), ff * 16);
  }
  {
    int grid = cus * 4;  // 4 waves per SIMD
    double ff = (double)grid * 256 * iters * 2.0;
    double ms = time_ms([&] { fma_rate<8><<<grid, 256>>>(out, iters, 1.0); }, 5);
    printf("%-28s wg/cu=4  %.3f ms  %.2f TFLOP/s\n", "v_fma_f64 nacc=8", ms, ff * 8 / ms * 1e-9);
  }
  // ---- HBM
  size_t nbytes = (size_t)4 << 30; size_t n2 = nbytes / 16;
  double2 *src, *dst; CK(hipMalloc(&src, nbytes)); CK(hipMalloc(&dst, nbytes));
  CK(hipMemset(src, 1, nbytes));
  double ms = time_ms([&] { copy_kernel<<<cus * 8, 256>>>(src, dst, n2); }, 5);
  printf("HBM copy 4GiB->4GiB: %.3f ms  %.2f TB/s (read+write)\n", ms, 2.0 * nbytes / ms * 1e-9);
  ms = time_ms([&] { write_kernel<<<cus * 8, 256>>>(dst, n2, 1.5); }, 5);
  printf("HBM write 4GiB: %.3f ms  %.2f TB/s\n", ms, 1.0 * nbytes / ms * 1e-9);
  return bad ? 1 : 0;
}
