// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950 (dispatch packet without the barrier bit)?
// hip_ext.h says the flag "is not supported on AMD GFX9xx boards"; this measures what the runtime on this image actually does.
//   A: one workgroup that spins for 2 ms.   B: one workgroup that stamps its start time.
//   in-order B starts after A ends (and the distance is the back-to-back launch gap); an any-order B that overlaps starts before.
// Build: hipcc --offload-arch=gfx950 -O2 anyorder.hip -o bin/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void spin_kernel(unsigned long long ticks, unsigned long long* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    out[0] = t0;
    out[1] = __builtin_amdgcn_s_memrealtime();
  }
}

int main() {
  unsigned long long* d;
  CK(hipMalloc(&d, 64));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned long long h[4];
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemsetAsync(d, 0, 64, s));
      CK(hipStreamSynchronize(s));
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, 200000ULL, d);          // 2 ms at 100 MHz
      if (mode == 0) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, 100ULL, d + 2);
      else hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, 100ULL, d + 2);
      CK(hipGetLastError());
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
      std::printf("%s: B starts %+.1f us relative to the END of A (A ran %.1f us)\n", mode ? "any-order" : "in-order ",
                  ((double)h[2] - (double)h[1]) / 100.0, ((double)h[1] - (double)h[0]) / 100.0);
    }
  }
  return 0;
}
