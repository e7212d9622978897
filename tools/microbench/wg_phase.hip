// Diagnostic: where and when do the workgroups of a 2-workgroups-per-CU kernel run?  Each workgroup records
// (XCC_ID, HW_ID, start, end) with the constant-rate s_memrealtime clock; the host prints, for a few CUs, the
// timeline of the workgroups that ran there.  Used to check whether co-resident workgroups stay phase-locked.
//   hipcc --offload-arch=gfx950 -O2 wg_phase.hip -o wg_phase && ./wg_phase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>

__global__ __launch_bounds__(256, 2) void probe(unsigned long long* rec, int spin) {
  extern __shared__ double lds[];  // 73,728 B requested: two workgroups per CU, as the GEMM kernels
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  double a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = fma(a, 1.0000001, 1e-9);
  if (a == 12345.0) lds[0] = a;
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    rec[4 * blockIdx.x + 0] = xcc; rec[4 * blockIdx.x + 1] = hw; rec[4 * blockIdx.x + 2] = t0; rec[4 * blockIdx.x + 3] = t1;
  }
}

int main() {
  const int nb = 2048, spin = 20000;
  unsigned long long* d;
  hipMalloc(&d, sizeof(unsigned long long) * 4 * nb);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
  hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 73728, 0, d, spin);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(4 * nb);
  hipMemcpy(h.data(), d, sizeof(unsigned long long) * 4 * nb, hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull;
  for (int b = 0; b < nb; ++b) tmin = std::min(tmin, h[4 * b + 2]);
  // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
  std::map<unsigned, std::vector<int>> per_cu;
  for (int b = 0; b < nb; ++b) {
    const unsigned hw = (unsigned)h[4 * b + 1], xcc = (unsigned)h[4 * b];
    const unsigned key = (xcc << 16) | (hw & 0xFF00);  // xcc, se, sh, cu
    per_cu[key].push_back(b);
  }
  printf("distinct CUs seen: %zu\n", per_cu.size());
  int shown = 0;
  for (auto& kv : per_cu) {
    if (shown++ >= 6) break;
    auto& v = kv.second;
    std::sort(v.begin(), v.end(), [&](int a, int b) { return h[4 * a + 2] < h[4 * b + 2]; });
    printf("xcc %u se %u cu %u:", kv.first >> 16, (kv.first >> 13) & 7, (kv.first >> 8) & 15);
    for (int b : v) printf("  [blk %d wave_slot %u  %llu..%llu]", b, (unsigned)h[4 * b + 1] & 15, h[4 * b + 2] - tmin, h[4 * b + 3] - tmin);
    printf("\n");
  }
  // first-generation pairing: which block ids share a CU at t ~ 0
  int same_parity = 0, pairs = 0;
  for (auto& kv : per_cu) {
    auto& v = kv.second;
    if (v.size() >= 2) { ++pairs; same_parity += ((h[4 * v[0] + 1] & 1) == (h[4 * v[1] + 1] & 1)); }
  }
  printf("first two workgroups per CU: %d pairs, %d with equal wave-slot parity\n", pairs, same_parity);
  return 0;
}
