// Which operand does each BLGP bit of v_mfma_f64_16x16x4_f64 negate?  (On f64 MFMA the BLGP field is the
// NEG modifier set.)  Exact-integer data, so any sign error is visible.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int BLGP>
__global__ void k(const double* A, const double* B, const double* C, double* D) {
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)], b = B[(l >> 4) * 16 + (l & 15)];
  double4_t c;
  for (int r = 0; r < 4; ++r) c[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, BLGP);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}
int main() {
  double hA[64], hB[64], hC[256], hD[256], AB[256];
  for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 4; ++kk) hA[i * 4 + kk] = (i + 1) * 10 + kk;
  for (int kk = 0; kk < 4; ++kk) for (int j = 0; j < 16; ++j) hB[kk * 16 + j] = (kk + 1) * 100 + 3 * j * j + j;
  for (int i = 0; i < 256; ++i) hC[i] = 1000000.0 + 7 * i;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j]; AB[i * 16 + j] = s; }
  double *A, *B, *C, *D;
  hipMalloc(&A, 512); hipMalloc(&B, 512); hipMalloc(&C, 2048); hipMalloc(&D, 2048);
  hipMemcpy(A, hA, 512, hipMemcpyHostToDevice); hipMemcpy(B, hB, 512, hipMemcpyHostToDevice); hipMemcpy(C, hC, 2048, hipMemcpyHostToDevice);
  auto check = [&](int blgp) {
    hipMemcpy(hD, D, 2048, hipMemcpyDeviceToHost);
    const char* names[4] = {"+AB+C", "-AB+C", "+AB-C", "-AB-C"};
    for (int v = 0; v < 4; ++v) {
      bool ok = true;
      for (int i = 0; i < 256; ++i) { double e = ((v & 1) ? -AB[i] : AB[i]) + ((v & 2) ? -hC[i] : hC[i]); if (hD[i] != e) ok = false; }
      if (ok) { printf("blgp=%d -> D = %s\n", blgp, names[v]); return; }
    }
    printf("blgp=%d -> no match (D[0]=%f AB[0]=%f C[0]=%f)\n", blgp, hD[0], AB[0], hC[0]);
  };
  k<0><<<1, 64>>>(A, B, C, D); hipDeviceSynchronize(); check(0);
  k<1><<<1, 64>>>(A, B, C, D); hipDeviceSynchronize(); check(1);
  k<2><<<1, 64>>>(A, B, C, D); hipDeviceSynchronize(); check(2);
  k<3><<<1, 64>>>(A, B, C, D); hipDeviceSynchronize(); check(3);
  k<4><<<1, 64>>>(A, B, C, D); hipDeviceSynchronize(); check(4);
  return 0;
}
