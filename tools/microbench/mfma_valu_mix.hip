// Does a VALU (or SALU, LDS) instruction cost f64-MFMA pipe time on gfx950?  One wave per SIMD (a 256-thread workgroup) or two (512 threads),
// a stream of v_mfma_f64_16x16x4 on 16 independent accumulators (no dependency stalls: 64 cycles of pipe per MFMA), and `extra` other
// instructions behind every MFMA, all in one asm block with fixed registers (the compiler schedules nothing).
// Build: hipcc --offload-arch=gfx950 -O2 mfma_valu_mix.hip -o bin/mfma_valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>

#define MF(i) "v_mfma_f64_16x16x4_f64 v[" #i ":" #i "+7], v[128:129], v[130:131], v[" #i ":" #i "+7]\n\t"
#define STREAM(X) MF(0) X MF(8) X MF(16) X MF(24) X MF(32) X MF(40) X MF(48) X MF(56) X MF(64) X MF(72) X MF(80) X MF(88) X MF(96) X MF(104) X MF(112) X MF(120) X

#define KERNEL(name, X)                                                                                              \
  __global__ __launch_bounds__(512) void name(unsigned long long* cyc, int reps) {                                   \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                                  \
    for (int r = 0; r < reps; ++r)                                                                                   \
      asm volatile(STREAM(X) ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","s20","s21","s22","s23", "memory"); \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                                  \
    if ((threadIdx.x & 63) == 0) atomicMax(&cyc[0], r1 - r0);   /* the slowest wave (arbitration is oldest-first) */    \
  }

KERNEL(k_none, "")
KERNEL(k_valu1, "v_add_u32 v132, v132, v133\n\t")
KERNEL(k_valu2, "v_add_u32 v132, v132, v133\n\tv_add_u32 v134, v134, v133\n\t")
KERNEL(k_valu64, "v_lshl_add_u64 v[132:133], v[132:133], 0, v[134:135]\n\t")
KERNEL(k_salu2, "s_add_u32 s20, s20, 1\n\ts_addc_u32 s21, s21, 0\n\t")
KERNEL(k_cnd2, "v_cndmask_b32 v132, v132, v133, vcc\n\tv_cndmask_b32 v134, v134, v133, vcc\n\t")
KERNEL(k_fma64, "v_fma_f64 v[132:133], v[132:133], v[132:133], v[134:135]\n\t")
KERNEL(k_dsread, "ds_read2_b64 v[132:135], v131 offset1:16\n\t")

int main() {
  unsigned long long* d; hipMalloc(&d, 64);
  struct { const char* n; void (*k)(unsigned long long*, int); } ks[] = {
      {"nothing between", k_none}, {"1 v_add_u32", k_valu1}, {"2 v_add_u32", k_valu2}, {"1 v_lshl_add_u64", k_valu64}, {"s_add_u32 + s_addc_u32", k_salu2},
      {"2 v_cndmask_b32", k_cnd2}, {"1 v_fma_f64", k_fma64}, {"1 ds_read2_b64", k_dsread}};
  const int reps = 64;
  for (int threads : {256, 512}) {
    std::printf("%d threads per workgroup = %d wave(s) per SIMD, one workgroup; ns per MFMA of the SLOWEST wave (ideal: %s)\n", threads, threads / 256, threads == 256 ? "27.1 at 2.36 GHz = 64 cycles" : "54.2: two waves share the pipe");
    for (auto& e : ks) {
      unsigned long long best = ~0ull, c;
      for (int rep = 0; rep < 5; ++rep) {
        hipMemset(d, 0, 8);
        hipLaunchKernelGGL(e.k, dim3(1), dim3(threads), 0, 0, d, reps);
        hipDeviceSynchronize();
        hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
        if (c < best) best = c;
      }
      std::printf("  %-26s %7.2f ns per MFMA per wave\n", e.n, best * 10.0 / (reps * 16));
    }
  }
  return 0;
}
