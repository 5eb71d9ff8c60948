// bank_probe.hip -- does the choice of VECTOR REGISTERS matter to the LONG stage's inner loop?  The four-chain step of
// draw_wide.hip (CBW_L4A: 16 fp64 instructions) with hard-coded registers in three layouts, at two waves per SIMD like the
// kernel: "asis" (what the compiler gave the product instance), "alt" (sources of one instruction spread over the two
// bank pairs, v[4n] / v[4n+2], where the instruction has two or three distinct ones), "same" (every operand v[4n]).
//   hipcc -O3 --offload-arch=gfx950 -o tools/build/bank_probe tools/bank_probe.hip && tools/build/bank_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define STEP(T0, T1, T2, T3, R0, R1, R2, R3, I0, I1, I2, I3, CR0, CR1, CR2, CR3, CI0, CI1, CI2, CI3) \
  "v_mul_f64 " T0 ", " I0 ", " I0 "\n\t"                                                            \
  "v_mul_f64 " T1 ", " I1 ", " I1 "\n\t"                                                            \
  "v_mul_f64 " T2 ", " I2 ", " I2 "\n\t"                                                            \
  "v_mul_f64 " T3 ", " I3 ", " I3 "\n\t"                                                            \
  "v_fma_f64 " T0 ", " R0 ", " R0 ", -" T0 "\n\t"                                                   \
  "v_fma_f64 " T1 ", " R1 ", " R1 ", -" T1 "\n\t"                                                   \
  "v_fma_f64 " T2 ", " R2 ", " R2 ", -" T2 "\n\t"                                                   \
  "v_fma_f64 " T3 ", " R3 ", " R3 ", -" T3 "\n\t"                                                   \
  "v_fma_f64 " I0 ", " R0 ", " I0 ", " CI0 "\n\t"                                                   \
  "v_fma_f64 " I1 ", " R1 ", " I1 ", " CI1 "\n\t"                                                   \
  "v_fma_f64 " I2 ", " R2 ", " I2 ", " CI2 "\n\t"                                                   \
  "v_fma_f64 " I3 ", " R3 ", " I3 ", " CI3 "\n\t"                                                   \
  "v_fma_f64 " R0 ", " T0 ", 0.5, " CR0 "\n\t"                                                      \
  "v_fma_f64 " R1 ", " T1 ", 0.5, " CR1 "\n\t"                                                      \
  "v_fma_f64 " R2 ", " T2 ", 0.5, " CR2 "\n\t"                                                      \
  "v_fma_f64 " R3 ", " T3 ", 0.5, " CR3 "\n\t"
#define X5(S) S S S S S
#define X60(S) X5(S) X5(S) X5(S) X5(S) X5(S) X5(S) X5(S) X5(S) X5(S) X5(S) X5(S) X5(S)

#define ASIS STEP("v[4:5]", "v[6:7]", "v[12:13]", "v[14:15]", "v[76:77]", "v[82:83]", "v[86:87]", "v[90:91]", \
                  "v[84:85]", "v[88:89]", "v[92:93]", "v[94:95]", "v[64:65]", "v[66:67]", "v[72:73]", "v[70:71]", \
                  "v[68:69]", "v[74:75]", "v[80:81]", "v[124:125]")
// r: 4n, i: 4n+2, cr: 4n, ci: 4n+2 (the cross term's three sources cannot all differ), t: 4n+2
#define ALT STEP("v[6:7]", "v[10:11]", "v[14:15]", "v[18:19]", "v[64:65]", "v[72:73]", "v[80:81]", "v[88:89]", \
                 "v[66:67]", "v[74:75]", "v[82:83]", "v[90:91]", "v[68:69]", "v[76:77]", "v[84:85]", "v[92:93]", \
                 "v[70:71]", "v[78:79]", "v[86:87]", "v[94:95]")
// the cross term's addend in the other pair than its second factor: r: 4n, i: 4n+2, ci: 4n (with r), cr: 4n+2, t: 4n+2
#define ALT2 STEP("v[6:7]", "v[10:11]", "v[14:15]", "v[18:19]", "v[64:65]", "v[72:73]", "v[80:81]", "v[88:89]", \
                  "v[66:67]", "v[74:75]", "v[82:83]", "v[90:91]", "v[70:71]", "v[78:79]", "v[86:87]", "v[94:95]", \
                  "v[68:69]", "v[76:77]", "v[84:85]", "v[92:93]")
#define SAME STEP("v[4:5]", "v[8:9]", "v[12:13]", "v[16:17]", "v[64:65]", "v[68:69]", "v[72:73]", "v[76:77]", \
                  "v[80:81]", "v[84:85]", "v[88:89]", "v[92:93]", "v[96:97]", "v[100:101]", "v[104:105]", "v[108:109]", \
                  "v[112:113]", "v[116:117]", "v[120:121]", "v[124:125]")

#define CLOBBERS                                                                                                          \
  "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v64", "v65", \
      "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81",     \
      "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97",     \
      "v100", "v101", "v104", "v105", "v108", "v109", "v112", "v113", "v116", "v117", "v120", "v121", "v124", "v125"

template <int kVariant>
__global__ void __launch_bounds__(256, 2) probe(unsigned long long *out, int chunks) {
  extern __shared__ uint32_t pad[];  // 80 KiB per workgroup: two workgroups per CU = two waves per SIMD
  if (threadIdx.x == 0) pad[0] = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int c = 0; c < chunks; ++c) {
    if (kVariant == 0) asm volatile(X60(ASIS) ::: CLOBBERS);
    if (kVariant == 1) asm volatile(X60(ALT) ::: CLOBBERS);
    if (kVariant == 2) asm volatile(X60(ALT2) ::: CLOBBERS);
    if (kVariant == 3) asm volatile(X60(SAME) ::: CLOBBERS);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) atomicAdd(out, t1 - t0);
}

int main() {
  unsigned long long *d, h;
  hipMalloc(&d, 8);
  const int chunks = 2000, blocks = 512;
  const char *names[4] = {"asis", "alt", "alt2", "same"};
  for (int rep = 0; rep < 2; ++rep) {
    for (int v = 0; v < 4; ++v) {
      hipMemset(d, 0, 8);
      hipEvent_t a, b;
      hipEventCreate(&a);
      hipEventCreate(&b);
      hipEventRecord(a);
      if (v == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 80 * 1024, 0, d, chunks);
      if (v == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 80 * 1024, 0, d, chunks);
      if (v == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 80 * 1024, 0, d, chunks);
      if (v == 3) hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 80 * 1024, 0, d, chunks);
      hipEventRecord(b);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, a, b);
      hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
      const double per_wave = (double) h / (blocks * 4.0);
      printf("%-5s %.3f ms; cycles per wave and chunk of 960 instructions: %.0f = %.2f per instruction (two waves per SIMD: 8.00 at full issue)\n",
             names[v], ms, per_wave / chunks, per_wave / chunks / 960.0);
    }
  }
  return 0;
}
