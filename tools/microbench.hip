// microbench.hip -- device-rate probes that size the hot path's two rooflines on the box at hand:
//   (1) fp64 z<-z^2+c issue rate (the iterate phase), by waves per SIMD and by per-lane ILP;
//   (2) random-address u64 atomic-add rate (the scatter phase), by footprint and scope.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o tools/microbench tools/microbench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ double step(double cr, double ci, double &r, double &i) {
  const double ii = i * i;
  const double t = __builtin_fma(r, r, -ii);
  const double nr = cr + t;
  const double ni = __builtin_fma(r + r, i, ci);
  r = nr; i = ni;
  return __builtin_fma(ni, ni, nr * nr);
}

// ILP independent orbits per lane, n steps each, sticky escape flag (no exec changes in the loop)
template <int ILP>
__global__ void __launch_bounds__(256) iterate_kernel(int n, double *out, unsigned long long *clk) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  double cr[ILP], ci[ILP], r[ILP], i[ILP];
  unsigned esc = 0;
#pragma unroll
  for (int k = 0; k < ILP; ++k) {
    // points inside the main cardioid: never escape, values stay bounded
    cr[k] = -0.1 + 1e-4 * (tid % 977) + 1e-3 * k;
    ci[k] = 0.05 + 1e-4 * (tid % 613);
    r[k] = cr[k]; i[k] = ci[k];
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  for (int s = 0; s < n; ++s) {
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
      esc |= (step(cr[k], ci[k], r[k], i[k]) > 4.0) ? 1u : 0u;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
  double acc = esc;
#pragma unroll
  for (int k = 0; k < ILP; ++k) acc += r[k] + i[k];
  out[tid] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = rt1 - rt0; }
}

template <int ILP>
void run_iterate(int waves_per_simd, int n) {
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (waves_per_simd blocks of 4 waves)
  double *out; unsigned long long *clk;
  CK(hipMalloc(&out, sizeof(double) * blocks * 256));
  CK(hipMalloc(&clk, 16));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(iterate_kernel<ILP>, dim3(blocks), dim3(256), 0, 0, n / 8, out, clk);  // warm-up
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(iterate_kernel<ILP>, dim3(blocks), dim3(256), 0, 0, n, out, clk);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
  const double iters = (double) blocks * 256 * ILP * n;
  const double rate = iters / (ms * 1e-3);
  const double mhz = (double) h[0] / ((double) h[1] / 100.0);  // s_memrealtime ticks at 100 MHz
  printf("iterate ilp=%d waves/simd=%d : %.3f ms  %.3f Titer/s  %.2f TFLOP/s(10/iter)  issue(8 ops/iter)=%.1f%% of 39.3T  clk=%.0f MHz\n",
         ILP, waves_per_simd, ms, rate / 1e12, rate * 10 / 1e12, 100.0 * rate * 8 / 39.32e12, mhz);
  CK(hipFree(out)); CK(hipFree(clk));
}


// The LONG stage's hand-written chunk (two orbits per lane, doubled coordinates: 6 fp64 instructions +
// 1 compare per orbit-step), as in cudabrot_amd/csrc/draw_wave.hip (CB_STEP2), so that its bare rate can
// be read next to the product kernel's.  Variants: 0 = scalar live masks + exact lane-step count (the
// product's), 1 = live masks only (one s_and per orbit-step), 2 = vector instructions only.
#define MB_V0(la, c0) "s_and_b64 " la ", " la ", " c0 "\n\t"
#define MB_STEP2_GEN(S1, S2, S3, S4, S5, S6)              \
  "v_mul_f64 %[a0], %[ia], %[ia]\n\t"                     \
  "v_mul_f64 %[a1], %[ib], %[ib]\n\t"                     \
  S1                                                      \
  "v_fma_f64 %[a0], %[ra], %[ra], -%[a0]\n\t"             \
  S2                                                      \
  "v_fma_f64 %[a1], %[rb], %[rb], -%[a1]\n\t"             \
  S3                                                      \
  "v_fma_f64 %[ia], %[ra], %[ia], %[cia]\n\t"             \
  S4                                                      \
  "v_fma_f64 %[ib], %[rb], %[ib], %[cib]\n\t"             \
  S5                                                      \
  "v_fma_f64 %[ra], %[a0], 0.5, %[cra]\n\t"               \
  S6                                                      \
  "v_fma_f64 %[rb], %[a1], 0.5, %[crb]\n\t"               \
  "v_mul_f64 %[a0], %[ra], %[ra]\n\t"                     \
  "v_mul_f64 %[a1], %[rb], %[rb]\n\t"                     \
  "v_fma_f64 %[a0], %[ia], %[ia], %[a0]\n\t"              \
  "v_fma_f64 %[a1], %[ib], %[ib], %[a1]\n\t"              \
  "v_cmp_nlt_f64_e64 %[c0], %[k16], %[a0]\n\t"            \
  "v_cmp_nlt_f64_e64 %[c1], %[k16], %[a1]\n\t"
#define MB_STEP2 MB_STEP2_GEN("s_and_b64 %[la], %[la], %[c0]\n\t", "s_and_b64 %[lb], %[lb], %[c1]\n\t", \
                              "s_bcnt1_i32_b64 %[t0], %[la]\n\t", "s_bcnt1_i32_b64 %[t1], %[lb]\n\t",   \
                              "s_add_u32 %[cnt], %[cnt], %[t0]\n\t", "s_add_u32 %[cnt], %[cnt], %[t1]\n\t")
#define MB_LATCH2 MB_STEP2_GEN("s_and_b64 %[la], %[la], %[c0]\n\t", "s_and_b64 %[lb], %[lb], %[c1]\n\t", "", "", "", "")
#define MB_PURE2 MB_STEP2_GEN("", "", "", "", "", "")
#define MB_X4(M) M M M M
#define MB_X32(M) MB_X4(M) MB_X4(M) MB_X4(M) MB_X4(M) MB_X4(M) MB_X4(M) MB_X4(M) MB_X4(M)

template <int VARIANT>
__global__ void __launch_bounds__(256) chunk_kernel(int n_chunks, double *out, unsigned *cnt_out) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  double cra = -0.2 + 2e-4 * (tid % 977), cia = 0.1 + 2e-4 * (tid % 613);      // doubled coordinates
  double crb = -0.24 + 2e-4 * (tid % 811), cib = 0.08 + 2e-4 * (tid % 577);
  double ra = cra, ia = cia, rb = crb, ib = cib;
  unsigned total = 0;
  // the product rotates s_setprio among the four waves of a SIMD; here every wave keeps the level of its slot
  const unsigned slot = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11)) & 3u;
  if (slot == 0) __builtin_amdgcn_s_setprio(0);
  if (slot == 1) __builtin_amdgcn_s_setprio(1);
  if (slot == 2) __builtin_amdgcn_s_setprio(2);
  if (slot == 3) __builtin_amdgcn_s_setprio(3);
  for (int c = 0; c < n_chunks; ++c) {
    unsigned long long la = ~0ull, lb = ~0ull, c0, c1;
    unsigned cnt, t0, t1;
    double a0, a1;
#define MB_OPERANDS                                                                                          \
    : [ra] "+v"(ra), [ia] "+v"(ia), [rb] "+v"(rb), [ib] "+v"(ib), [la] "+s"(la), [lb] "+s"(lb), [a0] "=&v"(a0), \
      [a1] "=&v"(a1), [c0] "=&s"(c0), [c1] "=&s"(c1), [cnt] "=&s"(cnt), [t0] "=&s"(t0), [t1] "=&s"(t1)          \
    : [cra] "v"(cra), [cia] "v"(cia), [crb] "v"(crb), [cib] "v"(cib), [k16] "s"(16.0) : "scc"
    if (VARIANT == 0) {
      asm volatile("s_mov_b32 %[cnt], 0\n\ts_mov_b64 %[c0], -1\n\ts_mov_b64 %[c1], -1\n\t" MB_X32(MB_STEP2)
                   "s_and_b64 %[la], %[la], %[c0]\n\ts_and_b64 %[lb], %[lb], %[c1]\n\t" MB_OPERANDS);
    } else if (VARIANT == 1) {
      asm volatile("s_mov_b32 %[cnt], 0\n\ts_mov_b64 %[c0], -1\n\ts_mov_b64 %[c1], -1\n\t" MB_X32(MB_LATCH2)
                   "s_and_b64 %[la], %[la], %[c0]\n\ts_and_b64 %[lb], %[lb], %[c1]\n\t" MB_OPERANDS);
    } else {
      asm volatile("s_mov_b32 %[cnt], 0\n\t" MB_X32(MB_PURE2)
                   "s_and_b64 %[la], %[la], %[c0]\n\ts_and_b64 %[lb], %[lb], %[c1]\n\t" MB_OPERANDS);
    }
#undef MB_OPERANDS
    total += cnt + (unsigned) __popcll(la & lb);
  }
  out[tid] = ra + ia + rb + ib;
  if ((threadIdx.x & 63) == 0) cnt_out[tid >> 6] = total;
}

template <int VARIANT>
void run_chunk(int waves_per_simd, int n_chunks) {
  const int blocks = 256 * waves_per_simd;
  double *out; unsigned *cnt;
  CK(hipMalloc(&out, sizeof(double) * blocks * 256));
  CK(hipMalloc(&cnt, sizeof(unsigned) * blocks * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(chunk_kernel<VARIANT>, dim3(blocks), dim3(256), 0, 0, n_chunks / 8, out, cnt);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(chunk_kernel<VARIANT>, dim3(blocks), dim3(256), 0, 0, n_chunks, out, cnt);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double iters = (double) blocks * 256 * 2 * 32 * n_chunks;
  const double rate = iters / (ms * 1e-3);
  static const char *const kName[] = {"live masks + exact count", "live masks only", "vector only"};
  printf("asm chunk (2 orbits/lane, 6+1 per step, %s) waves/simd=%d : %.3f ms  %.3f Titer/s  %.1f TFLOP/s(10/iter)  "
         "issue(7 slots/iter)=%.1f%% of 39.3T\n",
         kName[VARIANT], waves_per_simd, ms, rate / 1e12, rate * 10 / 1e12, 100.0 * rate * 7 / 39.32e12);
  CK(hipFree(out)); CK(hipFree(cnt));
}

__device__ __forceinline__ uint32_t xs(uint32_t &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

// mode 0: agent-scope atomic add u64; 1: workgroup-scope atomic add u64; 2: plain u64 store;
// 3: agent atomic add u32; 4: agent atomic u64 with lanes of a wave in one 4 KiB window
template <int MODE>
__global__ void __launch_bounds__(256) scatter_kernel(unsigned long long *buf, unsigned long long n_elems, int per_lane) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t s = tid * 2654435761u + 12345u;
  uint32_t sw = (tid >> 6) * 2246822519u + 777u;  // wave-uniform stream for MODE 4
  for (int k = 0; k < per_lane; ++k) {
    unsigned long long idx = (((unsigned long long) xs(s) << 32) | xs(s)) % n_elems;
    if (MODE == 4) {
      const unsigned long long base = ((((unsigned long long) xs(sw) << 32) | xs(sw)) % (n_elems - 512)) & ~511ull;
      idx = base + (xs(s) & 511u);
    }
    if (MODE == 0 || MODE == 4) __hip_atomic_fetch_add(buf + idx, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 1) __hip_atomic_fetch_add(buf + idx, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 2) buf[idx] = k;
    if (MODE == 3) __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(buf) + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int MODE>
void run_scatter(const char *name, unsigned long long *buf, unsigned long long n_elems, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;
  const int per_lane = 512;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(scatter_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, buf, n_elems, 16);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(scatter_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, buf, n_elems, per_lane);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double ops = (double) blocks * 256 * per_lane;
  printf("scatter %-22s footprint=%7.1f MiB waves/simd=%d : %.3f ms  %.2f Gop/s  (%.1f GB/s at 16 B/op)\n", name,
         n_elems * 8.0 / 1048576.0, waves_per_simd, ms, ops / (ms * 1e-3) / 1e9, ops * 16 / (ms * 1e-3) / 1e9);
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s  CUs=%d  clock=%d kHz  mem=%.1f GiB\n", p.name, p.multiProcessorCount, p.clockRate, p.totalGlobalMem / 1073741824.0);
  for (int w = 1; w <= 8; w *= 2) run_iterate<1>(w, 40000);
  for (int w = 1; w <= 4; w *= 2) run_iterate<2>(w, 40000);
  for (int w = 1; w <= 2; w *= 2) run_iterate<4>(w, 40000);
  for (int w = 1; w <= 4; w *= 2) run_chunk<0>(w, 1500);
  for (int w = 1; w <= 4; w *= 2) run_chunk<1>(w, 1500);
  for (int w = 1; w <= 4; w *= 2) run_chunk<2>(w, 1500);
  if (getenv("MB_SKIP_SCATTER")) return 0;
  const unsigned long long small = 4096ull * 4096ull, big = 20000ull * 20000ull;
  unsigned long long *buf; CK(hipMalloc(&buf, big * 8)); CK(hipMemset(buf, 0, big * 8));
  for (int w = 2; w <= 8; w *= 2) run_scatter<0>("u64 atomic agent", buf, small, w);
  run_scatter<0>("u64 atomic agent", buf, big, 4);
  run_scatter<1>("u64 atomic workgroup", buf, small, 4);
  run_scatter<1>("u64 atomic workgroup", buf, big, 4);
  run_scatter<3>("u32 atomic agent", buf, small, 4);
  run_scatter<2>("u64 plain store", buf, small, 4);
  run_scatter<2>("u64 plain store", buf, big, 4);
  run_scatter<4>("u64 atomic 4KiB-window", buf, small, 4);
  run_scatter<4>("u64 atomic 4KiB-window", buf, big, 4);
  CK(hipFree(buf));
  return 0;
}
