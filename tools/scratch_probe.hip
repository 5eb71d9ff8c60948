// Does a resident kernel that uses scratch (spilled registers) keep OTHER queues' kernels off the GPU?
// Kernel A: 512 workgroups of 256 threads, two per CU, spinning for ~5 ms -- built with and without a private array that
// the compiler must keep in scratch.  Kernel B: one wave, on another stream, launched right after A.  Printed: when B
// ended relative to A's start and end.        build: hipcc --offload-arch=gfx950 -O3 -o tools/build/scratch_probe tools/scratch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool kScratch>
__global__ void __launch_bounds__(256, 2) spin_kernel(unsigned long long ticks, uint32_t *out, uint32_t salt) {
  uint32_t acc = threadIdx.x;
  if (kScratch) {
    volatile uint32_t heap[24];  // dynamically indexed: lives in scratch
    for (uint32_t k = 0; k < 24; ++k) heap[k] = salt * k + threadIdx.x;
    acc += heap[(salt + threadIdx.x) % 24u];
  }
  const unsigned long long t0 = wall_clock64();
  for (uint32_t turn = 0; turn < (1u << 20) && wall_clock64() - t0 < ticks; ++turn) __builtin_amdgcn_s_sleep(16);
  if (acc == 0xdeadbeefu) out[0] = acc;
}
__global__ void __launch_bounds__(64) one_wave_kernel(unsigned long long *stamp) {
  if (threadIdx.x == 0) stamp[0] = wall_clock64();
}
__global__ void __launch_bounds__(64) stamp_kernel(unsigned long long *stamp) {
  if (threadIdx.x == 0) stamp[0] = wall_clock64();
}

int main() {
  hipStream_t s0, s1;
  CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  unsigned long long *stamps;
  uint32_t *out;
  CHECK(hipMalloc(&stamps, 4 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&out, 64));
  for (int with_scratch = 0; with_scratch < 2; ++with_scratch) {
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipMemset(stamps, 0, 4 * sizeof(unsigned long long)));
      CHECK(hipDeviceSynchronize());
      hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s0, stamps + 0);
      if (with_scratch) {
        hipLaunchKernelGGL(spin_kernel<true>, dim3(512), dim3(256), 0, s0, 500000ull, out, 3u + rep);
      } else {
        hipLaunchKernelGGL(spin_kernel<false>, dim3(512), dim3(256), 0, s0, 500000ull, out, 3u + rep);
      }
      hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s0, stamps + 1);
      // give A a head start of a few hundred microseconds on the host side, then B on the other stream
      for (volatile int spin = 0; spin < 2000000; ++spin) {}
      hipLaunchKernelGGL(one_wave_kernel, dim3(1), dim3(64), 0, s1, stamps + 2);
      CHECK(hipDeviceSynchronize());
      unsigned long long h[4];
      CHECK(hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost));
      printf("A %s scratch: A ran %.3f ms; the one-wave kernel on the other stream ended %.3f ms after A's start\n",
             with_scratch ? "with" : "without", (h[1] - h[0]) / 1e5, ((double) h[2] - (double) h[0]) / 1e5);
    }
  }
  return 0;
}
