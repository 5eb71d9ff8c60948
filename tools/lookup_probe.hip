// lookup_probe.hip -- how many random BYTE loads per second a wave-scheduled kernel can make from a small table
// (the "map for HEAD" lead of DESIGN.md 7: one lookup per sample = 839 M per 7 ms launch = 120 G/s), by table size
// and waves per SIMD, with a little integer work between the loads (the generator's share).
//   hipcc --offload-arch=gfx950 -O3 -o tools/lookup_probe tools/lookup_probe.hip && tools/lookup_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) lookup_kernel(const uint8_t *table, uint32_t mask, int n, uint32_t *out) {
  uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u, acc = 0;
  for (int k = 0; k < n; k += 4) {   // four independent loads in flight, ~10 integer instructions per load
    uint32_t a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x ^= x << 13; x ^= x >> 17; x ^= x << 5;
      a[j] = x & mask;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc += table[a[j]];
  }
  if (acc == 0xffffffffu) out[0] = acc;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  uint8_t *table;
  uint32_t *out;
  const size_t max_bytes = 256u << 20;
  CK(hipMalloc(&table, max_bytes));
  CK(hipMemset(table, 1, max_bytes));
  CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int n = 4096;
  for (size_t mib : {2, 8, 32, 128}) {
    for (int waves : {2, 4, 8}) {
      const int blocks = p.multiProcessorCount * waves;  // 256 threads = 4 waves per block = 1 per SIMD
      const uint32_t mask = (uint32_t) (mib << 20) - 1u;
      hipLaunchKernelGGL(lookup_kernel, dim3(blocks), dim3(256), 0, 0, table, mask, 64, out);
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(lookup_kernel, dim3(blocks), dim3(256), 0, 0, table, mask, n, out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double loads = (double) blocks * 256.0 * n;
      printf("random byte loads, table %4zu MiB, %d waves/SIMD: %.3f ms  %.1f G loads/s\n", mib, waves, ms, loads / ms * 1e-6);
    }
  }
  return 0;
}
