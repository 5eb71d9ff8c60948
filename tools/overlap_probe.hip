// overlap_probe.hip -- does a memory-bound kernel on a second stream run BESIDE the draw kernel?
//
// The draw kernel fills every SIMD's wave slots for a whole launch; whatever LDS and VGPRs it leaves
// free decide whether the scatter kernels of the previous launch can be co-resident.  This probe
// times (a) a draw launch alone, (b) a streaming read kernel alone, (c) both at once on two
// streams, for several LDS footprints of the streaming kernel.
//
// build: hipcc -O2 --offload-arch=gfx950 -Iinclude tools/overlap_probe.hip -Lcudabrot_amd -lcudabrot_amd
//        -Wl,-rpath,$PWD/cudabrot_amd -o tools/overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "cudabrot_amd.h"

#define CHECK(x)                                                                 \
  do {                                                                           \
    int e_ = (int) (x);                                                          \
    if (e_ != 0) {                                                               \
      printf("error %d at %s:%d: %s\n", e_, __FILE__, __LINE__, cb_error_string(e_)); \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

// grid-stride streaming read: 4 independent 16-byte loads in flight per lane; lds_bytes of dynamic
// LDS are touched so that the allocation is real.
__global__ void __launch_bounds__(256) stream_read(const uint4 *src, size_t n_vec, unsigned *sink) {
  extern __shared__ unsigned lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const size_t stride = (size_t) gridDim.x * blockDim.x;
  size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = lds[(threadIdx.x + 1) & 255];
  for (; i + 3 * stride < n_vec; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    acc += a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// The same with a register allocation of at least kVgprs (a clobbered top register) and, with kScratch, a
// private array indexed at run time (scratch memory): which guests find room beside the draw kernel?
template <int kVgprs, bool kScratch>
__global__ void __launch_bounds__(256) stream_read_shaped(const uint4 *src, size_t n_vec, unsigned *sink) {
  extern __shared__ unsigned lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (kVgprs == 32) asm volatile("" ::: "v31");
  if (kVgprs == 48) asm volatile("" ::: "v47");
  if (kVgprs == 56) asm volatile("" ::: "v55");
  if (kVgprs == 64) asm volatile("" ::: "v63");
  if (kVgprs == 72) asm volatile("" ::: "v71");
  unsigned priv[16];
  if (kScratch) {
    for (int k = 0; k < 16; ++k) priv[k] = lds[(threadIdx.x + k) & 255];
  }
  const size_t stride = (size_t) gridDim.x * blockDim.x;
  size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = lds[(threadIdx.x + 1) & 255];
  for (; i + 3 * stride < n_vec; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    acc += a.x ^ b.y ^ c.z ^ d.w;
    if (kScratch) {
      priv[a.x & 15u] += acc;
      acc ^= priv[b.y & 15u];
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv) {
  const size_t read_bytes = (size_t) 6 << 30;
  cb_fractal_dimensions dims = {};
  dims.w = 4096; dims.h = 4096;
  dims.min_real = -2; dims.max_real = 2; dims.min_imag = -2; dims.max_imag = 2;
  const char *msg = nullptr;
  if (!cb_recompute_pixel_deltas(&dims, &msg)) { printf("dims: %s\n", msg); return 1; }
  cb_iteration_control it = {20000, 20};
  const uint32_t T = 262144, spt = 3200;

  hipStream_t sa, sb;
  CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  void *hist, *states, *ws, *carry, *src, *sink;
  const size_t ws_bytes = cb_scatter_workspace_bytes(&dims, T, spt);
  CHECK(hipMalloc(&hist, (size_t) 4096 * 4096 * 8));
  CHECK(hipMemset(hist, 0, (size_t) 4096 * 4096 * 8));
  CHECK(hipMalloc(&states, cb_rng_state_bytes(T)));
  CHECK(hipMalloc(&ws, ws_bytes));
  CHECK(hipMalloc(&carry, cb_carry_bytes(T)));
  CHECK(hipMemset(carry, 0, cb_carry_bytes(T)));
  CHECK(hipMalloc(&src, read_bytes));
  CHECK(hipMemset(src, 1, read_bytes));
  CHECK(hipMalloc(&sink, 64));
  CHECK(cb_initialize_rng(1337, 0, T, states, sa));

  hipEvent_t a0, a1, b0, b1;
  CHECK(hipEventCreate(&a0)); CHECK(hipEventCreate(&a1));
  CHECK(hipEventCreate(&b0)); CHECK(hipEventCreate(&b1));

  auto draw = [&]() {
    CHECK(cb_draw_buddhabrot(&dims, (cb_pixel *) hist, &it, states, T, spt, nullptr, CB_KERNEL_DEFAULT, ws,
                             ws_bytes, carry, sa));
  };
  for (int k = 0; k < 3; ++k) draw();  // warm up, fill the pipeline of carried orbits
  CHECK(hipDeviceSynchronize());

  float ms;
  for (int k = 0; k < 2; ++k) {
    CHECK(hipEventRecord(a0, sa)); draw(); CHECK(hipEventRecord(a1, sa));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventElapsedTime(&ms, a0, a1));
    printf("draw alone: %.3f ms\n", ms);
  }

  const int lds_cases[] = {1024, 8192, 15360, 16384, 24576, 32768, 65536};
  const int wg_per_cu[] = {1, 2, 4, 8};
  for (int lds : lds_cases) {
    for (int wpc : wg_per_cu) {
      if ((size_t) lds * wpc > 160 * 1024) continue;
      const int grid = 256 * wpc;
      CHECK(hipFuncSetAttribute((const void *) stream_read, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      // alone
      CHECK(hipEventRecord(b0, sb));
      hipLaunchKernelGGL(stream_read, dim3(grid), dim3(256), lds, sb, (const uint4 *) src, read_bytes / 16,
                         (unsigned *) sink);
      CHECK(hipEventRecord(b1, sb));
      CHECK(hipDeviceSynchronize());
      float alone;
      CHECK(hipEventElapsedTime(&alone, b0, b1));
      // beside the draw kernel
      CHECK(hipEventRecord(a0, sa)); draw(); CHECK(hipEventRecord(a1, sa));
      CHECK(hipEventRecord(b0, sb));
      hipLaunchKernelGGL(stream_read, dim3(grid), dim3(256), lds, sb, (const uint4 *) src, read_bytes / 16,
                         (unsigned *) sink);
      CHECK(hipEventRecord(b1, sb));
      CHECK(hipDeviceSynchronize());
      float d, s, span;
      CHECK(hipEventElapsedTime(&d, a0, a1));
      CHECK(hipEventElapsedTime(&s, b0, b1));
      CHECK(hipEventElapsedTime(&span, a0, b1));
      printf("lds %6d B x %d WG/CU: read alone %.3f ms (%.2f TB/s) | together: draw %.3f ms, read %.3f ms (%.2f TB/s), "
             "span from draw start to read end %.3f ms\n",
             lds, wpc, alone, read_bytes / alone * 1e-9, d, s, read_bytes / s * 1e-9, span);
      fflush(stdout);
    }
  }
  // shaped guests: registers / scratch at the LDS sizes a small sort kernel would need, one workgroup per CU
  struct Shape { const char *name; const void *fn; };
  const Shape shapes[] = {
      {"32 VGPR", (const void *) stream_read_shaped<32, false>}, {"48 VGPR", (const void *) stream_read_shaped<48, false>},
      {"56 VGPR", (const void *) stream_read_shaped<56, false>}, {"64 VGPR", (const void *) stream_read_shaped<64, false>},
      {"72 VGPR", (const void *) stream_read_shaped<72, false>}, {"32 VGPR + scratch", (const void *) stream_read_shaped<32, true>},
  };
  for (const Shape &sh : shapes) {
    for (int lds : {20480, 28672}) {
      CHECK(hipFuncSetAttribute(sh.fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      void *args[] = {&src, (void *) nullptr, &sink};
      size_t n_vec = read_bytes / 16;
      args[1] = &n_vec;
      CHECK(hipEventRecord(b0, sb));
      CHECK(hipLaunchKernel(sh.fn, dim3(256), dim3(256), args, lds, sb));
      CHECK(hipEventRecord(b1, sb));
      CHECK(hipDeviceSynchronize());
      float alone;
      CHECK(hipEventElapsedTime(&alone, b0, b1));
      CHECK(hipEventRecord(a0, sa)); draw(); CHECK(hipEventRecord(a1, sa));
      CHECK(hipEventRecord(b0, sb));
      CHECK(hipLaunchKernel(sh.fn, dim3(256), dim3(256), args, lds, sb));
      CHECK(hipEventRecord(b1, sb));
      CHECK(hipDeviceSynchronize());
      float d, s2, span;
      CHECK(hipEventElapsedTime(&d, a0, a1));
      CHECK(hipEventElapsedTime(&s2, b0, b1));
      CHECK(hipEventElapsedTime(&span, a0, b1));
      printf("guest %-18s lds %5d B: alone %.3f ms | together: draw %.3f ms, guest %.3f ms, span %.3f ms  -> %s\n",
             sh.name, lds, alone, d, s2, span, span < d + 0.5f * alone ? "RUNS BESIDE the draw kernel" : "waits for it");
      fflush(stdout);
    }
  }
  return 0;
}
