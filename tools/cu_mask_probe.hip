// cu_mask_probe.hip -- can the draw kernel and the scatter share the GPU SPATIALLY (CU masks on two streams)?
//
// Draw and scatter cannot share a CU, so a step is draw + scatter.  hipExtStreamCreateWithCUMask
// gives a stream a subset of the CUs.  This probe (1) finds out which bits of the mask are which XCD, (2) times a draw
// launch of 3/4 of the waves on 6 XCDs, the scatter on 2 XCDs, and both at once, against the two in a row on
// the whole GPU.
//
// build: hipcc -O2 --offload-arch=gfx950 -Iinclude tools/cu_mask_probe.hip -Lcudabrot_amd -lcudabrot_amd
//        -Wl,-rpath,$PWD/cudabrot_amd -o tools/cu_mask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cudabrot_amd.h"

#define CHECK(x)                                                                  \
  do {                                                                            \
    int e_ = (int) (x);                                                           \
    if (e_ != 0) {                                                                \
      printf("error %d at %s:%d\n", e_, __FILE__, __LINE__);                      \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__global__ void __launch_bounds__(256) where_am_i(unsigned *xcc_hist, unsigned *cu_seen) {
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xfu;  // HW_REG_XCC_ID
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));          // HW_REG_HW_ID
    const unsigned cu = (hw >> 8) & 0xfu, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
    atomicAdd(&xcc_hist[xcc], 1u);
    const unsigned id = (se * 2u + sh) * 16u + cu;  // < 256
    atomicOr(&cu_seen[xcc * 8u + (id >> 5)], 1u << (id & 31u));
  }
  // stay a while so that the workgroups spread over everything the mask allows
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < 200000ull) {
  }
}

static hipStream_t masked_stream(const unsigned *mask) {
  hipStream_t s;
  CHECK(hipExtStreamCreateWithCUMask(&s, 8, mask));
  return s;
}

static float ms_between(hipEvent_t a, hipEvent_t b) {
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

int main() {
  unsigned *d_hist, *d_seen, h_hist[16], h_seen[64];
  CHECK(hipMalloc(&d_hist, 64));
  CHECK(hipMalloc(&d_seen, 256));
  // (1) which mask bits are which XCD?
  for (int test = 0; test < 4; ++test) {
    unsigned mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const char *what = "";
    if (test == 0) { mask[0] = 0xffffffffu; what = "bits 0..31"; }
    if (test == 1) { for (int w = 0; w < 8; ++w) mask[w] = 0x01010101u; what = "every 8th bit"; }
    if (test == 2) { mask[0] = mask[1] = 0xffffffffu; what = "bits 0..63"; }
    if (test == 3) { for (int w = 0; w < 8; ++w) mask[w] = 0x03030303u; what = "bits 8k, 8k+1"; }
    hipStream_t s = masked_stream(mask);
    CHECK(hipMemsetAsync(d_hist, 0, 64, s));
    CHECK(hipMemsetAsync(d_seen, 0, 256, s));
    hipLaunchKernelGGL(where_am_i, dim3(4096), dim3(256), 0, s, d_hist, d_seen);
    CHECK(hipStreamSynchronize(s));
    CHECK(hipMemcpy(h_hist, d_hist, 64, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h_seen, d_seen, 256, hipMemcpyDeviceToHost));
    printf("mask %-14s: workgroups per XCC:", what);
    for (int x = 0; x < 8; ++x) printf(" %u", h_hist[x]);
    printf("; distinct CUs per XCC:");
    for (int x = 0; x < 8; ++x) {
      int n = 0;
      for (int w = 0; w < 8; ++w) n += __builtin_popcount(h_seen[x * 8 + w]);
      printf(" %d", n);
    }
    printf("\n");
    CHECK(hipStreamDestroy(s));
  }

  // (2) draw on 6 XCDs beside scatter on 2
  const int W = 4096, H = 4096;
  cb_fractal_dimensions dims;
  memset(&dims, 0, sizeof(dims));
  dims.w = W;
  dims.h = H;
  dims.min_real = -2.0;
  dims.max_real = 2.0;
  dims.min_imag = -2.0;
  dims.max_imag = 2.0;
  { const char *msg = nullptr; if (!cb_recompute_pixel_deltas(&dims, &msg)) { printf("dims: %s\n", msg ? msg : "?"); return 1; } }
  cb_iteration_control it = {20000, 20};
  const unsigned T = 262144, T34 = 196608, spt = 3200;
  cb_pixel *hist;
  CHECK(hipMalloc(&hist, (size_t) W * H * sizeof(cb_pixel)));
  CHECK(hipMemset(hist, 0, (size_t) W * H * sizeof(cb_pixel)));
  void *states, *states34, *carry, *carry34, *ws0, *ws1;
  CHECK(hipMalloc(&states, cb_rng_state_bytes(T)));
  CHECK(hipMalloc(&states34, cb_rng_state_bytes(T34)));
  CHECK(hipMalloc(&carry, cb_carry_bytes(T)));
  CHECK(hipMalloc(&carry34, cb_carry_bytes(T34)));
  CHECK(hipMemset(carry, 0, cb_carry_bytes(T)));
  CHECK(hipMemset(carry34, 0, cb_carry_bytes(T34)));
  const size_t ws_bytes = cb_scatter_workspace_bytes(&dims, T, spt), ws34_bytes = cb_scatter_workspace_bytes(&dims, T34, spt);
  CHECK(hipMalloc(&ws0, ws_bytes));
  CHECK(hipMalloc(&ws1, ws34_bytes));
  CHECK(cb_initialize_rng(1337, 0, T, states, nullptr));
  CHECK(cb_initialize_rng(1337, 0, T34, states34, nullptr));
  hipEvent_t e[4];
  for (auto &x : e) CHECK(hipEventCreate(&x));

  // the mapping printed above decides these: try "the first 6 XCDs / the last 2" in both plausible layouts
  for (int layout = 0; layout < 2; ++layout) {
    unsigned six[8], two[8];
    for (int w = 0; w < 8; ++w) {
      if (layout == 0) {  // the same 32-bit word for every XCC: CUs 0..23 / 24..31
        six[w] = 0x00ffffffu;
        two[w] = 0xff000000u;
      } else {            // three of every four CUs / the fourth
        six[w] = 0x77777777u;
        two[w] = 0x88888888u;
      }
    }
    hipStream_t sa = masked_stream(six), sb = masked_stream(two);
    // warm up + a full stream in ws0 for the scatter to work on
    for (int k = 0; k < 3; ++k) {
      CHECK(cb_draw_buddhabrot(&dims, hist, &it, states, T, spt, nullptr, CB_KERNEL_DEFAULT, ws0, ws_bytes, carry, nullptr));
      CHECK(hipDeviceSynchronize());
      if (k < 2) CHECK(cb_flush_scatter(&dims, hist, T, ws0, ws_bytes, nullptr));
      CHECK(cb_draw_buddhabrot(&dims, hist, &it, states34, T34, spt, nullptr, CB_KERNEL_DEFAULT, ws1, ws34_bytes, carry34, sa));
      CHECK(hipDeviceSynchronize());
      CHECK(cb_flush_scatter(&dims, hist, T34, ws1, ws34_bytes, nullptr));
      CHECK(hipDeviceSynchronize());
    }
    // whole GPU, one after the other (the scatter of ws0 must be repeatable: flush re-reads the stream, so re-draw)
    CHECK(hipEventRecord(e[0], nullptr));
    CHECK(cb_draw_buddhabrot(&dims, hist, &it, states, T, spt, nullptr, CB_KERNEL_DEFAULT, ws0, ws_bytes, carry, nullptr));
    CHECK(hipEventRecord(e[1], nullptr));
    CHECK(cb_flush_scatter(&dims, hist, T, ws0, ws_bytes, nullptr));
    CHECK(hipEventRecord(e[2], nullptr));
    CHECK(hipDeviceSynchronize());
    printf("layout %d: whole GPU: draw %.3f ms + scatter %.3f ms\n", layout, ms_between(e[0], e[1]), ms_between(e[1], e[2]));
    // a full stream again
    CHECK(cb_draw_buddhabrot(&dims, hist, &it, states, T, spt, nullptr, CB_KERNEL_DEFAULT, ws0, ws_bytes, carry, nullptr));
    CHECK(hipDeviceSynchronize());
    // scatter alone on the 2 XCDs
    CHECK(hipEventRecord(e[0], sb));
    CHECK(cb_flush_scatter(&dims, hist, T, ws0, ws_bytes, sb));
    CHECK(hipEventRecord(e[1], sb));
    CHECK(hipDeviceSynchronize());
    printf("layout %d: scatter alone on 2 XCDs: %.3f ms\n", layout, ms_between(e[0], e[1]));
    // draw of 3/4 of the waves alone on the 6 XCDs
    CHECK(hipEventRecord(e[0], sa));
    CHECK(cb_draw_buddhabrot(&dims, hist, &it, states34, T34, spt, nullptr, CB_KERNEL_DEFAULT, ws1, ws34_bytes, carry34, sa));
    CHECK(hipEventRecord(e[1], sa));
    CHECK(hipDeviceSynchronize());
    printf("layout %d: draw of 3/4 of the waves alone on 6 XCDs: %.3f ms\n", layout, ms_between(e[0], e[1]));
    CHECK(cb_flush_scatter(&dims, hist, T34, ws1, ws34_bytes, nullptr));
    // both at once
    CHECK(cb_draw_buddhabrot(&dims, hist, &it, states, T, spt, nullptr, CB_KERNEL_DEFAULT, ws0, ws_bytes, carry, nullptr));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e[0], sa));
    CHECK(hipEventRecord(e[2], sb));
    CHECK(cb_draw_buddhabrot(&dims, hist, &it, states34, T34, spt, nullptr, CB_KERNEL_DEFAULT, ws1, ws34_bytes, carry34, sa));
    CHECK(cb_flush_scatter(&dims, hist, T, ws0, ws_bytes, sb));
    CHECK(hipEventRecord(e[1], sa));
    CHECK(hipEventRecord(e[3], sb));
    CHECK(hipDeviceSynchronize());
    printf("layout %d: at once: draw (3/4, 6 XCDs) %.3f ms, scatter (2 XCDs) %.3f ms\n", layout, ms_between(e[0], e[1]),
           ms_between(e[2], e[3]));
    CHECK(cb_flush_scatter(&dims, hist, T34, ws1, ws34_bytes, nullptr));
    CHECK(hipDeviceSynchronize());
    CHECK(hipStreamDestroy(sa));
    CHECK(hipStreamDestroy(sb));
  }
  return 0;
}
