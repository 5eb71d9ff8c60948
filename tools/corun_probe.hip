// What does the draw launch (C3) lose to a neighbour, per instruction the neighbour issues, by KIND of instruction?
// The product's draw call (through the C-ABI, stream A) beside a synthetic kernel on stream B that fits the CUs the way
// the scatter does (one workgroup of 1024 threads per CU, 64 registers, priority 0) and issues ONE kind of instruction
// in a loop for a fixed wall time, counting its turns.  Printed per kind: the draw launch's time, the neighbour's
// instructions per SIMD-cycle... and so the draw's milliseconds lost per 10^9 neighbour wave-instructions.
//   build: hipcc --offload-arch=gfx950 -O3 -I include -o tools/build/corun_probe tools/corun_probe.hip -L cudabrot_amd -lcudabrot_amd
//   run  : LD_LIBRARY_PATH=cudabrot_amd tools/build/corun_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "cudabrot_amd.h"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define CB(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s: %d\n", #x, rc_); exit(1); } } while (0)

enum Kind { kNone = 0, kValu, kValuDep, kSalu, kLdsAdd, kLdsAddConflict, kLdsRtn, kVmemLoad, kBranch, kSleep, kVmemL2, kVmemL1, kVmemStore, kVmemLoadSlow, kF64Full, kF64Quarter, kF64Zero, kKinds };
static const char *kNames[kKinds] = {"none", "valu (4 chains)", "valu (1 chain)", "salu", "ds_add (no conflicts)",
                                     "ds_add (random banks)", "ds_add_rtn + wait", "16-byte loads + wait", "taken branches", "s_sleep only",
                                     "... from 4 MiB (L2 hits)", "... from 16 KiB (L1 hits)", "16-byte stores (1 GiB)", "16-byte loads, 1 in flight",
                                     "fp64 fma, 4 chains, 64 lanes", "... 16 of 64 lanes (EXEC)", "... 64 lanes, operands 0"};

// One turn = kBody instructions of the kind.  duty: after each turn the wave sleeps `sleep` x 64 cycles.
constexpr int kBody = 64;
template <int KIND>
__global__ void __launch_bounds__(1024, 8) neighbour_kernel(unsigned long long ticks, unsigned long long *turns_out,
                                                            const uint4 *src, uint32_t src_mask, uint32_t sleep) {
  extern __shared__ uint32_t lds[];
  __builtin_amdgcn_s_setprio(0);
  const unsigned long long t0 = wall_clock64();
  unsigned long long turns = 0;
  float a = threadIdx.x, b = 1.5f, c = 2.5f, d = 3.5f;
  uint32_t s = blockIdx.x;
  const uint32_t lane_addr = (threadIdx.x & 1023u) * 4u;                          // conflict-free
  uint32_t rnd = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  uint4 acc = make_uint4(0, 0, 0, 0);
  uint32_t at = (blockIdx.x * 1024u + threadIdx.x) & src_mask;
  for (uint32_t guard = 0; guard < (1u << 22); ++guard) {
    if ((guard & 7u) == 0u && wall_clock64() - t0 >= ticks) break;
    if (KIND == kValu) {
#pragma unroll
      for (int k = 0; k < kBody / 4; ++k) {
        asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
      }
    } else if (KIND == kF64Full || KIND == kF64Quarter || KIND == kF64Zero) {
      // fp64 fma chains with all lanes, with a quarter of them (EXEC), and with all lanes on operands that never change
      double da = KIND == kF64Zero ? 0.0 : (double) a, db = KIND == kF64Zero ? 0.0 : 1.25, dc = KIND == kF64Zero ? 0.0 : 2.5, dd = KIND == kF64Zero ? 0.0 : 3.75;
      if (KIND != kF64Quarter || (threadIdx.x & 63u) < 16u) {
#pragma unroll
        for (int k = 0; k < kBody / 4; ++k) {
          asm volatile("v_fma_f64 %0, %0, %0, %0\n\tv_fma_f64 %1, %1, %1, %1\n\tv_fma_f64 %2, %2, %2, %2\n\tv_fma_f64 %3, %3, %3, %3"
                       : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd));
        }
      }
      a += (float) (da + db + dc + dd);
    } else if (KIND == kValuDep) {
#pragma unroll
      for (int k = 0; k < kBody; ++k) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a));
    } else if (KIND == kSalu) {
#pragma unroll
      for (int k = 0; k < kBody; ++k) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) : : "scc");
    } else if (KIND == kLdsAdd || KIND == kLdsAddConflict) {
#pragma unroll
      for (int k = 0; k < kBody; ++k) {
        uint32_t addr = lane_addr;
        if (KIND == kLdsAddConflict) {
          rnd = rnd * 1664525u + 1013904223u;
          addr = (rnd >> 16) & 0xfffcu;
        }
        asm volatile("ds_add_u32 %0, %1" : : "v"(addr), "v"(1u) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (KIND == kLdsRtn) {
#pragma unroll
      for (int k = 0; k < kBody / 8; ++k) {
        uint32_t r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(r[j]) : "v"(lane_addr), "v"(1u) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) acc.x ^= r[j];
      }
    } else if (KIND == kVmemStore) {
#pragma unroll
      for (int k = 0; k < kBody / 8; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const_cast<uint4 *>(src)[at] = acc;
          at = (at + 1024u * 256u) & src_mask;
        }
      }
    } else if (KIND == kVmemLoadSlow) {
#pragma unroll
      for (int k = 0; k < kBody / 8; ++k) {
        const uint4 v = src[at];
        at = (at + 1024u * 256u + (v.x & 1u)) & src_mask;  // (the next address waits for this load)
        acc.x ^= v.w;
      }
    } else if (KIND == kVmemLoad || KIND == kVmemL2 || KIND == kVmemL1) {
#pragma unroll
      for (int k = 0; k < kBody / 8; ++k) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[j] = src[at];
          at = (at + 1024u * 256u) & (KIND == kVmemL2 ? (1u << 18) - 1u : KIND == kVmemL1 ? 1023u : src_mask);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc.x ^= v[j].x ^ v[j].w;
      }
    } else if (KIND == kBranch) {
#pragma unroll
      for (int k = 0; k < kBody / 2; ++k) asm volatile("s_cmp_eq_u32 %0, %0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n\t1:" : : "s"(s) : "scc");
    } else if (KIND == kSleep) {
      __builtin_amdgcn_s_sleep(64);
    }
    if (sleep) __builtin_amdgcn_s_sleep(127);
    ++turns;
  }
  if (a + b + c + d == 12345.0f || s == 0xdeadbeefu || acc.x == 0x12345u) turns_out[1] = 1;
  if ((threadIdx.x & 63u) == 0u) atomicAdd(turns_out, turns);
}

// One wave that sleeps beside the two and reads both clocks: the shader clock's frequency while they run.
__global__ void __launch_bounds__(64) clock_sampler_kernel(unsigned long long ticks, unsigned long long *out) {
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  for (uint32_t turn = 0; turn < (1u << 22) && wall_clock64() - w0 < ticks; ++turn) __builtin_amdgcn_s_sleep(64);
  if (threadIdx.x == 0) {
    out[0] = wall_clock64() - w0;  // 100 MHz
    out[1] = clock64() - c0;       // shader clock
  }
}

int main(int argc, char **argv) {
  const uint32_t sleep = argc > 1 ? (uint32_t) atoi(argv[1]) : 0u;
  cb_fractal_dimensions dims;
  memset(&dims, 0, sizeof(dims));
  dims.w = 4096; dims.h = 4096; dims.min_real = -2.0; dims.max_real = 2.0; dims.min_imag = -2.0; dims.max_imag = 2.0;
  const char *msg = nullptr;
  if (cb_recompute_pixel_deltas(&dims, &msg) != 1) { fprintf(stderr, "canvas: %s\n", msg ? msg : "?"); return 1; }
  cb_iteration_control it = {20000, 20};
  const uint32_t threads = 262144, spt = 50 * 64;
  hipStream_t sa, sb, sc;
  CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
  void *states, *ws, *carry, *src;
  cb_pixel *hist;
  cb_counters *counters;
  unsigned long long *turns;
  const size_t ws_bytes = cb_scatter_workspace_bytes(&dims, threads, spt);
  CHECK(hipMalloc(&states, cb_rng_state_bytes(threads)));
  CHECK(hipMalloc(&ws, ws_bytes));
  CHECK(hipMalloc(&carry, cb_carry_bytes(threads)));
  CHECK(hipMemset(carry, 0, cb_carry_bytes(threads)));
  CHECK(hipMalloc(&hist, sizeof(cb_pixel) * 4096ull * 4096ull));
  CHECK(hipMemset(hist, 0, sizeof(cb_pixel) * 4096ull * 4096ull));
  CHECK(hipMalloc(&counters, sizeof(cb_counters)));
  CHECK(hipMemset(counters, 0, sizeof(cb_counters)));
  CHECK(hipMalloc(&turns, 16));
  unsigned long long *clocks;
  CHECK(hipMalloc(&clocks, 16));
  const uint32_t src_mask = (1u << 26) - 1u;  // 2^26 x 16 B = 1 GiB
  CHECK(hipMalloc(&src, ((size_t) src_mask + 1) * 16));
  CHECK(hipMemset(src, 1, ((size_t) src_mask + 1) * 16));
  CB(cb_initialize_rng(1337, 0, threads, states, sa));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const auto draw = [&]() {
    CB(cb_draw_buddhabrot(&dims, hist, &it, states, threads, spt, counters, 0, ws, ws_bytes, carry, sa));
  };
  const bool no_flush = getenv("CORUN_NO_FLUSH") != nullptr;  // (a draw kernel built to write a stream the sort cannot read)
  const auto flush = [&]() { if (!no_flush) CB(cb_flush_scatter(&dims, hist, threads, ws, ws_bytes, sa)); };
  for (int k = 0; k < 3; ++k) { draw(); flush(); }
  CHECK(hipDeviceSynchronize());
  // the neighbour runs as long as the draw launch did beside it the time before (first: as long as the draw alone)
  printf("neighbour: one workgroup of 1024 threads per CU (256 workgroups), %s; draw: C3, 64 passes\n",
         sleep ? "sleeping 127 x 64 cycles after every 64 instructions" : "no pauses");
  double alone = 0;
  for (int kind = 0; kind < kKinds; ++kind) {
    double ms_sum = 0, rate_sum = 0, ghz_sum = 0;
    const int reps = 3;
    double life_ms = alone > 0 ? alone : 6.0;
    for (int rep = -2; rep < reps; ++rep) {  // (two settling turns: the neighbour's life follows the draw's time)
      const unsigned long long ticks = (unsigned long long) (life_ms * 1e5);
      CHECK(hipMemset(turns, 0, 16));
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0, sa));
      draw();
      CHECK(hipEventRecord(e1, sa));
      // (the draw launch first: its waves take the low registers of every SIMD, as in the product's pipeline)
      for (volatile int spin = 0; spin < 300000; ++spin) {}
      hipLaunchKernelGGL(clock_sampler_kernel, dim3(1), dim3(64), 0, sc, (unsigned long long) (life_ms * 0.8e5), clocks);
      const size_t lds_bytes = 65536;
#define LAUNCH(K) hipLaunchKernelGGL(neighbour_kernel<K>, dim3(256), dim3(1024), lds_bytes, sb, ticks, turns, (const uint4 *) src, src_mask, sleep)
      switch (kind) {
        case kNone: break;
        case kValu: LAUNCH(kValu); break;
        case kValuDep: LAUNCH(kValuDep); break;
        case kSalu: LAUNCH(kSalu); break;
        case kLdsAdd: LAUNCH(kLdsAdd); break;
        case kLdsAddConflict: LAUNCH(kLdsAddConflict); break;
        case kLdsRtn: LAUNCH(kLdsRtn); break;
        case kVmemLoad: LAUNCH(kVmemLoad); break;
        case kBranch: LAUNCH(kBranch); break;
        case kSleep: LAUNCH(kSleep); break;
        case kVmemL2: LAUNCH(kVmemL2); break;
        case kVmemL1: LAUNCH(kVmemL1); break;
        case kVmemStore: LAUNCH(kVmemStore); break;
        case kVmemLoadSlow: LAUNCH(kVmemLoadSlow); break;
        case kF64Full: LAUNCH(kF64Full); break;
        case kF64Quarter: LAUNCH(kF64Quarter); break;
        case kF64Zero: LAUNCH(kF64Zero); break;
      }
      CHECK(hipDeviceSynchronize());
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long h[2];
      CHECK(hipMemcpy(h, turns, 16, hipMemcpyDeviceToHost));
      flush();
      CHECK(hipDeviceSynchronize());
      unsigned long long ck[2];
      CHECK(hipMemcpy(ck, clocks, 16, hipMemcpyDeviceToHost));
      life_ms = ms;
      if (rep < 0) continue;
      ghz_sum += ck[0] ? (double) ck[1] / (double) ck[0] * 0.1 : 0.0;
      ms_sum += ms;
      rate_sum += (double) h[0] * (kind == kVmemLoadSlow ? kBody / 8 : kBody) / ((double) ticks / 1e5);  // wave-instructions per ms, all waves together
    }
    const double ms = ms_sum / reps, rate = rate_sum / reps;
    if (kind == kNone) alone = ms;
    // wave-instructions the neighbour issued WHILE the draw ran, and what each 10^9 of them cost the draw
    const double beside = rate * ms;
    printf("%-28s draw %.3f ms (+%.3f)  shader clock %.2f GHz  neighbour %6.1f M wave-instructions per ms  draw's loss per 10^9 of them: %.2f ms\n",
           kNames[kind], ms, ms - alone, ghz_sum / reps, rate / 1e6, beside > 0 ? (ms - alone) / (beside / 1e9) : 0.0);
  }
  return 0;
}
