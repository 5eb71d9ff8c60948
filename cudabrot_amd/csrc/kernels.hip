// kernels.hip -- hand-written gfx950 device code for the Buddhabrot hot path.
//
// Compiled with -ffp-contract=off (see device_math.h).  wave = 64 lanes everywhere.
//
// Kernels
//   rng_init_kernel    InitializeRNG (cudabrot.cu:146-149): one XORWOW state per thread, jump-ahead
//                      by subsequence * 2^67 through GF(2) matrices held in device memory.
//   draw_simple_kernel DrawBuddhabrot (cudabrot.cu:379-414) one lane = one reference thread, in
//                      lock-step.  Validation baseline: ~2 % lane efficiency at deep max_iter
//                      (SURVEY.md H2).
//   (draw_wave_kernel, the product path, lives in draw_wave.hip.)
#include "draw_common.h"

namespace cb {

// ------------------------------------------------------------------------------------------------
// RNG init
// ------------------------------------------------------------------------------------------------

struct SeedState {
  uint32_t x[5];
  uint32_t d;
};

// x <- M x over GF(2); M in rocRAND's image layout m[(word*32+bit)*5 + k] (rocrand_xorwow.h:49-66).
// The matrix address is wave-uniform, so its rows arrive through scalar loads.
__device__ __forceinline__ void gf2_mat_vec(const uint32_t *__restrict__ m, uint32_t x[5]) {
  uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0;
#pragma unroll 1
  for (int wd = 0; wd < 5; ++wd) {
    const uint32_t xv = x[wd];
#pragma unroll 4
    for (int j = 0; j < 32; ++j) {
      const uint32_t *row = m + (wd * 32 + j) * 5;
      const uint32_t sel = 0u - ((xv >> j) & 1u);
      r0 ^= sel & row[0];
      r1 ^= sel & row[1];
      r2 ^= sel & row[2];
      r3 ^= sel & row[3];
      r4 ^= sel & row[4];
    }
  }
  x[0] = r0;
  x[1] = r1;
  x[2] = r2;
  x[3] = r3;
  x[4] = r4;
}

__global__ void __launch_bounds__(256)
rng_init_kernel(SeedState seed, unsigned long long first_subsequence, uint32_t n_threads,
                uint32_t *__restrict__ states, const uint32_t *__restrict__ matrices) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = tid < n_threads;
  const unsigned long long subseq = first_subsequence + (valid ? tid : 0u);
  uint32_t x[5] = {seed.x[0], seed.x[1], seed.x[2], seed.x[3], seed.x[4]};
  // discard_subsequence (rocrand_xorwow.h:149-158): x <- A^(subseq * 2^67) x, d unchanged.
  // Binary digits instead of rocRAND's base-4 digits: one matrix product per set bit.
  for (int b = 0; b < kSeqJumpMatrices; ++b) {
    const bool bit = valid && ((subseq >> b) & 1ull);
    if (__ballot(bit) == 0ull) {
      if (__ballot(valid && (subseq >> b) != 0ull) == 0ull) break;  // no lane has higher bits
      continue;
    }
    if (bit) gf2_mat_vec(matrices + (size_t) b * kMatrixWords, x);
  }
  if (valid) {
    states[0 * (size_t) n_threads + tid] = x[0];
    states[1 * (size_t) n_threads + tid] = x[1];
    states[2 * (size_t) n_threads + tid] = x[2];
    states[3 * (size_t) n_threads + tid] = x[3];
    states[4 * (size_t) n_threads + tid] = x[4];
    states[5 * (size_t) n_threads + tid] = seed.d;
  }
}

hipError_t launch_rng_init(uint64_t seed, uint64_t first_subsequence, uint32_t n_threads,
                           uint32_t *d_states, const uint32_t *d_matrices, hipStream_t stream) {
  if (n_threads == 0) return hipSuccess;
  SeedState s;
  seed_state(seed, s.x, &s.d);
  const uint32_t blocks = (n_threads + 255u) / 256u;
  hipLaunchKernelGGL(rng_init_kernel, dim3(blocks), dim3(256), 0, stream, s,
                     (unsigned long long) first_subsequence, n_threads, d_states, d_matrices);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Shared pieces of the draw kernels
// ------------------------------------------------------------------------------------------------

// Per-lane statistics, summed over the wave at kernel end (one atomic per counter per wave).
struct LaneStats {
  unsigned long long samples = 0, rejected = 0, never_escaped = 0, too_fast = 0, recorded = 0,
                     iterate_steps = 0, replay_steps = 0, increments = 0, reserved = 0,
                     status = 0;
};

__device__ __forceinline__ void flush_stats(cb_counters *counters, const LaneStats &s) {
  if (!counters) return;
  const unsigned long long v[10] = {
      wave_sum(s.samples),       wave_sum(s.rejected),     wave_sum(s.never_escaped),
      wave_sum(s.too_fast),      wave_sum(s.recorded),     wave_sum(s.iterate_steps),
      wave_sum(s.replay_steps),  wave_sum(s.increments),   wave_sum(s.reserved),
      wave_sum(s.status)};
  if (lane_id() == 0) {
    unsigned long long *c = reinterpret_cast<unsigned long long *>(counters);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      if (v[i]) __hip_atomic_fetch_add(c + i, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (v[9]) __hip_atomic_fetch_or(c + 9, v[9], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------------------------------------
// draw_simple_kernel: the reference's loop structure, one lane per reference thread
// ------------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) draw_simple_kernel(DrawArgs a) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = tid < a.n_threads;
  const Canvas cv = make_canvas(a);
  LaneStats st;
  if (valid) {
    Xorwow rng = load_rng(a.states, a.n_threads, tid);
    for (uint32_t sample = 0; sample < a.samples_per_thread; ++sample) {  // cudabrot.cu:390
      const double real = sample_coordinate(rng);
      const double imag = sample_coordinate(rng);
      st.samples++;
      if (!a.burning_ship && (in_main_cardioid(real, imag) || in_order2_bulb(real, imag))) {  // cudabrot.cu:397-399
        st.rejected++;
        continue;
      }
      // IterateMandelbrot, cudabrot.cu:319-340
      double r = real, i = imag;
      int k = a.max_iter;
      for (int it = 0; it < a.max_iter; ++it) {
        if ((a.burning_ship ? mandel_step_ship(real, imag, r, i) : mandel_step(real, imag, r, i)) > 4.0) {
          k = it;
          break;
        }
      }
      if (k >= a.max_iter) {  // cudabrot.cu:407
        st.never_escaped++;
        st.iterate_steps += (unsigned long long) (a.max_iter > 0 ? a.max_iter : 0);
        continue;
      }
      st.iterate_steps += (unsigned long long) k + 1ull;
      if (k < a.min_iter) {  // cudabrot.cu:408
        st.too_fast++;
        continue;
      }
      st.recorded++;
      // IterateAndRecord, cudabrot.cu:347-365; bounded so that a wave always terminates
      r = real;
      i = imag;
      for (int it = 0; it <= a.max_iter; ++it) {
        const double m = a.burning_ship ? mandel_step_ship(real, imag, r, i) : mandel_step(real, imag, r, i);
        st.replay_steps++;
        st.increments += increment_pixel_counter(r, i, a.hist, cv) ? 1ull : 0ull;
        if (m > 4.0) break;
        if (it == a.max_iter) st.status |= CB_STATUS_REPLAY_RUNAWAY;
      }
    }
    store_rng(a.states, a.n_threads, tid, rng);
  }
  flush_stats(a.counters, st);
}

hipError_t launch_draw_simple(const DrawArgs &a, hipStream_t stream) {
  if (a.n_threads == 0 || a.samples_per_thread == 0) return hipSuccess;
  const uint32_t blocks = (a.n_threads + 255u) / 256u;
  hipLaunchKernelGGL(draw_simple_kernel, dim3(blocks), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace cb
