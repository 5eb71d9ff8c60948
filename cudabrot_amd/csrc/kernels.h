// kernels.h -- internal launch interface between the C ABI (capi.hip) and the device code.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cudabrot_amd.h"

namespace cb {

// Number of GF(2) jump matrices kept on the device: A^(2^(67+b)), b in [0,64).
constexpr int kSeqJumpMatrices = 64;
constexpr int kMatrixWords = 160 * 5;

// Host: fills out[kSeqJumpMatrices][kMatrixWords] (xorwow_host.cpp).
void build_sequence_jump_matrices(uint32_t *out);
// Host: seeded XORWOW state before any jump (rocrand_xorwow.h:104-123): x[0..4], d.
void seed_state(uint64_t seed, uint32_t x[5], uint32_t *d);

struct DrawArgs {
  // canvas (cudabrot.cu:46-58) + exact-reciprocal fast path
  double min_real, min_imag, delta_real, delta_imag, inv_delta_real, inv_delta_imag;
  int w, h, pow2_real, pow2_imag;
  // iteration control (cudabrot.cu:62-67)
  int max_iter, min_iter;
  // stage split: samples enter the long-iterate stage after exactly head_steps iterations
  int head_steps;
  uint32_t n_threads;
  uint32_t samples_per_thread;
  unsigned long long *hist;
  uint32_t *states;  // six planes of n_threads
  cb_counters *counters;
};

hipError_t launch_rng_init(uint64_t seed, uint64_t first_subsequence, uint32_t n_threads,
                           uint32_t *d_states, const uint32_t *d_matrices, hipStream_t stream);
hipError_t launch_draw_simple(const DrawArgs &a, hipStream_t stream);
hipError_t launch_draw_wave(const DrawArgs &a, bool timed, hipStream_t stream);

// Steps per chunk of the long-iterate stage; head_steps is chosen so that
// (max_iter - head_steps) % kChunk == 0.
constexpr int kChunk = 16;
int choose_head_steps(int max_iter, int min_iter);

}  // namespace cb
