// draw_common.h -- pieces shared by the draw kernels (kernels.hip, draw_wave.hip).
#pragma once

#include "device_math.h"
#include "kernels.h"

namespace cb {

__device__ __forceinline__ Canvas make_canvas(const DrawArgs &a) {
  Canvas c;
  c.min_real = a.min_real;
  c.min_imag = a.min_imag;
  c.delta_real = a.delta_real;
  c.delta_imag = a.delta_imag;
  c.inv_delta_real = a.inv_delta_real;
  c.inv_delta_imag = a.inv_delta_imag;
  c.w = a.w;
  c.h = a.h;
  c.pow2_real = a.pow2_real;
  c.pow2_imag = a.pow2_imag;
  c.rcp_delta_real = a.rcp_delta_real;
  c.rcp_delta_imag = a.rcp_delta_imag;
  return c;
}

// Generator states live in six planes of n words (x0..x4, d): coalesced loads and stores.
__device__ __forceinline__ Xorwow load_rng(const uint32_t *states, uint32_t n, uint32_t tid) {
  Xorwow s;
  s.x0 = states[0 * (size_t) n + tid];
  s.x1 = states[1 * (size_t) n + tid];
  s.x2 = states[2 * (size_t) n + tid];
  s.x3 = states[3 * (size_t) n + tid];
  s.x4 = states[4 * (size_t) n + tid];
  s.d = states[5 * (size_t) n + tid];
  return s;
}

__device__ __forceinline__ void store_rng(uint32_t *states, uint32_t n, uint32_t tid,
                                          const Xorwow &s) {
  states[0 * (size_t) n + tid] = s.x0;
  states[1 * (size_t) n + tid] = s.x1;
  states[2 * (size_t) n + tid] = s.x2;
  states[3 * (size_t) n + tid] = s.x3;
  states[4 * (size_t) n + tid] = s.x4;
  states[5 * (size_t) n + tid] = s.d;
}

}  // namespace cb
