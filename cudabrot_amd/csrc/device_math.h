// device_math.h -- the canonical arithmetic of the hot path, shared by every kernel.
//
// Bit-exactness contract: this file is compiled with
// -ffp-contract=off; every fused operation is an explicit __builtin_fma; iterate and replay use the
// SAME mandel_step, so a replayed orbit retraces the tested orbit bit for bit (the hazard the
// reference warns of at cudabrot.cu:342-346).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cb {

// XORWOW generator state (rocRAND 4.2 rocrand_xorwow.h:72-89, the live part): 160 xorshift bits and
// the Weyl value.  Kept in registers for the whole launch; the reference reloads it from global
// memory for every sample (cudabrot.cu:382,392-393).
struct Xorwow {
  uint32_t x0, x1, x2, x3, x4, d;
};

// rocrand(&state): rocrand_xorwow.h:165-177.
__device__ __forceinline__ uint32_t xorwow_next(Xorwow &s) {
  const uint32_t t = s.x0 ^ (s.x0 >> 2);
  s.x0 = s.x1;
  s.x1 = s.x2;
  s.x2 = s.x3;
  s.x3 = s.x4;
  s.x4 = (s.x4 ^ (s.x4 << 4)) ^ (t ^ (t << 1));
  s.d += 362437u;
  return s.d + s.x4;
}

// One coordinate of a starting point: curand_uniform_double(rng) * 4.0 - 2.0 (cudabrot.cu:392-393)
// with rocRAND's mapping U = 2^-53 + v * 2^-53, v = x1 | ((x2 >> 11) << 32)
// (rocrand_uniform.h:102-109).  U = (v + 1) * 2^-53 and U*4 - 2 = (v + 1 - 2^52) * 2^-51 are both
// exactly representable, so every formulation that is exact in each step gives the same bits; this
// one needs two conversions and two FMAs (hi*2^32 + lo < 2^53 is exact; 2^-51 - 2 is representable
// and the final sum is exact).
__device__ __forceinline__ double sample_coordinate(Xorwow &s) {
  const uint32_t v1 = xorwow_next(s);
  const uint32_t v2 = xorwow_next(s);
  const double lo = (double) v1;
  const double hi = (double) (v2 >> 11);
  const double v = __builtin_fma(hi, 4294967296.0, lo);
  return __builtin_fma(v, 0x1p-51, 0x1p-51 - 2.0);
}

// InMainCardioid, cudabrot.cu:284-290 (q*q + imag_squared is one FMA in the canonical sequence).
__device__ __forceinline__ bool in_main_cardioid(double real, double imag) {
  const double imag_squared = imag * imag;
  double q = real - 0.25;
  q = __builtin_fma(q, q, imag_squared);
  return (q * (q + (real - 0.25))) < (imag_squared * 0.25);
}

// InOrder2Bulb, cudabrot.cu:294-298.  In the canonical sequence (what hipcc makes of the reference on
// gfx950, and x86 clang likewise) the rounded imag*imag of the cardioid test is reused and the OTHER
// product is fused: fma(tmp, tmp, imag*imag).
__device__ __forceinline__ bool in_order2_bulb(double real, double imag) {
  const double imag_squared = imag * imag;
  const double tmp = real + 1.0;
  return __builtin_fma(tmp, tmp, imag_squared) < (1.0 / 16.0);
}

// One z <- z^2 + c step (cudabrot.cu:331-333 and :357-359); returns |z|^2 as tested at :336/:363.
//   ii = i*i; t = fma(r,r,-ii); nr = cr + t; ni = fma(r+r, i, ci); m = fma(ni,ni, nr*nr)
// 7 fp64 instructions (3 FMA, 2 MUL, 2 ADD) = the 10 algorithmic flops of SURVEY.md section 8(d).
__device__ __forceinline__ double mandel_step(double cr, double ci, double &r, double &i) {
  const double ii = i * i;
  const double t = __builtin_fma(r, r, -ii);
  const double nr = cr + t;
  const double ni = __builtin_fma(r + r, i, ci);
  r = nr;
  i = ni;
  return __builtin_fma(ni, ni, nr * nr);
}

// ---- the same arithmetic on DOUBLED coordinates (draw_wave_kernel) ----------------------------------
//
// Scaling by a power of two commutes with IEEE rounding (no overflow; no result below 2^-1022, which
// would take |z| < 2^-511), so with C = 2c, Z = 2z every rounded intermediate of the canonical
// sequence has an exact image:
//   II = I*I        = 4 (i*i)                    T  = fma(R,R,-II)  = 4 t
//   NI = fma(R,I,CI) = 2 fma(r+r,i,ci)           NR = fma(T,0.5,CR) = 2 (cr + t)
//   M  = fma(NI,NI,NR*NR) = 4 m,   tested against 16
// The doubling r+r of the cross term comes for free: SIX fp64 instructions per step instead of seven,
// the same bits (halved) in every register.  Cardioid and bulb tests scale the same way.
__device__ __forceinline__ double sample_coordinate2(Xorwow &s) {  // 2 * sample_coordinate
  const uint32_t v1 = xorwow_next(s);
  const uint32_t v2 = xorwow_next(s);
  const double lo = (double) v1;
  const double hi = (double) (v2 >> 11);
  const double v = __builtin_fma(hi, 4294967296.0, lo);
  return __builtin_fma(v, 0x1p-50, 0x1p-50 - 4.0);
}

// in_main_cardioid(R/2, I/2): II = 4 im^2, X = 2 (re - 1/4), Q = 4 q, S = 4 (q + (re - 1/4)),
// Q*S = 16 q (q + ...), and 16 * (im^2 / 4) = II.
__device__ __forceinline__ bool in_main_cardioid2(double R, double I) {
  const double II = I * I;
  const double X = R - 0.5;
  const double Q = __builtin_fma(X, X, II);
  const double S = __builtin_fma(X, 2.0, Q);
  return (Q * S) < II;
}

// in_order2_bulb(R/2, I/2): fma(T,T,II) = 4 fma(tmp,tmp,im^2), against 4/16.
__device__ __forceinline__ bool in_order2_bulb2(double R, double I) {
  const double II = I * I;
  const double T = R + 2.0;
  return __builtin_fma(T, T, II) < 0.25;
}

// mandel_step on doubled coordinates; returns 4 |z|^2 (escape: > 16).
__device__ __forceinline__ double mandel_step2(double CR, double CI, double &R, double &I) {
  const double II = I * I;
  const double T = __builtin_fma(R, R, -II);
  const double NI = __builtin_fma(R, I, CI);
  const double NR = __builtin_fma(T, 0.5, CR);
  R = NR;
  I = NI;
  return __builtin_fma(NI, NI, NR * NR);
}

// RENDER_BURNING_SHIP (cudabrot.cu:15-17,327-330,353-356): real and imag are replaced by their
// magnitudes before the step.  The squares do not notice; the cross term becomes 2|r||i|, which hipcc
// (gfx950) forms as fma(|i|, |r|+|r|, ci).
__device__ __forceinline__ double mandel_step_ship(double cr, double ci, double &r, double &i) {
  const double ii = i * i;
  const double t = __builtin_fma(r, r, -ii);
  const double nr = cr + t;
  const double ni = __builtin_fma(__builtin_fabs(r) + __builtin_fabs(r), __builtin_fabs(i), ci);
  r = nr;
  i = ni;
  return __builtin_fma(ni, ni, nr * nr);
}
__device__ __forceinline__ double mandel_step2_ship(double CR, double CI, double &R, double &I) {
  const double II = I * I;
  const double T = __builtin_fma(R, R, -II);
  const double NI = __builtin_fma(__builtin_fabs(R), __builtin_fabs(I), CI);
  const double NR = __builtin_fma(T, 0.5, CR);
  R = NR;
  I = NI;
  return __builtin_fma(NI, NI, NR * NR);
}

// Canvas geometry as the kernels consume it: FractalDimensions (cudabrot.cu:46-58) plus the exact
// reciprocal fast path of SURVEY.md H3.
struct Canvas {
  double min_real, min_imag;
  double delta_real, delta_imag;
  double inv_delta_real, inv_delta_imag;  // valid iff pow2_real / pow2_imag
  int w, h;
  int pow2_real, pow2_imag;  // delta is a power of two: x / delta == x * (1/delta) bit for bit
  double rcp_delta_real, rcp_delta_imag;  // RN(1 / delta), an estimate only (DrawArgs)
};

// IncrementPixelCounter, cudabrot.cu:302-314, with the += made a device-scope atomic (the
// reference's plain += loses updates under races) and 64-bit indexing.  Returns
// true if a counter was incremented.
// The pixel a point falls on: true and (row, col) if it is on the canvas.
__device__ __forceinline__ bool pixel_of(double real, double imag, const Canvas &c, int &row,
                                         int &col) {
  if ((real < c.min_real) || (imag < c.min_imag)) return false;
  const double fx = real - c.min_real;
  const double fy = imag - c.min_imag;
  // (int) of a double: v_cvt_i32_f64 saturates where x86 yields INT_MIN; both fail the bounds test.
  col = c.pow2_real ? (int) (fx * c.inv_delta_real) : (int) (fx / c.delta_real);
  row = c.pow2_imag ? (int) (fy * c.inv_delta_imag) : (int) (fy / c.delta_imag);
  return (row >= 0) && (row < c.h) && (col >= 0) && (col < c.w);
}

__device__ __forceinline__ void add_to_pixel(unsigned long long *data, const Canvas &c, int row,
                                             int col, unsigned long long n) {
  unsigned long long *p =
      data + ((unsigned long long) row * (unsigned long long) c.w + (unsigned long long) col);
  // result unused -> no-return global_atomic_add_x2, agent scope
  __hip_atomic_fetch_add(p, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool increment_pixel_counter(double real, double imag,
                                                        unsigned long long *data, const Canvas &c) {
  int row, col;
  if (!pixel_of(real, imag, c, row, col)) return false;
  add_to_pixel(data, c, row, col, 1ull);
  return true;
}

// Lane index inside the wave and prefix population count of a 64-bit lane mask.
__device__ __forceinline__ int lane_id() { return (int) (threadIdx.x & 63u); }
__device__ __forceinline__ int mask_prefix(unsigned long long mask) {
  return (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32),
                                         __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace cb
