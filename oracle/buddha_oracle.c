/*
 * buddha_oracle.c -- CPU restatement of the cudabrot hot path.  TEST INFRASTRUCTURE ONLY
 * (see buddha_oracle.h for who may use it and how its parity is pinned).
 *
 * Build: gcc -O2 -ffp-contract=off -mfma -fopenmp (oracle/Makefile).  -ffp-contract=off is part of
 * the contract: every fused operation below is an explicit __builtin_fma, nothing else may fuse.
 *
 * Canonical fp64 sequence -- what ROCm clang's default
 * contraction makes of cudabrot.cu:331-336 on gfx950, i.e. what the reference's own HIP build
 * computes in both its iterate and its replay loop:
 *     ii = i*i;  t = fma(r,r,-ii);  nr = cr + t;  ni = fma(r+r, i, ci);
 *     m  = fma(ni,ni, nr*nr);  escaped iff m > 4
 */
#include "buddha_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * XORWOW (rocRAND 4.2).  The xorshift part is linear over GF(2) on 160 bits; a jump by n steps
 * is multiplication by A^n.  rocRAND ships precomputed tables; this restatement derives them
 * from the step function by repeated squaring and tests/ checks them against rocRAND's tables.
 * Table layout (rocrand_xorwow.h:49-66): m[(i*32+j)*5+k] = word k of the image of bit j of x[i].
 * ------------------------------------------------------------------------------------------ */

#define NB 160

typedef struct { uint32_t img[NB][5]; } gf2mat;

/* rocrand_xorwow.h:165-177 without the Weyl part */
static void xorshift_step(uint32_t x[5]) {
  const uint32_t t = x[0] ^ (x[0] >> 2);
  x[0] = x[1];
  x[1] = x[2];
  x[2] = x[3];
  x[3] = x[4];
  x[4] = (x[4] ^ (x[4] << 4)) ^ (t ^ (t << 1));
}

/* rocrand_xorwow.h:49-66 mul_mat_vec_inplace */
static void mat_vec(const gf2mat *m, uint32_t v[5]) {
  uint32_t r[5] = {0, 0, 0, 0, 0};
  for (int b = 0; b < NB; b++) {
    if (v[b >> 5] & (1u << (b & 31))) {
      for (int k = 0; k < 5; k++) r[k] ^= m->img[b][k];
    }
  }
  memcpy(v, r, sizeof(r));
}

/* out = a o a */
static void mat_square(gf2mat *out, const gf2mat *a) {
  for (int b = 0; b < NB; b++) {
    uint32_t v[5];
    memcpy(v, a->img[b], sizeof(v));
    mat_vec(a, v);
    memcpy(out->img[b], v, sizeof(v));
  }
}

/* pow2[e] = A^(2^e), e in [0,131): offset jumps use e<64, subsequence jumps use 67+e. */
#define NPOW 131
static gf2mat *g_pow2 = NULL;

static void build_tables(void) {
#pragma omp critical(orc_tables)
  {
    if (!g_pow2) {
      gf2mat *p = (gf2mat *) malloc(sizeof(gf2mat) * NPOW);
      if (!p) abort();
      for (int b = 0; b < NB; b++) {
        uint32_t v[5] = {0, 0, 0, 0, 0};
        v[b >> 5] = 1u << (b & 31);
        xorshift_step(v);
        memcpy(p[0].img[b], v, sizeof(v));
      }
      for (int e = 1; e < NPOW; e++) mat_square(&p[e], &p[e - 1]);
      g_pow2 = p;
    }
  }
}

void orc_xorwow_jump_matrix(int i, uint32_t out[800]) {
  build_tables();
  memcpy(out, g_pow2[2 * i].img, 800 * sizeof(uint32_t));
}

void orc_xorwow_sequence_jump_matrix(int i, uint32_t out[800]) {
  build_tables();
  /* 67 + 2*i <= 129 for i < 32 */
  memcpy(out, g_pow2[67 + 2 * i].img, 800 * sizeof(uint32_t));
}

/* rocrand_xorwow.h:100-131 */
void orc_xorwow_init(uint64_t seed, uint64_t subsequence, uint64_t offset, orc_xorwow *st) {
  build_tables();
  st->x[0] = 123456789u;
  st->x[1] = 362436069u;
  st->x[2] = 521288629u;
  st->x[3] = 88675123u;
  st->x[4] = 5783321u;
  st->d = 6615241u;
  const uint32_t s0 = ((uint32_t) seed) ^ 0x2c7f967fu;
  const uint32_t s1 = ((uint32_t) (seed >> 32)) ^ 0xa03697cbu;
  const uint32_t t0 = 1228688033u * s0;
  const uint32_t t1 = 2073658381u * s1;
  st->x[0] += t0;
  st->x[1] ^= t0;
  st->x[2] += t1;
  st->x[3] ^= t1;
  st->x[4] += t0;
  st->d += t1 + t0;
  /* discard_subsequence (:149-158): x <- A^(subsequence * 2^67) x; d unchanged */
  for (int b = 0; b < 64; b++) {
    if ((subsequence >> b) & 1) mat_vec(&g_pow2[67 + b], st->x);
  }
  /* discard (:134-145): x <- A^offset x; d += (u32) offset * 362437 */
  for (int b = 0; b < 64; b++) {
    if ((offset >> b) & 1) mat_vec(&g_pow2[b], st->x);
  }
  st->d += ((uint32_t) offset) * 362437u;
}

void orc_xorwow_init_range(uint64_t seed, uint64_t first, uint64_t n, orc_xorwow *st) {
  if (n == 0) return;
  orc_xorwow_init(seed, first, 0, &st[0]);
  for (uint64_t t = 1; t < n; t++) {
    st[t] = st[t - 1];
    mat_vec(&g_pow2[67], st[t].x);
  }
}

/* rocrand_xorwow.h:165-177 */
uint32_t orc_xorwow_next(orc_xorwow *st) {
  xorshift_step(st->x);
  st->d += 362437u;
  return st->d + st->x[4];
}

/* rocrand_uniform.h:102-109 (two u32 -> (0,1]) and :454-460 */
double orc_uniform_double(orc_xorwow *st) {
  const uint32_t v1 = orc_xorwow_next(st);
  const uint32_t v2 = orc_xorwow_next(st);
  const uint64_t v = ((uint64_t) v1) | (((uint64_t) (v2 >> 11)) << 32);
  /* ROCRAND_2POW53_INV_DOUBLE = 1.1102230246251565e-16 = 2^-53 exactly; both operations exact */
  return 1.1102230246251565e-16 + ((double) v * 1.1102230246251565e-16);
}

/* ------------------------------------------------------------------------------------------
 * Fractal arithmetic
 * ------------------------------------------------------------------------------------------ */

/* cudabrot.cu:505-527 */
int orc_recompute_pixel_deltas(orc_dims *d) {
  if (d->w <= 0) return 0;
  if (d->h <= 0) return 0;
  if (d->max_real <= d->min_real) return 0;
  if (d->max_imag <= d->min_imag) return 0;
  d->delta_imag = (d->max_imag - d->min_imag) / ((double) d->h);
  d->delta_real = (d->max_real - d->min_real) / ((double) d->w);
  return 1;
}

/* cudabrot.cu:284-290; `q*q + imag_squared` contracts to one fma on gfx950 */
int orc_in_main_cardioid(double real, double imag) {
  const double imag_squared = imag * imag;
  double q = real - 0.25;
  q = __builtin_fma(q, q, imag_squared);
  return (q * (q + (real - 0.25))) < (imag_squared * 0.25);
}

/* cudabrot.cu:294-298.  hipcc (gfx950) and x86 clang both reuse the rounded imag*imag of the cardioid
 * test and fuse the OTHER product: fma(tmp, tmp, imag*imag). */
int orc_in_order2_bulb(double real, double imag) {
  const double imag_squared = imag * imag;
  const double tmp = real + 1;
  return __builtin_fma(tmp, tmp, imag_squared) < (1.0 / 16.0);
}

/* RENDER_BURNING_SHIP (cudabrot.cu:15-17): a compile-time switch in the reference, a run-time one
 * here.  With it, real and imag are replaced by their magnitudes before each step (:327-330,
 * :353-356) -- only the cross term notices -- and the cardioid / bulb shortcut is skipped (:397-399). */
static int g_burning_ship = 0;
void orc_set_burning_ship(int on) { g_burning_ship = on ? 1 : 0; }
int orc_get_burning_ship(void) { return g_burning_ship; }

/* One z <- z^2 + c step, cudabrot.cu:331-333 (= :357-359); returns |z|^2 as of :336 (= :363). */
static inline double mandel_step(double cr, double ci, double *r, double *i) {
  const double ii = (*i) * (*i);
  const double t = __builtin_fma(*r, *r, -ii);
  const double nr = cr + t;
  /* ship: hipcc (gfx950) and x86 clang both form |r|+|r| and fuse its product with |i| */
  const double ni = g_burning_ship
                        ? __builtin_fma(__builtin_fabs(*r) + __builtin_fabs(*r), __builtin_fabs(*i), ci)
                        : __builtin_fma((*r) + (*r), *i, ci);
  *r = nr;
  *i = ni;
  return __builtin_fma(ni, ni, nr * nr);
}

/* cudabrot.cu:319-340 */
int orc_iterate_mandelbrot(double start_real, double start_imag, int max_iterations) {
  double real = start_real, imag = start_imag;
  for (int i = 0; i < max_iterations; i++) {
    if (mandel_step(start_real, start_imag, &real, &imag) > 4) return i;
  }
  return max_iterations;
}

/* cudabrot.cu:302-314; returns 1 if a counter was incremented */
static inline int increment_pixel_counter(double real, double imag, uint64_t *data,
                                          const orc_dims *d, int atomic) {
  if ((real < d->min_real) || (imag < d->min_imag)) return 0;
  const int col = (int) ((real - d->min_real) / d->delta_real);
  const int row = (int) ((imag - d->min_imag) / d->delta_imag);
  if ((row >= 0) && (row < d->h) && (col >= 0) && (col < d->w)) {
    /* the reference indexes with int32 row*w+col (:312); widened so that w*h >= 2^31 is defined */
    uint64_t *p = data + ((uint64_t) row * (uint64_t) d->w + (uint64_t) col);
    if (atomic) {
      __atomic_fetch_add(p, 1, __ATOMIC_RELAXED);
    } else {
      *p += 1;
    }
    return 1;
  }
  return 0;
}

/* cudabrot.cu:347-365 */
static inline void iterate_and_record(double start_real, double start_imag, uint64_t *data,
                                      const orc_dims *d, int atomic, orc_counters *c) {
  double real = start_real, imag = start_imag;
  while (1) {
    const double m = mandel_step(start_real, start_imag, &real, &imag);
    c->replay_steps++;
    c->increments += (uint64_t) increment_pixel_counter(real, imag, data, d, atomic);
    if (m > 4) break;
  }
}

/* cudabrot.cu:347-365 for a caller-given starting point (the caller guarantees that it escapes);
 * returns the iterations executed and adds the in-canvas increments to *increments if not NULL. */
uint64_t orc_iterate_and_record(const orc_dims *dims, uint64_t *hist, double start_real, double start_imag,
                                uint64_t *increments) {
  orc_counters c;
  memset(&c, 0, sizeof(c));
  iterate_and_record(start_real, start_imag, hist, dims, 0, &c);
  if (increments) *increments += c.increments;
  return c.replay_steps;
}

/* One "thread" of DrawBuddhabrot, cudabrot.cu:381-413 */
static void draw_thread(const orc_dims *dims, uint64_t *hist, const orc_iters *it,
                        orc_xorwow *rng, int samples_per_thread, int atomic, orc_counters *c) {
  const int max_iterations = it->max_escape_iterations;
  const int min_iterations = it->min_escape_iterations;
  for (int sample = 0; sample < samples_per_thread; sample++) {
    const double real = (orc_uniform_double(rng) * 4.0) - 2.0;
    const double imag = (orc_uniform_double(rng) * 4.0) - 2.0;
    c->samples++;
    if (!g_burning_ship && (orc_in_main_cardioid(real, imag) || orc_in_order2_bulb(real, imag))) {
      c->rejected++;
      continue;
    }
    const int k = orc_iterate_mandelbrot(real, imag, max_iterations);
    if (k >= max_iterations) {
      c->never_escaped++;
      c->iterate_steps += (uint64_t) (max_iterations > 0 ? max_iterations : 0);
      continue;
    }
    c->iterate_steps += (uint64_t) k + 1;
    if (k < min_iterations) {
      c->too_fast++;
      continue;
    }
    c->recorded++;
    iterate_and_record(real, imag, hist, dims, atomic, c);
  }
}

static void counters_add(orc_counters *dst, const orc_counters *src) {
  dst->samples += src->samples;
  dst->rejected += src->rejected;
  dst->never_escaped += src->never_escaped;
  dst->too_fast += src->too_fast;
  dst->recorded += src->recorded;
  dst->iterate_steps += src->iterate_steps;
  dst->replay_steps += src->replay_steps;
  dst->increments += src->increments;
}

void orc_draw_buddhabrot(const orc_dims *dims, uint64_t *hist, const orc_iters *it,
                         orc_xorwow *states, uint64_t n_threads, int samples_per_thread,
                         orc_counters *counters) {
  orc_counters c;
  memset(&c, 0, sizeof(c));
  for (uint64_t t = 0; t < n_threads; t++) {
    draw_thread(dims, hist, it, &states[t], samples_per_thread, 0, &c);
  }
  if (counters) counters_add(counters, &c);
}

int orc_draw_buddhabrot_omp(const orc_dims *dims, uint64_t *hist, const orc_iters *it,
                            orc_xorwow *states, uint64_t n_threads, int samples_per_thread,
                            orc_counters *counters, int n_omp_threads) {
  int used = 1;
#ifdef _OPENMP
  if (n_omp_threads <= 0) n_omp_threads = omp_get_max_threads();
  used = n_omp_threads;
#else
  (void) n_omp_threads;
#endif
  orc_counters total;
  memset(&total, 0, sizeof(total));
#pragma omp parallel num_threads(used)
  {
    orc_counters c;
    memset(&c, 0, sizeof(c));
#pragma omp for schedule(dynamic, 64)
    for (int64_t t = 0; t < (int64_t) n_threads; t++) {
      draw_thread(dims, hist, it, &states[t], samples_per_thread, 1, &c);
    }
#pragma omp critical(orc_counters_sum)
    counters_add(&total, &c);
  }
  if (counters) counters_add(counters, &total);
  return used;
}

uint64_t orc_fnv1a_pixels(const uint64_t *hist, uint64_t n) {
  uint64_t fnv = 1469598103934665603ull;
  for (uint64_t i = 0; i < n; i++) fnv = (fnv ^ hist[i]) * 1099511628211ull;
  return fnv;
}

/* ------------------------------------------------------------------------------------------
 * Output stage (host side of the reference; cudabrot.cu:416-468, 548-577)
 * ------------------------------------------------------------------------------------------ */

/* cudabrot.cu:416-420 */
static uint16_t clamp_u16(double v) {
  if (v <= 0) return 0;
  if (v >= 0xffff) return 0xffff;
  return (uint16_t) v;
}

/* double -> uint16_t as the reference's implicit conversions do on x86-64 (cvttsd2si, low 16 bits);
 * NaN (empty histogram: scale = inf, 0*inf) yields 0 there, which this pins. */
static uint16_t to_u16_x86(double v) {
  if (v != v) return 0;
  return (uint16_t) (int64_t) v;
}

uint64_t orc_set_grayscale_pixels(const uint64_t *hist, int w, int h, double gamma,
                                  uint16_t *out, double *scale_out) {
  /* GetLinearColorScale, cudabrot.cu:425-439 */
  uint64_t max = 0;
  const uint64_t n = (uint64_t) w * (uint64_t) h;
  for (uint64_t i = 0; i < n; i++) {
    if (hist[i] > max) max = hist[i];
  }
  const double linear_scale = ((double) 0xffff) / ((double) max);
  if (scale_out) *scale_out = linear_scale;
  /* DoGammaCorrection, cudabrot.cu:443-449 */
  for (uint64_t i = 0; i < n; i++) {
    const double maxv = 0xffff;
    const double scaled = ((double) hist[i]) * linear_scale;
    if (gamma <= 0.0) {
      out[i] = to_u16_x86(scaled);
    } else {
      const double v = maxv * pow(scaled / maxv, 1 / gamma);
      out[i] = (v != v) ? 0 : clamp_u16(v);
    }
  }
  return max;
}

size_t orc_encode_pgm(const uint16_t *gray, int w, int h, uint8_t *buf) {
  /* cudabrot.cu:557-571 */
  int n = sprintf((char *) buf, "P5\n%d %d\n%d\n", w, h, 0xffff);
  uint8_t *p = buf + n;
  const uint64_t px = (uint64_t) w * (uint64_t) h;
  for (uint64_t i = 0; i < px; i++) {
    const uint16_t v = gray[i];
    *p++ = (uint8_t) (v >> 8);
    *p++ = (uint8_t) (v & 0xff);
  }
  return (size_t) (p - buf);
}
