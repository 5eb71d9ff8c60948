/*
 * buddha_oracle.h -- CPU restatement of the cudabrot hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the checker, not the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Nothing under cudabrot_amd/ links, imports or calls it.
 *
 * Parity pin: the single-thread form is checked bit-for-bit against
 *   (1) the reference's own lines (cudabrot.cu:43-67,284-414) compiled for the host by
 *       oracle/Makefile into oracle/_ref/ (tests/test_oracle_vs_ref.py, container only), and
 *   (2) the known-answer table of SURVEY.md Appendix B, committed under tests/golden/.
 *
 * Every function cites the reference lines (cudabrot.cu) or the rocRAND 4.2 header lines
 * (ROCRAND_VERSION 400200, /opt/rocm/include/rocrand) it restates.
 */
#ifndef BUDDHA_ORACLE_H_
#define BUDDHA_ORACLE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cudabrot.cu:46-58 */
typedef struct {
  int w, h;
  double min_real, min_imag, max_real, max_imag;
  double delta_real, delta_imag;
} orc_dims;

/* cudabrot.cu:62-67 */
typedef struct {
  int max_escape_iterations;
  int min_escape_iterations;
} orc_iters;

/* The live part of rocrand_state_xorwow (rocrand_xorwow.h:72-89): Weyl value + 160 xorshift bits. */
typedef struct {
  uint32_t d;
  uint32_t x[5];
} orc_xorwow;

/* Workload statistics; all exact counts. */
typedef struct {
  uint64_t samples;          /* starting points drawn (4 u32 draws each)            */
  uint64_t rejected;         /* inside main cardioid / period-2 bulb                */
  uint64_t never_escaped;    /* IterateMandelbrot returned max_iterations           */
  uint64_t too_fast;         /* escaped with k < min_escape_iterations              */
  uint64_t recorded;         /* orbits replayed into the histogram                  */
  uint64_t iterate_steps;    /* z<-z^2+c iterations executed by IterateMandelbrot   */
  uint64_t replay_steps;     /* iterations executed by IterateAndRecord             */
  uint64_t increments;       /* in-canvas histogram increments                      */
} orc_counters;

/* cudabrot.cu:505-527: validation + delta computation.  Returns 1 if valid, else 0. */
int orc_recompute_pixel_deltas(orc_dims *d);

/* rocrand_init(seed, subsequence, offset, &state), rocrand_xorwow.h:100-131 (cudabrot.cu:148). */
void orc_xorwow_init(uint64_t seed, uint64_t subsequence, uint64_t offset, orc_xorwow *st);
/* Same result as n calls of orc_xorwow_init for subsequences first..first+n-1 (one mat-vec each). */
void orc_xorwow_init_range(uint64_t seed, uint64_t first, uint64_t n, orc_xorwow *st);
/* rocrand(&state), rocrand_xorwow.h:165-177 */
uint32_t orc_xorwow_next(orc_xorwow *st);
/* rocrand_uniform_double(&state), rocrand_uniform.h:102-109,454-460 */
double orc_uniform_double(orc_xorwow *st);
/* Row-major [5*32][5] image table of A^(2^(67+2*i)), the layout of rocRAND's
 * h_xorwow_sequence_jump_matrices[i] (rocrand_xorwow_precomputed.h:3313); i in [0,32). */
void orc_xorwow_sequence_jump_matrix(int i, uint32_t out[800]);
/* Likewise A^(4^i) = h_xorwow_jump_matrices[i] (rocrand_xorwow_precomputed.h:1127). */
void orc_xorwow_jump_matrix(int i, uint32_t out[800]);

/* cudabrot.cu:284-298 */
int orc_in_main_cardioid(double real, double imag);
int orc_in_order2_bulb(double real, double imag);
/* cudabrot.cu:319-340 */
int orc_iterate_mandelbrot(double start_real, double start_imag, int max_iterations);
/* cudabrot.cu:347-365 (IterateAndRecord, with IncrementPixelCounter :302-314) for one starting point that is
 * known to escape; returns the iterations executed, adds the in-canvas increments to *increments. */
uint64_t orc_iterate_and_record(const orc_dims *dims, uint64_t *hist, double start_real, double start_imag,
                                uint64_t *increments);
/* RENDER_BURNING_SHIP (cudabrot.cu:15-17) as a run-time switch of this library (process-wide). */
void orc_set_burning_ship(int on);
int orc_get_burning_ship(void);

/*
 * DrawBuddhabrot (cudabrot.cu:379-414) for "threads" [0,n_threads), executed one after another
 * (race-free semantics of the non-atomic += at :312), samples_per_thread samples each (reference: 50).
 * hist is u64[w*h]; states[t] is advanced in place.  counters may be NULL.
 */
void orc_draw_buddhabrot(const orc_dims *dims, uint64_t *hist, const orc_iters *it,
                         orc_xorwow *states, uint64_t n_threads, int samples_per_thread,
                         orc_counters *counters);
/* Same multiset of samples, threads spread over OpenMP workers, atomic u64 increments.
 * n_omp_threads<=0 means omp_get_max_threads().  Returns the worker count used. */
int orc_draw_buddhabrot_omp(const orc_dims *dims, uint64_t *hist, const orc_iters *it,
                            orc_xorwow *states, uint64_t n_threads, int samples_per_thread,
                            orc_counters *counters, int n_omp_threads);

/* Width-independent pixel-value hash of SURVEY.md Appendix A step 4. */
uint64_t orc_fnv1a_pixels(const uint64_t *hist, uint64_t n);

/* SetGrayscalePixels (cudabrot.cu:416-468): out[w*h] host-endian u16; returns max count,
 * *scale_out = 65535/max. */
uint64_t orc_set_grayscale_pixels(const uint64_t *hist, int w, int h, double gamma,
                                  uint16_t *out, double *scale_out);
/* SaveImage (cudabrot.cu:548-577): writes the PGM to buf (header + big-endian u16), returns bytes
 * written; buf must hold 64 + 2*w*h bytes. */
size_t orc_encode_pgm(const uint16_t *gray, int w, int h, uint8_t *buf);

#ifdef __cplusplus
}
#endif
#endif
