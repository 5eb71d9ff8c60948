/*
 * ref_driver.cpp -- sequential host driver around the reference's own lines.  TEST INFRASTRUCTURE
 * ONLY.  oracle/Makefile extracts the line ranges below from /root/reference/cudabrot.cu into
 * oracle/_ref/*.inc at build time (never committed) and compiles this file twice:
 *   libref_fma.so : clang++ -ffp-contract=fast -mfma -fno-slp-vectorize -fno-vectorize
 *                   (= the contraction hipcc applies on gfx950 in both loops)
 *   libref_off.so : clang++ -ffp-contract=off   (ISO unfused; informational)
 * "Threads" run one after another, so the non-atomic += of cudabrot.cu:312 is race-free: these
 * are the reference's sequential semantics.
 */
#include "ref_shim.h"

#include "_ref/ref_types.inc"   /* cudabrot.cu:43-67   Pixel, FractalDimensions, IterationControl */
#include "_ref/ref_globals.inc" /* cudabrot.cu:70-101  the global struct g                         */
#include "_ref/ref_kernel.inc"  /* cudabrot.cu:284-414 device functions + DrawBuddhabrot           */
#include "_ref/ref_tonemap.inc" /* cudabrot.cu:416-468 Clamp .. SetGrayscalePixels                 */
#include "_ref/ref_save.inc"    /* cudabrot.cu:548-577 SaveImage                                   */

extern "C" {

/* cudabrot.cu:524-525 */
static void set_dims(FractalDimensions *d, int w, int h, double min_real, double max_real,
                     double min_imag, double max_imag) {
  memset(d, 0, sizeof(*d));
  d->w = w;
  d->h = h;
  d->min_real = min_real;
  d->max_real = max_real;
  d->min_imag = min_imag;
  d->max_imag = max_imag;
  d->delta_imag = (d->max_imag - d->min_imag) / ((double) d->h);
  d->delta_real = (d->max_real - d->min_real) / ((double) d->w);
}

/* Runs `passes` launches of DrawBuddhabrot over "threads" first_subseq .. first_subseq+n_threads-1
 * with seed 1337 (cudabrot.cu:37,148,179) into hist (Pixel = uint32_t, caller-zeroed). */
int ref_draw(int w, int h, double min_real, double max_real, double min_imag, double max_imag,
             int max_iter, int min_iter, unsigned long long first_subseq,
             unsigned long long n_threads, int passes, int samples_per_thread, uint32_t *hist) {
  FractalDimensions d;
  IterationControl it;
  set_dims(&d, w, h, min_real, max_real, min_imag, max_imag);
  it.max_escape_iterations = max_iter;
  it.min_escape_iterations = min_iter;
  ref_samples_per_thread = samples_per_thread;
  curandState_t *st = (curandState_t *) malloc(sizeof(curandState_t) * n_threads);
  if (!st) return 1;
  blockDim.x = 1;
  threadIdx.x = 0;
  for (unsigned long long t = 0; t < n_threads; t++) {
    curand_init(1337, first_subseq + t, 0, st + t);
  }
  for (int p = 0; p < passes; p++) {
    for (unsigned long long t = 0; t < n_threads; t++) {
      blockIdx.x = (unsigned) t;
      DrawBuddhabrot(d, (Pixel *) hist, it, st);
    }
  }
  free(st);
  return 0;
}

/* First n u32 outputs of (seed, subsequence, offset 0). */
void ref_rng_u32(unsigned long long seed, unsigned long long subseq, int n, uint32_t *out) {
  curandState_t st;
  curand_init(seed, subseq, 0, &st);
  for (int i = 0; i < n; i++) out[i] = rocrand(&st);
}

/* First sample as cudabrot.cu:392-393 draws it. */
void ref_first_sample(unsigned long long seed, unsigned long long subseq, double *real,
                      double *imag) {
  curandState_t st;
  curand_init(seed, subseq, 0, &st);
  *real = (curand_uniform_double(&st) * 4.0) - 2.0;
  *imag = (curand_uniform_double(&st) * 4.0) - 2.0;
}

int ref_iterate_mandelbrot(double re, double im, int max_iter) {
  return IterateMandelbrot(re, im, max_iter);
}

int ref_in_set_shortcut(double re, double im) {
  return (InMainCardioid(re, im) ? 1 : 0) | (InOrder2Bulb(re, im) ? 2 : 0);
}

/* SetGrayscalePixels on a caller-provided u32 histogram; prints the reference's "Max value" line. */
void ref_set_grayscale_pixels(uint32_t *hist, int w, int h, double gamma, uint16_t *out) {
  memset(&g, 0, sizeof(g));
  g.dimensions.w = w;
  g.dimensions.h = h;
  g.host_buddhabrot = hist;
  g.grayscale_image = out;
  g.gamma_correction = gamma;
  SetGrayscalePixels();
  fflush(stdout);
}

/* SaveImage; byte-swaps gray in place like the reference. */
void ref_save_image(const char *path, uint16_t *gray, int w, int h) {
  memset(&g, 0, sizeof(g));
  g.dimensions.w = w;
  g.dimensions.h = h;
  g.grayscale_image = gray;
  g.output_image = path;
  SaveImage();
  fflush(stdout);
}

}  /* extern "C" */
