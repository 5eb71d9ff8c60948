/*
 * ref_probe.hip -- the reference's own device functions, compiled FOR THE DEVICE.  TEST INFRASTRUCTURE ONLY.
 *
 * oracle/Makefile (`ref`) extracts cudabrot.cu:43-67 (types) and :284-365 (InMainCardioid, InOrder2Bulb,
 * IncrementPixelCounter, IterateMandelbrot, IterateAndRecord) from /root/reference at build time (the extracted
 * text is deleted afterwards, never committed) and compiles this file with plain `hipcc -O3
 * --offload-arch=gfx950` -- the reference's own `make hip` flags (Makefile:3,15-21), default floating-point
 * contraction, none of this repository's flags -- into oracle/_ref/libref_probe.so.  Those lines use nothing
 * but the types of :43-67, so no stand-in of any kind is involved: this is what the reference's HIP build
 * computes on gfx950, and tests/test_gpu_ref_probe.py holds the oracle (and through it the product) to it.
 *
 * The kernels below only CALL the reference's functions: one thread per point for the escape index, and a
 * single thread for the recording (the reference's `+=` of :312 is not atomic; one thread is its race-free
 * meaning).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "_ref/ref_types.inc"  /* cudabrot.cu:43-67   Pixel, FractalDimensions, IterationControl */
#include "_ref/ref_device.inc" /* cudabrot.cu:284-365 the five device functions                  */

__global__ void probe_points_kernel(const double *re, const double *im, int n, int max_iter, int *k_out,
                                    int *shortcut_out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  shortcut_out[t] = (InMainCardioid(re[t], im[t]) ? 1 : 0) | (InOrder2Bulb(re[t], im[t]) ? 2 : 0);
  k_out[t] = IterateMandelbrot(re[t], im[t], max_iter);
}

__global__ void probe_record_kernel(FractalDimensions d, const double *re, const double *im, int n, Pixel *hist) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  for (int t = 0; t < n; t++) IterateAndRecord(re[t], im[t], hist, &d);
}

#define PROBE_TRY(e)                 \
  do {                               \
    hipError_t pe_ = (e);            \
    if (pe_ != hipSuccess) {         \
      rc = (int) pe_;                \
      goto done;                     \
    }                                \
  } while (0)

extern "C" {

/* k_out[t] = IterateMandelbrot(re[t], im[t], max_iter); shortcut_out[t] = InMainCardioid | InOrder2Bulb << 1.
 * Host pointers.  Returns 0 or a hipError_t. */
int ref_probe_points(const double *re, const double *im, int n, int max_iter, int *k_out, int *shortcut_out) {
  int rc = 0;
  double *d_re = nullptr, *d_im = nullptr;
  int *d_k = nullptr, *d_s = nullptr;
  PROBE_TRY(hipMalloc(&d_re, sizeof(double) * n));
  PROBE_TRY(hipMalloc(&d_im, sizeof(double) * n));
  PROBE_TRY(hipMalloc(&d_k, sizeof(int) * n));
  PROBE_TRY(hipMalloc(&d_s, sizeof(int) * n));
  PROBE_TRY(hipMemcpy(d_re, re, sizeof(double) * n, hipMemcpyHostToDevice));
  PROBE_TRY(hipMemcpy(d_im, im, sizeof(double) * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(probe_points_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, d_re, d_im, n, max_iter, d_k, d_s);
  PROBE_TRY(hipGetLastError());
  PROBE_TRY(hipDeviceSynchronize());
  PROBE_TRY(hipMemcpy(k_out, d_k, sizeof(int) * n, hipMemcpyDeviceToHost));
  PROBE_TRY(hipMemcpy(shortcut_out, d_s, sizeof(int) * n, hipMemcpyDeviceToHost));
done:
  (void) hipFree(d_re);
  (void) hipFree(d_im);
  (void) hipFree(d_k);
  (void) hipFree(d_s);
  return rc;
}

/* IterateAndRecord for the n given starting points (each must escape), one after another in ONE device
 * thread, into hist_out (w*h Pixel = uint32, zeroed here).  Canvas as cudabrot.cu:524-525 computes it. */
int ref_probe_record(int w, int h, double min_real, double max_real, double min_imag, double max_imag,
                     const double *re, const double *im, int n, uint32_t *hist_out) {
  int rc = 0;
  FractalDimensions d;
  memset(&d, 0, sizeof(d));
  d.w = w;
  d.h = h;
  d.min_real = min_real;
  d.max_real = max_real;
  d.min_imag = min_imag;
  d.max_imag = max_imag;
  d.delta_imag = (d.max_imag - d.min_imag) / ((double) d.h);
  d.delta_real = (d.max_real - d.min_real) / ((double) d.w);
  const size_t hist_bytes = sizeof(Pixel) * (size_t) w * (size_t) h;
  double *d_re = nullptr, *d_im = nullptr;
  Pixel *d_hist = nullptr;
  PROBE_TRY(hipMalloc(&d_re, sizeof(double) * n));
  PROBE_TRY(hipMalloc(&d_im, sizeof(double) * n));
  PROBE_TRY(hipMalloc(&d_hist, hist_bytes));
  PROBE_TRY(hipMemset(d_hist, 0, hist_bytes));
  PROBE_TRY(hipMemcpy(d_re, re, sizeof(double) * n, hipMemcpyHostToDevice));
  PROBE_TRY(hipMemcpy(d_im, im, sizeof(double) * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(probe_record_kernel, dim3(1), dim3(1), 0, 0, d, d_re, d_im, n, d_hist);
  PROBE_TRY(hipGetLastError());
  PROBE_TRY(hipDeviceSynchronize());
  PROBE_TRY(hipMemcpy(hist_out, d_hist, hist_bytes, hipMemcpyDeviceToHost));
done:
  (void) hipFree(d_re);
  (void) hipFree(d_im);
  (void) hipFree(d_hist);
  return rc;
}

}  /* extern "C" */
