// host_output.cpp -- the output stage of the reference's host driver: tone mapping and the PGM
// writer (cudabrot.cu:416-468, 548-577).  Host code, as in the reference.
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "../../include/cudabrot_amd.h"

namespace {

// Clamp, cudabrot.cu:416-420
inline uint16_t clamp_u16(double v) {
  if (v <= 0) return 0;
  if (v >= 0xffff) return 0xffff;
  return (uint16_t) v;
}

// The reference converts double -> uint16_t implicitly (cudabrot.cu:447); x86-64 does that through
// cvttsd2si and keeps the low 16 bits.  NaN (an all-zero histogram gives scale = inf, 0 * inf) comes
// out as 0 there; pinned here so that the result does not depend on the compiler.
inline uint16_t to_u16(double v) {
  if (v != v) return 0;
  return (uint16_t) (int64_t) v;
}

}  // namespace

extern "C" {

// The value of one pixel: GetLinearColorScale's scale, DoGammaCorrection (or the plain scaling of
// cudabrot.cu:462-466 when gamma <= 0), the conversion to uint16.  ONE definition, used by the host
// path below and by the tables of the device path (tonemap.hip), so that both give the same bytes.
uint16_t cb_tone_value(uint64_t count, uint64_t max, double gamma) {
  const double linear_scale = ((double) 0xffff) / ((double) max);  // cudabrot.cu:436
  const double scaled = ((double) count) * linear_scale;
  if (gamma <= 0.0) return to_u16(scaled);
  const double top = 0xffff;
  const double v = top * pow(scaled / top, 1 / gamma);  // cudabrot.cu:443-449
  return (v != v) ? (uint16_t) 0 : clamp_u16(v);
}

void cb_set_grayscale_pixels(const cb_pixel *hist, int w, int h, double gamma, uint16_t *gray_out,
                             uint64_t *max_out, double *scale_out) {
  const uint64_t n = (uint64_t) w * (uint64_t) h;
  // GetLinearColorScale, cudabrot.cu:425-439
  uint64_t max = 0;
  for (uint64_t i = 0; i < n; i++) {
    if (hist[i] > max) max = hist[i];
  }
  if (max_out) *max_out = max;
  if (scale_out) *scale_out = ((double) 0xffff) / ((double) max);
  for (uint64_t i = 0; i < n; i++) gray_out[i] = cb_tone_value(hist[i], max, gamma);
}

int cb_save_image(const char *path, uint16_t *gray, int w, int h) {
  const uint64_t pixel_count = (uint64_t) w * (uint64_t) h;
  // big-endian samples, cudabrot.cu:566-570
  for (uint64_t i = 0; i < pixel_count; i++) {
    const uint16_t tmp = gray[i];
    gray[i] = (uint16_t) (((tmp & 0xff) << 8) | (tmp >> 8));
  }
  return cb_save_image_be(path, gray, w, h);
}

int cb_save_image_be(const char *path, const uint16_t *gray_be, int w, int h) {
  const uint64_t pixel_count = (uint64_t) w * (uint64_t) h;
  FILE *output = fopen(path, "wb");
  if (!output) return 1;
  if (fprintf(output, "P5\n%d %d\n%d\n", w, h, 0xffff) <= 0) {  // cudabrot.cu:557
    fclose(output);
    return 2;
  }
  if (!fwrite(gray_be, pixel_count * sizeof(uint16_t), 1, output)) {
    fclose(output);
    return 3;
  }
  fclose(output);
  return 0;
}

}  // extern "C"
