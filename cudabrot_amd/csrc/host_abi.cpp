// host_abi.cpp -- the entry points of the C ABI that are plain host arithmetic (no device call), kept apart from
// capi.hip so that the sanitized CPU build (tests/asan) links them without the HIP runtime.
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/cudabrot_amd.h"

extern "C" {

int cb_recompute_pixel_deltas(cb_fractal_dimensions *dims, const char **msg) {
  const char *m = nullptr;
  if (dims->w <= 0) {
    m = "Output width must be positive.";
  } else if (dims->h <= 0) {
    m = "Output height must be positive.";
  } else if (dims->max_real <= dims->min_real) {
    m = "Maximum real value must be greater than minimum real value.";
  } else if (dims->max_imag <= dims->min_imag) {
    // (sic) the reference's wording, cudabrot.cu:520-521
    m = "Minimum imaginary value must be greater than maximum imaginary value.";
  }
  if (m) {
    if (msg) *msg = m;
    return 0;
  }
  dims->delta_imag = (dims->max_imag - dims->min_imag) / ((double) dims->h);
  dims->delta_real = (dims->max_real - dims->min_real) / ((double) dims->w);
  return 1;
}

// The one gate of the test / tuning knobs ("Diagnostics and test knobs"): a CUDABROT_AMD_* variable is
// read only when CUDABROT_AMD_DEBUG=1 is set as well, so a stray variable in a user's environment cannot change
// the path the product takes.
const char *cb_debug_knob(const char *name) {
  const char *gate = getenv("CUDABROT_AMD_DEBUG");
  if (!gate || strcmp(gate, "1") != 0 || !name) return nullptr;
  return getenv(name);
}

size_t cb_rng_state_bytes(uint32_t n_threads) { return (size_t) n_threads * 6u * sizeof(uint32_t); }

}  // extern "C"
