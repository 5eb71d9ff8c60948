// tonemap.hip -- the output stage on the device: SetGrayscalePixels
// (cudabrot.cu:425-468) and the byte swap of SaveImage (cudabrot.cu:566-570).
//
// The reference scans the histogram for its maximum, maps every pixel through
//   v = 65535 * pow(count * scale / 65535, 1 / gamma)          (cudabrot.cu:443-449)
// on one host thread and swaps the bytes of the result: at 20000 x 20000 that is a 3.2 GB copy to
// the host and tens of seconds of pow().  Here the histogram stays on the device; only 2 bytes per
// pixel, already big-endian, travel.
//
// Bit-exactness.  The value of a pixel depends on its count alone (and on max, gamma), and pow() of
// the host's libm and of the device library differ in the last place, which after the truncation to
// uint16 shows as off-by-one pixels.  So the HOST evaluates the map -- the same function
// cb_set_grayscale_pixels uses (cb_tone_value, host_output.cpp) -- and the device only looks it up:
//   CB_TONE_LUT         one uint16 per count in [0, max]: identical to the host path by construction
//   CB_TONE_THRESHOLDS  for every output value k the smallest count that reaches it (65535
//                       bisections over the host map); a pixel's value is the number of thresholds
//                       <= its count.  Identical whenever the host map is monotone in the count,
//                       which holds for a pow() that is within an ulp of exact; used when max is too
//                       large for a table (> 2^24).
#include <math.h>
#include <stdlib.h>

#include <vector>

#include "kernels.h"

extern "C" uint16_t cb_tone_value(uint64_t count, uint64_t max, double gamma);

namespace cb {

namespace {

constexpr uint32_t kToneThreads = 256;
constexpr uint64_t kLutMaxEntries = 1ull << 24;

__global__ void __launch_bounds__(kToneThreads) hist_max_kernel(const unsigned long long *hist,
                                                                unsigned long long n,
                                                                unsigned long long *out) {
  unsigned long long m = 0;
  const unsigned long long stride = (unsigned long long) gridDim.x * kToneThreads;
  for (unsigned long long i = (unsigned long long) blockIdx.x * kToneThreads + threadIdx.x; i < n;
       i += stride) {
    const unsigned long long v = hist[i];
    m = v > m ? v : m;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long o = __shfl_xor(m, d, 64);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63u) == 0u && m != 0ull) {
    __hip_atomic_fetch_max(out, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ __forceinline__ uint16_t big_endian(uint32_t v) {
  return (uint16_t) (((v & 0xffu) << 8) | ((v >> 8) & 0xffu));
}

// two pixels per thread: one 4-byte store
__global__ void __launch_bounds__(kToneThreads) tone_lut_kernel(const unsigned long long *hist,
                                                                unsigned long long n,
                                                                const uint16_t *lut, uint16_t *out_be) {
  const unsigned long long stride = (unsigned long long) gridDim.x * kToneThreads * 2ull;
  for (unsigned long long i = ((unsigned long long) blockIdx.x * kToneThreads + threadIdx.x) * 2ull;
       i < n; i += stride) {
    const uint32_t a = big_endian(lut[hist[i]]);
    if (i + 1 < n) {
      const uint32_t b = big_endian(lut[hist[i + 1]]);
      *reinterpret_cast<uint32_t *>(out_be + i) = a | (b << 16);  // n even or not: i is even
    } else {
      out_be[i] = (uint16_t) a;
    }
  }
}

// thr[k], k in [0, 65536): smallest count whose value is >= k (thr[0] = 0; ~0 if none).  The value
// of a count is the largest k with thr[k] <= count.
__device__ __forceinline__ uint32_t value_by_thresholds(unsigned long long c,
                                                        const unsigned long long *thr) {
  uint32_t lo = 0, hi = 65536;  // invariant: thr[lo] <= c, and (hi == 65536 or thr[hi] > c)
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (thr[mid] <= c) {
      lo = mid;
    } else {
      hi = mid;
    }
  }
  return lo;
}

__global__ void __launch_bounds__(kToneThreads) tone_thresholds_kernel(const unsigned long long *hist,
                                                                       unsigned long long n,
                                                                       const unsigned long long *thr,
                                                                       uint16_t *out_be) {
  const unsigned long long stride = (unsigned long long) gridDim.x * kToneThreads * 2ull;
  for (unsigned long long i = ((unsigned long long) blockIdx.x * kToneThreads + threadIdx.x) * 2ull;
       i < n; i += stride) {
    const uint32_t a = big_endian(value_by_thresholds(hist[i], thr));
    if (i + 1 < n) {
      const uint32_t b = big_endian(value_by_thresholds(hist[i + 1], thr));
      *reinterpret_cast<uint32_t *>(out_be + i) = a | (b << 16);
    } else {
      out_be[i] = (uint16_t) a;
    }
  }
}

uint32_t tone_grid(unsigned long long n) {
  unsigned long long blocks = (n / 2 + kToneThreads) / kToneThreads;
  if (blocks > 256ull * 16ull) blocks = 256ull * 16ull;
  return (uint32_t) (blocks ? blocks : 1);
}

#define CB_TONE_TRY(expr)                           \
  do {                                              \
    const hipError_t e_ = (expr);                   \
    if (e_ != hipSuccess) return (int) e_;          \
  } while (0)

struct DeviceBuffer {  // freed on every return path
  void *p = nullptr;
  ~DeviceBuffer() {
    if (p) (void) hipFree(p);
  }
};

// thr[k] = smallest count in [0, max] whose value is >= k, by bisection over the host map
std::vector<unsigned long long> build_thresholds(unsigned long long max, double gamma) {
  std::vector<unsigned long long> thr(65536, ~0ull);
  thr[0] = 0;
  const uint32_t top = cb_tone_value(max, max, gamma);
  unsigned long long lo = 0;  // thr is non-decreasing: each search starts at the previous answer
  for (uint32_t k = 1; k <= top; ++k) {
    unsigned long long a = lo, b = max;  // value(max) = top >= k
    if (cb_tone_value(a, max, gamma) >= k) {
      b = a;
    } else {
      while (b - a > 1ull) {  // value(a) < k <= value(b)
        const unsigned long long mid = a + (b - a) / 2ull;
        if (cb_tone_value(mid, max, gamma) >= k) {
          b = mid;
        } else {
          a = mid;
        }
      }
    }
    thr[k] = b;
    lo = b;
  }
  return thr;
}

}  // namespace

}  // namespace cb

extern "C" int cb_tone_map_device(const cb_pixel *d_hist, int w, int h, double gamma, int mode,
                                  uint16_t *d_gray_be, uint64_t *max_out, double *scale_out,
                                  void *stream_v) {
  using namespace cb;
  if (!d_hist || !d_gray_be || w <= 0 || h <= 0) return (int) hipErrorInvalidValue;
  if (mode != CB_TONE_AUTO && mode != CB_TONE_LUT && mode != CB_TONE_THRESHOLDS) {
    return (int) hipErrorInvalidValue;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  const unsigned long long n = (unsigned long long) w * (unsigned long long) h;
  const unsigned long long *hist = reinterpret_cast<const unsigned long long *>(d_hist);
  DeviceBuffer d_max, d_table;
  unsigned long long max = 0;

  // GetLinearColorScale, cudabrot.cu:425-439
  CB_TONE_TRY(hipMalloc(&d_max.p, sizeof(unsigned long long)));
  CB_TONE_TRY(hipMemsetAsync(d_max.p, 0, sizeof(unsigned long long), stream));
  hipLaunchKernelGGL(hist_max_kernel, dim3(tone_grid(n)), dim3(kToneThreads), 0, stream, hist, n,
                     static_cast<unsigned long long *>(d_max.p));
  CB_TONE_TRY(hipGetLastError());
  CB_TONE_TRY(hipMemcpyAsync(&max, d_max.p, sizeof(max), hipMemcpyDeviceToHost, stream));
  CB_TONE_TRY(hipStreamSynchronize(stream));
  if (max_out) *max_out = max;
  if (scale_out) *scale_out = ((double) 0xffff) / ((double) max);

  if (mode == CB_TONE_AUTO) mode = (max < kLutMaxEntries) ? CB_TONE_LUT : CB_TONE_THRESHOLDS;
  if (mode == CB_TONE_LUT) {
    if (max >= (1ull << 32)) return (int) hipErrorInvalidValue;  // not a sensible table
    std::vector<uint16_t> lut((size_t) max + 1);
    for (unsigned long long c = 0; c <= max; ++c) lut[(size_t) c] = cb_tone_value(c, max, gamma);
    CB_TONE_TRY(hipMalloc(&d_table.p, lut.size() * sizeof(uint16_t)));
    CB_TONE_TRY(hipMemcpyAsync(d_table.p, lut.data(), lut.size() * sizeof(uint16_t),
                               hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(tone_lut_kernel, dim3(tone_grid(n)), dim3(kToneThreads), 0, stream, hist, n,
                       static_cast<const uint16_t *>(d_table.p), d_gray_be);
    CB_TONE_TRY(hipGetLastError());
    CB_TONE_TRY(hipStreamSynchronize(stream));  // before the host table goes out of scope
  } else {
    const std::vector<unsigned long long> thr = build_thresholds(max, gamma);
    CB_TONE_TRY(hipMalloc(&d_table.p, thr.size() * sizeof(unsigned long long)));
    CB_TONE_TRY(hipMemcpyAsync(d_table.p, thr.data(), thr.size() * sizeof(unsigned long long),
                               hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(tone_thresholds_kernel, dim3(tone_grid(n)), dim3(kToneThreads), 0, stream, hist,
                       n, static_cast<const unsigned long long *>(d_table.p), d_gray_be);
    CB_TONE_TRY(hipGetLastError());
    CB_TONE_TRY(hipStreamSynchronize(stream));
  }
  return 0;
}
