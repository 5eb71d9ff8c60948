// scatter.hip -- the deferred, tile-binned scatter (see kernels.h, BinLayout).
//
// Input: the pixel stream the REPLAY stage of draw_wave_kernel wrote -- per wave a region of packed
// (row << 16 | col) words.  Output: the same increments added to the u64 histogram, with one
// 64-byte memory-side request per 8 pixels of a tile instead of one per increment.
//
//   bin_count_kernel       one workgroup per wave region: LDS histogram over tiles -> count[wave][tile]
//   bin_scan_waves_kernel  per tile: exclusive prefix over waves (in place), tile totals
//   bin_scan_tiles_kernel  exclusive prefix over the tile totals -> tile_base[]
//   bin_scatter_kernel     one workgroup per wave region, chunks of 8192 entries: rank inside
//                          (chunk, tile) by LDS atomics, sort the chunk in LDS, write each tile's run to
//                          its place in `sorted` as 14-bit in-tile offsets (coalesced 2-byte stores)
//   bin_accumulate_kernel  one workgroup per tile: LDS u32 histogram of the tile's bucket, then the
//                          tile is added to the u64 histogram with coalesced device-scope atomics
//
// Everything is a counting sort: no global atomics before the final flush, and the bytes written are
// the same from run to run.  All of it is HBM-streaming work (6 + 4 + 4 + 2 + 2 = 18 bytes per
// increment end to end) that the draw kernel's fp64 loop leaves idle.
#include "kernels.h"

namespace cb {

namespace {

constexpr uint32_t kScatterThreads = 512;
constexpr uint32_t kChunkEntries = 8192;  // 16 per thread
constexpr uint32_t kPerThread = kChunkEntries / kScatterThreads;

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

__device__ __forceinline__ uint32_t tile_of(uint32_t e, uint32_t tiles_x) {
  return ((e >> 16) >> kTileShift) * tiles_x + ((e & 0xffffu) >> kTileShift);
}
__device__ __forceinline__ uint32_t offset_of(uint32_t e) {
  return (((e >> 16) & (kTileSize - 1u)) << kTileShift) | (e & (kTileSize - 1u));
}
__device__ __forceinline__ void lds_inc(uint32_t *p) {
  __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_u32
}

__global__ void __launch_bounds__(256) bin_count_kernel(BinLayout b) {
  extern __shared__ uint32_t lds[];  // [n_tiles]
  const uint32_t wv = blockIdx.x;
  const uint32_t n = b.wave_count[wv];
  const uint32_t *src = b.stream + (size_t) wv * b.cap;
  for (uint32_t t = threadIdx.x; t < b.n_tiles; t += blockDim.x) lds[t] = 0u;
  __syncthreads();
  const uint32_t n4 = n >> 2;
  const uint4 *src4 = reinterpret_cast<const uint4 *>(src);  // region base is 16-byte aligned
  for (uint32_t i = threadIdx.x; i < n4; i += blockDim.x) {
    const uint4 v = src4[i];
    lds_inc(&lds[tile_of(v.x, b.tiles_x)]);
    lds_inc(&lds[tile_of(v.y, b.tiles_x)]);
    lds_inc(&lds[tile_of(v.z, b.tiles_x)]);
    lds_inc(&lds[tile_of(v.w, b.tiles_x)]);
  }
  for (uint32_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) {
    lds_inc(&lds[tile_of(src[i], b.tiles_x)]);
  }
  __syncthreads();
  uint32_t *dst = b.count + (size_t) wv * b.n_tiles;
  for (uint32_t t = threadIdx.x; t < b.n_tiles; t += blockDim.x) dst[t] = lds[t];
}

// count[w][t] <- sum of count[w'][t] for w' < w; tile_base[t] <- total of tile t (scanned next).
__global__ void __launch_bounds__(256) bin_scan_waves_kernel(BinLayout b) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= b.n_tiles) return;
  unsigned long long run = 0;
  for (uint32_t w = 0; w < b.n_waves; ++w) {
    uint32_t *p = b.count + (size_t) w * b.n_tiles + t;
    const uint32_t c = *p;
    *p = (uint32_t) run;
    run += c;
  }
  b.tile_base[t] = run;
}

// tile_base[t] <- sum of totals of tiles < t; tile_base[n_tiles] <- grand total.  One workgroup.
__global__ void __launch_bounds__(1024) bin_scan_tiles_kernel(BinLayout b) {
  __shared__ unsigned long long part[1024];
  constexpr uint32_t kPer = kMaxTiles / 1024;  // 4 tiles per thread
  unsigned long long v[kPer];
  unsigned long long sum = 0;
  for (uint32_t k = 0; k < kPer; ++k) {
    const uint32_t t = threadIdx.x * kPer + k;
    v[k] = t < b.n_tiles ? b.tile_base[t] : 0ull;
    sum += v[k];
  }
  part[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
    const unsigned long long add = threadIdx.x >= d ? part[threadIdx.x - d] : 0ull;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  unsigned long long run = part[threadIdx.x] - sum;
  for (uint32_t k = 0; k < kPer; ++k) {
    const uint32_t t = threadIdx.x * kPer + k;
    if (t < b.n_tiles) b.tile_base[t] = run;
    run += v[k];
  }
  if (threadIdx.x == 1023) b.tile_base[b.n_tiles] = part[1023];
}

__global__ void __launch_bounds__(kScatterThreads) bin_scatter_kernel(BinLayout b) {
  extern __shared__ uint32_t lds[];
  // dynamic LDS: cursor[n_tiles] | cnt[n_tiles] | lstart[n_tiles] | part[512] | pos[chunk] | off[chunk]
  uint32_t *cursor = lds;
  uint32_t *cnt = cursor + b.n_tiles;
  uint32_t *lstart = cnt + b.n_tiles;
  uint32_t *part = lstart + b.n_tiles;
  uint32_t *pos = part + kScatterThreads;
  uint16_t *off = reinterpret_cast<uint16_t *>(pos + kChunkEntries);

  const uint32_t wv = blockIdx.x;
  const uint32_t n = b.wave_count[wv];
  const uint32_t *src = b.stream + (size_t) wv * b.cap;
  const uint32_t *prefix = b.count + (size_t) wv * b.n_tiles;
  for (uint32_t t = threadIdx.x; t < b.n_tiles; t += kScatterThreads) {
    cursor[t] = (uint32_t) b.tile_base[t] + prefix[t];  // all entries together are < 2^32
  }
  // bins per thread for the in-chunk scan
  const uint32_t bins_per_thread = (b.n_tiles + kScatterThreads - 1u) / kScatterThreads;

  for (uint32_t base = 0; base < n; base += kChunkEntries) {
    const uint32_t m = (n - base) < kChunkEntries ? (n - base) : kChunkEntries;
    for (uint32_t t = threadIdx.x; t < b.n_tiles; t += kScatterThreads) cnt[t] = 0u;
    __syncthreads();
    // 1. rank of every entry inside (chunk, tile)
    uint32_t e[kPerThread], r[kPerThread];
#pragma unroll
    for (uint32_t k = 0; k < kPerThread; ++k) {
      const uint32_t i = k * kScatterThreads + threadIdx.x;
      e[k] = 0u;
      r[k] = 0u;
      if (i < m) {
        e[k] = src[base + i];
        r[k] = __hip_atomic_fetch_add(&cnt[tile_of(e[k], b.tiles_x)], 1u, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_rtn_u32
      }
    }
    __syncthreads();
    // 2. exclusive scan of cnt over tiles -> lstart (start of each tile's run in the sorted chunk)
    {
      const uint32_t t0 = threadIdx.x * bins_per_thread;
      uint32_t sum = 0;
      for (uint32_t k = 0; k < bins_per_thread; ++k) {
        const uint32_t t = t0 + k;
        if (t < b.n_tiles) sum += cnt[t];
      }
      part[threadIdx.x] = sum;
      __syncthreads();
      for (uint32_t d = 1; d < kScatterThreads; d <<= 1) {
        const uint32_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
      }
      uint32_t run = part[threadIdx.x] - sum;
      for (uint32_t k = 0; k < bins_per_thread; ++k) {
        const uint32_t t = t0 + k;
        if (t < b.n_tiles) {
          lstart[t] = run;
          run += cnt[t];
        }
      }
    }
    __syncthreads();
    // 3. sort the chunk in LDS: destination index in `sorted` and in-tile offset, tile by tile
#pragma unroll
    for (uint32_t k = 0; k < kPerThread; ++k) {
      const uint32_t i = k * kScatterThreads + threadIdx.x;
      if (i < m) {
        const uint32_t t = tile_of(e[k], b.tiles_x);
        const uint32_t lp = lstart[t] + r[k];
        pos[lp] = cursor[t] + r[k];
        off[lp] = (uint16_t) offset_of(e[k]);
      }
    }
    __syncthreads();
    // 4. write the runs out (consecutive lanes -> consecutive places of a run) and advance cursors
    for (uint32_t i = threadIdx.x; i < m; i += kScatterThreads) b.sorted[pos[i]] = off[i];
    for (uint32_t t = threadIdx.x; t < b.n_tiles; t += kScatterThreads) cursor[t] += cnt[t];
    __syncthreads();
  }
}

__global__ void __launch_bounds__(512) bin_accumulate_kernel(BinLayout b, unsigned long long *hist,
                                                             int w, int h) {
  __shared__ uint32_t tile[kTilePixels];  // 64 KiB
  const uint32_t t = blockIdx.x;
  const unsigned long long begin = b.tile_base[t];
  const unsigned long long end = b.tile_base[t + 1];
  if (begin == end) return;  // wave-uniform: nothing landed on this tile
  for (uint32_t p = threadIdx.x; p < (uint32_t) kTilePixels; p += blockDim.x) tile[p] = 0u;
  __syncthreads();
  for (unsigned long long i = begin + threadIdx.x; i < end; i += blockDim.x) {
    lds_inc(&tile[b.sorted[i]]);
  }
  __syncthreads();
  const uint32_t row0 = (t / b.tiles_x) << kTileShift;
  const uint32_t col0 = (t % b.tiles_x) << kTileShift;
  for (uint32_t p = threadIdx.x; p < (uint32_t) kTilePixels; p += blockDim.x) {
    const uint32_t v = tile[p];
    if (v != 0u) {
      const uint32_t row = row0 + (p >> kTileShift);
      const uint32_t col = col0 + (p & (kTileSize - 1u));
      if (row < (uint32_t) h && col < (uint32_t) w) {  // always true for a recorded pixel
        __hip_atomic_fetch_add(hist + ((unsigned long long) row * (unsigned long long) w + col),
                               (unsigned long long) v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

}  // namespace

size_t bin_fixed_bytes(uint32_t n_waves, uint32_t n_tiles) {
  return round_up((size_t) n_waves * sizeof(uint32_t), 256) +
         round_up((size_t) n_waves * n_tiles * sizeof(uint32_t), 256) +
         round_up(((size_t) n_tiles + 1) * sizeof(unsigned long long), 256) + 512;
}

BinLayout make_bin_layout(void *workspace, size_t bytes, int w, int h, uint32_t n_waves) {
  BinLayout b;
  b.enabled = 0;
  b.n_waves = n_waves;
  b.cap = 0;
  b.n_tiles = 0;
  b.tiles_x = 0;
  b.wave_count = nullptr;
  b.stream = nullptr;
  b.count = nullptr;
  b.tile_base = nullptr;
  b.sorted = nullptr;
  if (!workspace || n_waves == 0 || w <= 0 || h <= 0 || w > 65536 || h > 65536) return b;
  const uint32_t tiles_x = ((uint32_t) w + kTileSize - 1u) >> kTileShift;
  const uint32_t tiles_y = ((uint32_t) h + kTileSize - 1u) >> kTileShift;
  if ((unsigned long long) tiles_x * tiles_y > kMaxTiles) return b;
  b.n_tiles = tiles_x * tiles_y;
  b.tiles_x = tiles_x;
  // align the carve to 256 bytes
  uintptr_t p = reinterpret_cast<uintptr_t>(workspace);
  const uintptr_t p_end = p + bytes;
  p = (p + 255) & ~(uintptr_t) 255;
  const size_t fixed = bin_fixed_bytes(n_waves, b.n_tiles);
  if (p + fixed >= p_end) return b;
  unsigned long long cap = (p_end - p - fixed) / (kBinBytesPerEntry * (size_t) n_waves);
  const unsigned long long cap_limit = 0xffffffffull / n_waves;  // all entries together < 2^32
  if (cap > cap_limit) cap = cap_limit;
  cap &= ~3ull;
  if (cap < kMinRegionEntries) return b;
  b.cap = (uint32_t) cap;
  b.wave_count = reinterpret_cast<uint32_t *>(p);
  p += round_up((size_t) n_waves * sizeof(uint32_t), 256);
  b.count = reinterpret_cast<uint32_t *>(p);
  p += round_up((size_t) n_waves * b.n_tiles * sizeof(uint32_t), 256);
  b.tile_base = reinterpret_cast<unsigned long long *>(p);
  p += round_up(((size_t) b.n_tiles + 1) * sizeof(unsigned long long), 256);
  b.stream = reinterpret_cast<uint32_t *>(p);
  p += (size_t) n_waves * b.cap * sizeof(uint32_t);
  p = (p + 255) & ~(uintptr_t) 255;
  b.sorted = reinterpret_cast<uint16_t *>(p);
  b.enabled = 1;
  return b;
}

hipError_t launch_binned_scatter(const BinLayout &b, unsigned long long *hist, int w, int h,
                                 hipStream_t stream) {
  if (!b.enabled) return hipSuccess;
  const size_t count_lds = (size_t) b.n_tiles * sizeof(uint32_t);
  hipLaunchKernelGGL(bin_count_kernel, dim3(b.n_waves), dim3(256), count_lds, stream, b);
  hipLaunchKernelGGL(bin_scan_waves_kernel, dim3((b.n_tiles + 255u) / 256u), dim3(256), 0, stream, b);
  hipLaunchKernelGGL(bin_scan_tiles_kernel, dim3(1), dim3(1024), 0, stream, b);
  const size_t scatter_lds = ((size_t) 3 * b.n_tiles + kScatterThreads + kChunkEntries) * sizeof(uint32_t) +
                             (size_t) kChunkEntries * sizeof(uint16_t);
  if (scatter_lds > 64 * 1024) {  // up to 98 KiB at 4096 tiles; gfx950 has 160 KiB per workgroup
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bin_scatter_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) scatter_lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(bin_scatter_kernel, dim3(b.n_waves), dim3(kScatterThreads), scatter_lds, stream, b);
  hipLaunchKernelGGL(bin_accumulate_kernel, dim3(b.n_tiles), dim3(512), 0, stream, b, hist, w, h);
  return hipGetLastError();
}

}  // namespace cb
