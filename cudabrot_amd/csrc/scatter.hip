// scatter.hip -- the deferred, tile-binned scatter (see kernels.h, BinLayout).
//
// Input: the pixel stream the REPLAY stage of draw_wave_kernel wrote -- per wave a region of packed
// (row << 16 | col) words.  Output: the same increments added to the u64 histogram, with one
// 64-byte memory-side request per 8 pixels of a tile slice instead of one per increment.
//
//   bin_count_kernel       one workgroup per wave region: LDS histogram over tiles -> count[tile][wave]
//   bin_scan_waves_kernel  one workgroup per tile: exclusive prefix over waves (in place), tile total
//   bin_scan_tiles_kernel  exclusive prefix over the tile totals -> tile_base[]; and over the number
//                          of accumulate slices per tile -> slice_base[]
//   bin_scatter_kernel     one workgroup per wave region, chunks of 8192 entries: rank inside
//                          (chunk, tile) by LDS atomics, sort the chunk in LDS, write each tile's run to
//                          its place in `sorted` as 14-bit in-tile offsets (consecutive lanes write
//                          consecutive places of a run)
//   bin_accumulate_kernel  one workgroup per slice (<= 65536 entries of one tile): LDS u32 histogram
//                          of the slice, then added to the u64 histogram with coalesced device-scope
//                          atomics.  Hot tiles are many slices, so the grid stays balanced.
//
// Everything is a counting sort: no global atomics before the final flush, and the bytes written are
// the same from run to run.  All of it is HBM-streaming work (4 + 4 + 4 + 2 + 2 = 16 bytes per
// increment end to end, plus 16 KiB of flush per slice) that the draw kernel's fp64 loop leaves idle.
#include <stdlib.h>

#include "kernels.h"

namespace cb {

namespace {

constexpr uint32_t kScatterThreads = 512;
constexpr uint32_t kChunkEntries = 8192;  // 16 per thread
constexpr uint32_t kPerThread = kChunkEntries / kScatterThreads;
constexpr uint32_t kSliceEntriesDefault = 262144;  // entries one accumulate workgroup takes: few enough
                                                    // flushes of the 64 KiB tile, many enough slices to balance
constexpr uint32_t kAccThreads = 512;

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

// Exclusive prefix of v over the workgroup's threads (in thread order) and the workgroup total.
// wave_totals: LDS scratch of blockDim/64 words; the caller separates two calls by a barrier.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *wave_totals,
                                                         uint32_t *total) {
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (uint32_t d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(inc, d, 64);
    if (lane >= d) inc += up;
  }
  if (lane == 63u) wave_totals[wid] = inc;
  __syncthreads();
  uint32_t before = 0, all = 0;
  const uint32_t n_waves = blockDim.x >> 6;
  for (uint32_t k = 0; k < n_waves; ++k) {
    const uint32_t wt = wave_totals[k];
    if (k < wid) before += wt;
    all += wt;
  }
  *total = all;
  return before + inc - v;
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
  // tile-major: the scan over waves reads each tile's row contiguously
  for (uint32_t t = threadIdx.x; t < b.n_tiles; t += blockDim.x) {
    b.count[(size_t) t * b.n_waves + wv] = lds[t];
  }
}

// count[t][w] <- sum of count[t][w'] for w' < w; tile_base[t] <- total of tile t (scanned next).
__global__ void __launch_bounds__(256) bin_scan_waves_kernel(BinLayout b) {
  __shared__ uint32_t wave_totals[4];
  uint32_t *row = b.count + (size_t) blockIdx.x * b.n_waves;
  const uint32_t per = (b.n_waves + blockDim.x - 1u) / blockDim.x;
  const uint32_t w0 = threadIdx.x * per;
  uint32_t sum = 0;
  for (uint32_t k = 0; k < per; ++k) {
    if (w0 + k < b.n_waves) sum += row[w0 + k];
  }
  uint32_t total = 0;
  uint32_t run = block_exclusive_scan(sum, wave_totals, &total);
  for (uint32_t k = 0; k < per; ++k) {
    if (w0 + k < b.n_waves) {
      const uint32_t c = row[w0 + k];
      row[w0 + k] = run;
      run += c;
    }
  }
  if (threadIdx.x == 0) b.tile_base[blockIdx.x] = total;
}

// tile_base[t] <- sum of totals of tiles < t (tile_base[n_tiles] <- grand total), and
// slice_base[t] <- number of accumulate slices of tiles < t.  One workgroup.
__global__ void __launch_bounds__(1024) bin_scan_tiles_kernel(BinLayout b) {
  __shared__ unsigned long long part[1024];
  __shared__ uint32_t wave_totals[16];
  constexpr uint32_t kPer = kMaxTiles / 1024;  // 4 tiles per thread
  unsigned long long v[kPer];
  unsigned long long sum = 0;
  uint32_t slices = 0;
  for (uint32_t k = 0; k < kPer; ++k) {
    const uint32_t t = threadIdx.x * kPer + k;
    v[k] = t < b.n_tiles ? b.tile_base[t] : 0ull;
    sum += v[k];
    slices += (uint32_t) ((v[k] + b.slice_entries - 1u) / b.slice_entries);
  }
  part[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan (64-bit)
    const unsigned long long add = threadIdx.x >= d ? part[threadIdx.x - d] : 0ull;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t slice_total = 0;
  uint32_t srun = block_exclusive_scan(slices, wave_totals, &slice_total);
  unsigned long long run = part[threadIdx.x] - sum;
  for (uint32_t k = 0; k < kPer; ++k) {
    const uint32_t t = threadIdx.x * kPer + k;
    if (t < b.n_tiles) {
      b.tile_base[t] = run;
      b.slice_base[t] = srun;
    }
    run += v[k];
    srun += (uint32_t) ((v[k] + b.slice_entries - 1u) / b.slice_entries);
  }
  if (threadIdx.x == 1023) {
    b.tile_base[b.n_tiles] = part[1023];
    b.slice_base[b.n_tiles] = slice_total;
  }
}

__global__ void __launch_bounds__(kScatterThreads) bin_scatter_kernel(BinLayout b) {
  extern __shared__ uint32_t lds[];
  // dynamic LDS: cursor[n_tiles] | cnt[n_tiles] | lstart[n_tiles] | wave_totals[8] | pos[chunk] | off[chunk]
  uint32_t *cursor = lds;
  uint32_t *cnt = cursor + b.n_tiles;
  uint32_t *lstart = cnt + b.n_tiles;
  uint32_t *wave_totals = lstart + b.n_tiles;
  uint32_t *pos = wave_totals + 8;
  uint16_t *off = reinterpret_cast<uint16_t *>(pos + kChunkEntries);

  const uint32_t wv = blockIdx.x;
  const uint32_t n = b.wave_count[wv];
  if (n == 0) return;
  const uint32_t *src = b.stream + (size_t) wv * b.cap;
  for (uint32_t t = threadIdx.x; t < b.n_tiles; t += kScatterThreads) {
    // all entries together are < 2^32, so 32-bit places suffice
    cursor[t] = (uint32_t) b.tile_base[t] + b.count[(size_t) t * b.n_waves + wv];
  }
  const uint32_t bins_per_thread = (b.n_tiles + kScatterThreads - 1u) / kScatterThreads;

  // the chunk after the current one is loaded while the current one is ranked and sorted
  uint32_t e_next[kPerThread];
#pragma unroll
  for (uint32_t k = 0; k < kPerThread; ++k) {
    const uint32_t i = k * kScatterThreads + threadIdx.x;
    e_next[k] = (i < n) ? src[i] : 0u;
  }
  for (uint32_t base = 0; base < n; base += kChunkEntries) {
    const uint32_t m = (n - base) < kChunkEntries ? (n - base) : kChunkEntries;
    for (uint32_t t = threadIdx.x; t < b.n_tiles; t += kScatterThreads) cnt[t] = 0u;
    // 1. rank of every entry inside (chunk, tile)
    uint32_t e[kPerThread], r[kPerThread];
#pragma unroll
    for (uint32_t k = 0; k < kPerThread; ++k) e[k] = e_next[k];
    {
      const uint32_t next = base + kChunkEntries;
#pragma unroll
      for (uint32_t k = 0; k < kPerThread; ++k) {
        const uint32_t i = next + k * kScatterThreads + threadIdx.x;
        e_next[k] = (i < n) ? src[i] : 0u;
      }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < kPerThread; ++k) {
      const uint32_t i = k * kScatterThreads + threadIdx.x;
      r[k] = 0u;
      if (i < m) {
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
        if (t0 + k < b.n_tiles) sum += cnt[t0 + k];
      }
      uint32_t total = 0;
      uint32_t run = block_exclusive_scan(sum, wave_totals, &total);
      for (uint32_t k = 0; k < bins_per_thread; ++k) {
        if (t0 + k < b.n_tiles) {
          lstart[t0 + k] = run;
          run += cnt[t0 + k];
        }
      }
    }
    __syncthreads();
    // 3. sort the chunk in LDS: destination place in `sorted` and in-tile offset, tile by tile
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

__global__ void __launch_bounds__(kAccThreads) bin_accumulate_kernel(BinLayout b,
                                                                     unsigned long long *hist,
                                                                     int w, int h) {
  __shared__ uint32_t tile[kTilePixels];  // 64 KiB
  // which (tile, slice) is this workgroup?  slice_base is an exclusive prefix: binary search
  const uint32_t s = blockIdx.x;
  if (s >= b.slice_base[b.n_tiles]) return;  // the grid is an upper bound
  uint32_t lo = 0, hi = b.n_tiles;  // invariant: slice_base[lo] <= s < slice_base[hi]
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (b.slice_base[mid] <= s) {
      lo = mid;
    } else {
      hi = mid;
    }
  }
  const uint32_t t = lo;
  const unsigned long long tile_begin = b.tile_base[t], tile_end = b.tile_base[t + 1];
  const unsigned long long begin = tile_begin + (unsigned long long) (s - b.slice_base[t]) * b.slice_entries;
  const unsigned long long end = (begin + b.slice_entries < tile_end) ? begin + b.slice_entries : tile_end;

  for (uint32_t p = threadIdx.x; p < (uint32_t) kTilePixels; p += kAccThreads) tile[p] = 0u;
  __syncthreads();
  // head up to a 16-byte boundary, then 8 entries per load, then the tail
  const uint16_t *src = b.sorted;
  unsigned long long body = (begin + 7ull) & ~7ull;
  if (body > end) body = end;
  for (unsigned long long i = begin + threadIdx.x; i < body; i += kAccThreads) lds_inc(&tile[src[i]]);
  const unsigned long long n8 = (end - body) >> 3;
  const uint4 *src8 = reinterpret_cast<const uint4 *>(src + body);
  for (unsigned long long i = threadIdx.x; i < n8; i += kAccThreads) {
    const uint4 v = src8[i];
    lds_inc(&tile[v.x & 0xffffu]);
    lds_inc(&tile[v.x >> 16]);
    lds_inc(&tile[v.y & 0xffffu]);
    lds_inc(&tile[v.y >> 16]);
    lds_inc(&tile[v.z & 0xffffu]);
    lds_inc(&tile[v.z >> 16]);
    lds_inc(&tile[v.w & 0xffffu]);
    lds_inc(&tile[v.w >> 16]);
  }
  for (unsigned long long i = body + (n8 << 3) + threadIdx.x; i < end; i += kAccThreads) {
    lds_inc(&tile[src[i]]);
  }
  __syncthreads();
  const uint32_t row0 = (t / b.tiles_x) << kTileShift;
  const uint32_t col0 = (t % b.tiles_x) << kTileShift;
  for (uint32_t p = threadIdx.x; p < (uint32_t) kTilePixels; p += kAccThreads) {
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
  return round_up((size_t) n_waves * sizeof(uint32_t), 256) +                   // wave_count
         round_up((size_t) n_waves * n_tiles * sizeof(uint32_t), 256) +         // count
         round_up(((size_t) n_tiles + 1) * sizeof(unsigned long long), 256) +   // tile_base
         round_up(((size_t) n_tiles + 1) * sizeof(uint32_t), 256) + 1024;       // slice_base + slack
}

BinLayout make_bin_layout(void *workspace, size_t bytes, int w, int h, uint32_t n_waves) {
  BinLayout b;
  b.enabled = 0;
  b.slice_entries = kSliceEntriesDefault;
  if (const char *e = getenv("CUDABROT_AMD_SLICE")) {  // tuning knob
    const long v = atol(e);
    if (v >= 4096 && v <= (1l << 30)) b.slice_entries = (uint32_t) v;
  }
  b.n_waves = n_waves;
  b.cap = 0;
  b.n_tiles = 0;
  b.tiles_x = 0;
  b.wave_count = nullptr;
  b.stream = nullptr;
  b.count = nullptr;
  b.tile_base = nullptr;
  b.slice_base = nullptr;
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
  b.slice_base = reinterpret_cast<uint32_t *>(p);
  p += round_up(((size_t) b.n_tiles + 1) * sizeof(uint32_t), 256);
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
  hipLaunchKernelGGL(bin_scan_waves_kernel, dim3(b.n_tiles), dim3(256), 0, stream, b);
  hipLaunchKernelGGL(bin_scan_tiles_kernel, dim3(1), dim3(1024), 0, stream, b);
  const size_t scatter_lds = ((size_t) 3 * b.n_tiles + 8 + kChunkEntries) * sizeof(uint32_t) +
                             (size_t) kChunkEntries * sizeof(uint16_t);
  if (scatter_lds > 64 * 1024) {  // up to 96 KiB at 4096 tiles; gfx950 has 160 KiB per workgroup
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bin_scatter_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) scatter_lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(bin_scatter_kernel, dim3(b.n_waves), dim3(kScatterThreads), scatter_lds, stream, b);
  // upper bound on the number of slices: one partial slice per tile + the full ones
  const unsigned long long max_entries = (unsigned long long) b.n_waves * b.cap;
  const uint32_t slices = b.n_tiles + (uint32_t) (max_entries / b.slice_entries) + 1u;
  hipLaunchKernelGGL(bin_accumulate_kernel, dim3(slices), dim3(kAccThreads), 0, stream, b, hist, w, h);
  return hipGetLastError();
}

}  // namespace cb
