// scatter.hip -- the deferred, tile-binned scatter (see kernels.h, BinLayout).
//
// Input: the pixel stream the REPLAY stage of draw_wave_kernel wrote -- per wave a region of packed
// (row << 16 | col) words.  Output: the same increments added to the u64 histogram, with one
// 64-byte memory-side request per 8 pixels of a tile slice instead of one per increment.
//
// One level (canvases of up to 4096 tiles of 128 x 128 pixels):
//   count      one workgroup per region (= wave region of the stream): LDS histogram over tiles
//              -> count[tile][region]
//   scan_rows  one workgroup per tile: exclusive prefix over its regions (in place), tile total
//   scan_keys  exclusive prefix over the tile totals -> tile_base[]; and over the number of
//              accumulate slices per tile -> slice_base[]
//   scatter    one workgroup per region, chunks of 8192 entries: rank inside (chunk, tile) by LDS
//              atomics, sort the chunk in LDS, write each tile's run to its place in `sorted` as
//              14-bit in-tile offsets (consecutive lanes write consecutive places of a run)
//   accumulate one workgroup per slice (a stretch of one tile's entries): LDS u32 histogram of the
//              slice, then added to the u64 histogram with coalesced device-scope atomics.  Hot
//              tiles are many slices, so the grid stays balanced.
//
// Two levels (up to 262144 tiles: 20000 x 20000 and beyond).  With more tiles than LDS counters, or
// runs too short to coalesce, the stream is first partitioned into GROUPS of 1024 consecutive
// tiles by the same count -> scan -> scatter (level A: the key is group * 16 + a lane-derived
// replica, which spreads the LDS atomics over 4 counters per group; whole 4-byte entries are
// moved).  The grouped stream is then cut into fixed-size regions and every region runs the
// one-level pipeline over the 1024 tiles of its group (level B).
//
// Everything is a counting sort: no global atomics before the final flush, and the bytes written are
// the same from run to run.  All of it is HBM-streaming work (4 + 4 + 4 + 2 + 2 = 16 bytes per
// increment end to end with one level, 28 with two, plus 16 KiB of flush per slice).
#include <stdlib.h>
#include <string.h>

#include "kernels.h"

namespace cb {

namespace {

constexpr uint32_t kScatterThreads = 512;
constexpr uint32_t kChunkEntries = 8192;  // 16 per thread
constexpr uint32_t kPerThread = kChunkEntries / kScatterThreads;
constexpr uint32_t kSliceEntriesDefault = 262144;  // entries one accumulate workgroup takes: few enough
                                                    // flushes of the 64 KiB tile, many enough slices to balance
constexpr uint32_t kAccThreads = 512;
constexpr uint32_t kGroupTiles = 1024;             // tiles per group (two levels)
constexpr uint32_t kGroupShift = 10;
constexpr uint32_t kReplicas = 4;                  // level-A keys per group (2..8 measured equal at 20000^2; 16: 5 % slower, 64: 15 %)
constexpr uint32_t kMaxGroups = 256;               // -> 262144 tiles (all planes of a fused render together)
constexpr uint32_t kRegionEntries = 262144;        // level-B region: 32 chunks

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// stream word -> tile (of the stack of planes), in-tile offset (BinLayout: e_* describe the word)
__device__ __forceinline__ uint32_t tile_of(uint32_t e, const BinLayout &b) {
  const uint32_t plane = (e >> b.e_chan_shift) & b.e_chan_mask;
  const uint32_t tile_row = plane * b.tiles_y + (((e >> b.e_row_shift) & b.e_row_mask) >> kTileShift);
  return tile_row * b.tiles_x + ((e & b.e_col_mask) >> kTileShift);
}
__device__ __forceinline__ uint32_t offset_of(uint32_t e, const BinLayout &b) {
  // a canvas narrower (lower) than a tile has a column (row) field of fewer bits than a tile coordinate
  return (((e >> b.e_row_shift) & b.e_row_mask & (kTileSize - 1u)) << kTileShift) |
         (e & b.e_col_mask & (kTileSize - 1u));
}
__device__ __forceinline__ void lds_inc(uint32_t *p) {
  __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_u32
}

// What a sorting pass works on.  kLevelA: wave regions of the stream -> keys (group, replica) ->
// `grouped`.  Otherwise (level B, or the only level): regions of the (grouped) stream -> tiles of the
// region's group -> `sorted`.
template <bool kLevelA>
struct Pass {
  __device__ static uint32_t n_regions(const BinLayout &b) { return kLevelA ? b.n_waves : *b.n_regions; }
  __device__ static const uint32_t *src(const BinLayout &b, uint32_t r) {
    if (kLevelA) return b.stream + (size_t) r * b.cap;
    return (b.two_level ? b.grouped : b.stream) + b.region_start[r];
  }
  __device__ static uint32_t count_of(const BinLayout &b, uint32_t r) {
    return kLevelA ? b.wave_count[r] : b.region_count[r];
  }
  __device__ static uint32_t n_keys(const BinLayout &b) {
    if (kLevelA) return b.n_groups * kReplicas;
    return b.two_level ? kGroupTiles : b.n_tiles;
  }
  // first global key of region r's key range, and the region's column in the count matrix
  __device__ static uint32_t key0(const BinLayout &b, uint32_t r) {
    return (kLevelA || !b.two_level) ? 0u : b.region_group[r] << kGroupShift;
  }
  __device__ static uint32_t column(const BinLayout &b, uint32_t r) {
    return (kLevelA || !b.two_level) ? r : b.region_index[r];
  }
  // j: the entry's index in its region.  The replica must be a function of the entry alone (both the
  // count and the scatter kernel evaluate it, with different thread mappings).
  __device__ static uint32_t key(const BinLayout &b, uint32_t e, uint32_t k0, uint32_t j) {
    const uint32_t t = tile_of(e, b);
    if (kLevelA) return ((t >> kGroupShift) * kReplicas) | (j & (kReplicas - 1u));
    return t - k0;
  }
  __device__ static uint32_t *counts(const BinLayout &b) { return kLevelA ? b.a_count : b.count; }
  __device__ static unsigned long long *bases(const BinLayout &b) { return kLevelA ? b.a_base : b.tile_base; }
  __device__ static uint32_t stride(const BinLayout &b) { return kLevelA ? b.n_waves : b.count_stride; }
  // keys of the whole pass and the row length of one key
  __device__ static uint32_t total_keys(const BinLayout &b) { return kLevelA ? b.n_groups * kReplicas : b.n_tiles; }
  __device__ static uint32_t row_length(const BinLayout &b, uint32_t key_global) {
    if (kLevelA) return b.n_waves;
    return b.two_level ? b.group_regions[key_global >> kGroupShift] : b.n_waves;
  }
  // the last group of a two-level layout may have fewer than kGroupTiles tiles
  __device__ static uint32_t keys_of_region(const BinLayout &b, uint32_t k0) {
    if (kLevelA || !b.two_level) return n_keys(b);
    const uint32_t left = b.n_tiles - k0;
    return left < kGroupTiles ? left : kGroupTiles;
  }
};

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

// The single-level region table: region r = wave r.
__global__ void __launch_bounds__(256) bin_wave_regions_kernel(BinLayout b) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < b.n_waves) {
    b.region_start[r] = (unsigned long long) r * b.cap;
    b.region_count[r] = b.wave_count[r];
  }
  if (r == 0) *b.n_regions = b.n_waves;
}

template <bool kLevelA>
__global__ void __launch_bounds__(kScatterThreads) bin_count_kernel(BinLayout b) {
  using P = Pass<kLevelA>;
  extern __shared__ uint32_t lds[];  // [n_keys]
  const uint32_t r = blockIdx.x;
  if (r >= P::n_regions(b)) return;  // the grid is an upper bound (level B)
  const uint32_t n = P::count_of(b, r);
  const uint32_t *src = P::src(b, r);
  const uint32_t k0 = P::key0(b, r), nk = P::keys_of_region(b, k0);
  for (uint32_t t = threadIdx.x; t < nk; t += blockDim.x) lds[t] = 0u;
  __syncthreads();
  // entries up to a 16-byte boundary, then four per load, then the tail (level-B regions start anywhere)
  uint32_t head = (4u - (uint32_t) ((reinterpret_cast<uintptr_t>(src) >> 2) & 3u)) & 3u;
  if (head > n) head = n;
  if (threadIdx.x < head) lds_inc(&lds[P::key(b, src[threadIdx.x], k0, threadIdx.x)]);
  const uint32_t n4 = (n - head) >> 2;
  const uint4 *src4 = reinterpret_cast<const uint4 *>(src + head);
  for (uint32_t i = threadIdx.x; i < n4; i += blockDim.x) {
    const uint4 v = src4[i];
    const uint32_t j = head + (i << 2);
    lds_inc(&lds[P::key(b, v.x, k0, j)]);
    lds_inc(&lds[P::key(b, v.y, k0, j + 1u)]);
    lds_inc(&lds[P::key(b, v.z, k0, j + 2u)]);
    lds_inc(&lds[P::key(b, v.w, k0, j + 3u)]);
  }
  for (uint32_t i = head + (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) {
    lds_inc(&lds[P::key(b, src[i], k0, i)]);
  }
  __syncthreads();
  // key-major: the scan over regions reads each key's row contiguously
  uint32_t *out = P::counts(b);
  const uint32_t col = P::column(b, r), stride = P::stride(b);
  for (uint32_t t = threadIdx.x; t < nk; t += blockDim.x) {
    out[(size_t) (k0 + t) * stride + col] = lds[t];
  }
}

// counts[key][c] <- sum of counts[key][c'] for c' < c; bases[key] <- total of the key (scanned next).
template <bool kLevelA>
__global__ void __launch_bounds__(256) bin_scan_rows_kernel(BinLayout b) {
  using P = Pass<kLevelA>;
  __shared__ uint32_t wave_totals[4];
  const uint32_t len = P::row_length(b, blockIdx.x);
  uint32_t *row = P::counts(b) + (size_t) blockIdx.x * P::stride(b);
  const uint32_t per = (len + blockDim.x - 1u) / blockDim.x;
  const uint32_t w0 = threadIdx.x * per;
  uint32_t sum = 0;
  for (uint32_t k = 0; k < per; ++k) {
    if (w0 + k < len) sum += row[w0 + k];
  }
  uint32_t total = 0;
  uint32_t run = block_exclusive_scan(sum, wave_totals, &total);
  for (uint32_t k = 0; k < per; ++k) {
    if (w0 + k < len) {
      const uint32_t c = row[w0 + k];
      row[w0 + k] = run;
      run += c;
    }
  }
  if (threadIdx.x == 0) P::bases(b)[blockIdx.x] = total;
}

// bases[key] <- sum of totals of keys < key (bases[n] <- grand total); for tiles also
// slice_base[t] <- number of accumulate slices of tiles < t.  One workgroup.
constexpr uint32_t kMaxKeys = kMaxGroups * kGroupTiles;
template <bool kLevelA>
__global__ void __launch_bounds__(1024) bin_scan_keys_kernel(BinLayout b) {
  using P = Pass<kLevelA>;
  __shared__ unsigned long long part[1024];
  __shared__ uint32_t wave_totals[16];
  unsigned long long *base = P::bases(b);
  const uint32_t nk = P::total_keys(b);
  const uint32_t per = (nk + 1023u) / 1024u;  // keys per thread, <= kMaxKeys / 1024
  const uint32_t t0 = threadIdx.x * per;
  unsigned long long sum = 0;
  uint32_t slices = 0;
  for (uint32_t k = 0; k < per; ++k) {
    const uint32_t t = t0 + k;
    const unsigned long long v = t < nk ? base[t] : 0ull;
    sum += v;
    slices += (uint32_t) ((v + b.slice_entries - 1u) / b.slice_entries);
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
  for (uint32_t k = 0; k < per; ++k) {
    const uint32_t t = t0 + k;
    if (t < nk) {
      const unsigned long long v = base[t];
      base[t] = run;
      if (!kLevelA) b.slice_base[t] = srun;
      run += v;
      srun += (uint32_t) ((v + b.slice_entries - 1u) / b.slice_entries);
    }
  }
  __syncthreads();
  if (threadIdx.x == 1023) {
    base[nk] = part[1023];
    if (!kLevelA) b.slice_base[nk] = slice_total;
  }
}

// Level-B region table from the group extents of `grouped` (a_base, scanned): group g's entries are
// cut into regions of kRegionEntries.  One workgroup, one thread per group.
__global__ void __launch_bounds__(kMaxGroups) bin_group_regions_kernel(BinLayout b) {
  __shared__ uint32_t first[kMaxGroups + 1];
  __shared__ uint32_t wave_totals[kMaxGroups / 64];
  const uint32_t g = threadIdx.x;
  unsigned long long begin = 0, end = 0;
  uint32_t mine = 0;
  if (g < b.n_groups) {
    begin = b.a_base[(size_t) g * kReplicas];
    end = b.a_base[(size_t) (g + 1) * kReplicas];
    mine = (uint32_t) ((end - begin + kRegionEntries - 1u) / kRegionEntries);
    b.group_regions[g] = mine;
  }
  uint32_t total = 0;
  first[g] = block_exclusive_scan(mine, wave_totals, &total);
  __syncthreads();
  if (g == 0) *b.n_regions = total;
  if (g < b.n_groups) {
    for (uint32_t k = 0; k < mine; ++k) {
      const uint32_t r = first[g] + k;
      const unsigned long long s = begin + (unsigned long long) k * kRegionEntries;
      b.region_start[r] = s;
      b.region_count[r] = (uint32_t) ((end - s) < kRegionEntries ? (end - s) : kRegionEntries);
      b.region_group[r] = g;
      b.region_index[r] = k;
    }
  }
}

template <bool kLevelA>
__global__ void __launch_bounds__(kScatterThreads) bin_scatter_kernel(BinLayout b) {
  using P = Pass<kLevelA>;
  extern __shared__ uint32_t lds[];
  // dynamic LDS: bounds[nk] {start of the key's run in the sorted chunk, its next place in the output} |
  // cnt[nk] | wave_totals[8] | sorted[chunk] {place in the output, payload}.  Pairs, so that sorting an
  // entry is one 8-byte read and one 8-byte write and copying it out one 8-byte read.
  const uint32_t nk_lds = P::n_keys(b);
  uint2 *bounds = reinterpret_cast<uint2 *>(lds);
  uint32_t *cnt = lds + 2u * nk_lds;
  uint32_t *wave_totals = cnt + nk_lds;
  uint2 *sorted = reinterpret_cast<uint2 *>(wave_totals + 8);

  const uint32_t r = blockIdx.x;
  if (r >= P::n_regions(b)) return;
  const uint32_t n = P::count_of(b, r);
  if (n == 0) return;
  const uint32_t *src = P::src(b, r);
  const uint32_t k0 = P::key0(b, r), nk = P::keys_of_region(b, k0);
  const uint32_t col = P::column(b, r), stride = P::stride(b);
  const uint32_t *counts = P::counts(b);
  const unsigned long long *bases = P::bases(b);
  for (uint32_t t = threadIdx.x; t < nk; t += kScatterThreads) {
    // all entries together are < 2^32, so 32-bit places suffice
    bounds[t].y = (uint32_t) bases[k0 + t] + counts[(size_t) (k0 + t) * stride + col];
  }
  const uint32_t bins_per_thread = (nk + kScatterThreads - 1u) / kScatterThreads;

  for (uint32_t base = 0; base < n; base += kChunkEntries) {
    const uint32_t m = (n - base) < kChunkEntries ? (n - base) : kChunkEntries;
    for (uint32_t t = threadIdx.x; t < nk; t += kScatterThreads) cnt[t] = 0u;
    __syncthreads();
    // 1. rank of every entry inside (chunk, key)
    uint32_t e[kPerThread], rank[kPerThread], key[kPerThread];
    uint32_t taken_in_chunk = m;
#pragma unroll
    for (uint32_t k = 0; k < kPerThread; ++k) {
      const uint32_t i = k * kScatterThreads + threadIdx.x;
      e[k] = (i < m) ? src[base + i] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < kPerThread; ++k) {
      const uint32_t i = k * kScatterThreads + threadIdx.x;
      rank[k] = 0u;
      key[k] = ~0u;  // beyond the chunk
      if (i < m) {
        key[k] = P::key(b, e[k], k0, base + i);
        rank[k] = __hip_atomic_fetch_add(&cnt[key[k]], 1u, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_rtn_u32
      }
    }
    __syncthreads();
    // 2. exclusive scan of cnt over keys -> start of each key's run in the sorted chunk
    {
      const uint32_t t0 = threadIdx.x * bins_per_thread;
      uint32_t sum = 0;
      for (uint32_t k = 0; k < bins_per_thread; ++k) {
        if (t0 + k < nk) sum += cnt[t0 + k];
      }
      uint32_t total = 0;
      uint32_t run = block_exclusive_scan(sum, wave_totals, &total);
      for (uint32_t k = 0; k < bins_per_thread; ++k) {
        if (t0 + k < nk) {
          bounds[t0 + k].x = run;
          run += cnt[t0 + k];
        }
      }
      taken_in_chunk = total;
    }
    __syncthreads();
    // 3. sort the chunk in LDS: destination place and payload, key by key
#pragma unroll
    for (uint32_t k = 0; k < kPerThread; ++k) {
      if (key[k] != ~0u) {
        const uint2 bd = bounds[key[k]];
        sorted[bd.x + rank[k]] = make_uint2(bd.y + rank[k], kLevelA ? e[k] : offset_of(e[k], b));
      }
    }
    __syncthreads();
    // 4. write the runs out (consecutive lanes -> consecutive places of a run) and advance the places
    if (kLevelA) {
      for (uint32_t i = threadIdx.x; i < taken_in_chunk; i += kScatterThreads) {
        const uint2 v = sorted[i];
        b.grouped[v.x] = v.y;
      }
    } else {
      for (uint32_t i = threadIdx.x; i < taken_in_chunk; i += kScatterThreads) {
        const uint2 v = sorted[i];
        b.sorted[v.x] = (uint16_t) v.y;
      }
    }
    for (uint32_t t = threadIdx.x; t < nk; t += kScatterThreads) bounds[t].y += cnt[t];
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
  const uint32_t plane_tiles = b.tiles_x * b.tiles_y;
  const uint32_t plane = t / plane_tiles, tt = t - plane * plane_tiles;
  const uint32_t row0 = (tt / b.tiles_x) << kTileShift;
  const uint32_t col0 = (tt % b.tiles_x) << kTileShift;
  hist += (unsigned long long) plane * b.plane_pixels;
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

// ---- workspace carving -----------------------------------------------------------------------------

struct Shape {
  bool ok;
  uint32_t tiles_x, n_tiles, two_level, n_groups, tiles_y;  // n_tiles: of all planes together
};

Shape shape_of(int w, int h, uint32_t planes) {
  Shape s = {false, 0, 0, 0, 0, 0};
  if (w <= 0 || h <= 0 || w > 65536 || h > 65536 || planes < 1u) return s;
  const uint32_t tiles_x = ((uint32_t) w + kTileSize - 1u) >> kTileShift;
  const uint32_t tiles_y = ((uint32_t) h + kTileSize - 1u) >> kTileShift;
  const unsigned long long n = (unsigned long long) tiles_x * tiles_y * planes;
  if (n > (unsigned long long) kMaxKeys) return s;
  s.ok = true;
  s.tiles_x = tiles_x;
  s.tiles_y = tiles_y;
  s.n_tiles = (uint32_t) n;
  s.two_level = n > kMaxTiles ? 1u : 0u;
  if (const char *e = getenv("CUDABROT_AMD_TWO_LEVEL")) {  // test knob: two levels on a small canvas
    if (atoi(e) != 0) s.two_level = 1u;
  }
  s.n_groups = s.two_level ? (s.n_tiles + kGroupTiles - 1u) / kGroupTiles : 0u;
  return s;
}

// Upper bound on the number of level-B regions for `entries` stream entries.
uint32_t max_regions_for(const Shape &s, uint32_t n_waves, unsigned long long entries) {
  if (!s.two_level) return n_waves;
  return (uint32_t) (entries / kRegionEntries) + s.n_groups + 1u;
}

// Bytes of everything but the per-entry buffers, for a region table of max_regions.
size_t fixed_bytes(const Shape &s, uint32_t n_waves, uint32_t max_regions) {
  size_t b = 0;
  b += round_up((size_t) n_waves * sizeof(uint32_t), 256);                          // wave_count
  b += round_up((size_t) max_regions * sizeof(unsigned long long), 256);            // region_start
  b += 3 * round_up((size_t) max_regions * sizeof(uint32_t), 256);                  // region_count/group/index
  b += round_up((size_t) (kMaxGroups + 1) * sizeof(uint32_t), 256);                 // group_regions
  b += 256;                                                                         // n_regions
  b += round_up((size_t) s.n_tiles * max_regions * sizeof(uint32_t), 256);          // count
  b += round_up(((size_t) s.n_tiles + 1) * sizeof(unsigned long long), 256);        // tile_base
  b += round_up(((size_t) s.n_tiles + 1) * sizeof(uint32_t), 256);                  // slice_base
  if (s.two_level) {
    const size_t keys = (size_t) s.n_groups * kReplicas;
    b += round_up(keys * n_waves * sizeof(uint32_t), 256);                          // a_count
    b += round_up((keys + 1) * sizeof(unsigned long long), 256);                    // a_base
  }
  return b + 1024;
}

size_t bytes_per_entry(const Shape &s) {
  return sizeof(uint32_t) + sizeof(uint16_t) + (s.two_level ? sizeof(uint32_t) : 0);
}

template <typename T>
T *carve(uintptr_t &p, size_t bytes) {
  T *r = reinterpret_cast<T *>(p);
  p += round_up(bytes, 256);
  return r;
}

}  // namespace

size_t bin_workspace_bytes(int w, int h, uint32_t n_waves, double entries_per_wave, int n_planes) {
  const Shape s = shape_of(w, h, (uint32_t) (n_planes > 0 ? n_planes : 1));
  if (!s.ok || n_waves == 0) return 0;
  if (entries_per_wave < 2.0 * kMinRegionEntries) entries_per_wave = 2.0 * kMinRegionEntries;
  const unsigned long long entries = (unsigned long long) (entries_per_wave * n_waves);
  return fixed_bytes(s, n_waves, max_regions_for(s, n_waves, entries)) + entries * bytes_per_entry(s) + 8192;
}

BinLayout make_bin_layout(void *workspace, size_t bytes, int w, int h, uint32_t n_waves, int n_channels) {
  BinLayout b;
  memset(&b, 0, sizeof(b));
  b.e_row_shift = 16;
  b.e_col_mask = 0xffffu;
  b.e_row_mask = 0xffffu;
  b.e_chan_shift = 0;
  b.e_chan_mask = 0;
  b.n_planes = n_channels > 0 ? (uint32_t) n_channels : 1u;
  b.plane_pixels = (unsigned long long) (w > 0 ? w : 0) * (unsigned long long) (h > 0 ? h : 0);
  if (n_channels > 0) {  // [channel | row | col], fields as narrow as the canvas allows
    uint32_t cb_ = 0, rb = 0, nb = 0;
    while ((1u << cb_) < (uint32_t) (w > 1 ? w : 1)) ++cb_;
    while ((1u << rb) < (uint32_t) (h > 1 ? h : 1)) ++rb;
    while ((1u << nb) < (uint32_t) n_channels) ++nb;
    // the draw kernel tags a word with channel << e_chan_shift whether or not the stream is used
    b.e_chan_shift = (cb_ + rb + nb <= 32u && cb_ + rb < 32u) ? cb_ + rb : 0u;
    b.e_chan_mask = (1u << nb) - 1u;
    if (w <= 0 || h <= 0 || cb_ + rb + nb > 32u) return b;  // no room: direct atomics
    b.e_row_shift = cb_;
    b.e_col_mask = (1u << cb_) - 1u;
    b.e_row_mask = (1u << rb) - 1u;
  }
  b.slice_entries = kSliceEntriesDefault;
  if (const char *e = getenv("CUDABROT_AMD_SLICE")) {  // tuning knob
    const long v = atol(e);
    if (v >= 4096 && v <= (1l << 30)) b.slice_entries = (uint32_t) v;
  }
  b.n_waves = n_waves;
  const Shape s = shape_of(w, h, b.n_planes);
  if (!workspace || n_waves == 0 || !s.ok) return b;
  b.n_tiles = s.n_tiles;
  b.tiles_x = s.tiles_x;
  b.tiles_y = s.tiles_y;
  b.two_level = s.two_level;
  b.n_groups = s.n_groups;
  // align the carve to 256 bytes
  uintptr_t p = reinterpret_cast<uintptr_t>(workspace);
  const uintptr_t p_end = p + bytes;
  p = (p + 255) & ~(uintptr_t) 255;
  if (p >= p_end) return b;
  // The region table (and with it the count matrix) is sized by the entries, which are sized by what
  // is left: shrink cap until everything fits.
  const size_t per_entry = bytes_per_entry(s);
  unsigned long long cap = (p_end - p) / (per_entry * (size_t) n_waves);
  const unsigned long long cap_limit = 0xffffffffull / n_waves;  // all entries together < 2^32
  if (cap > cap_limit) cap = cap_limit;
  for (;;) {
    cap &= ~3ull;
    if (cap < kMinRegionEntries) return b;
    const uint32_t mr = max_regions_for(s, n_waves, cap * n_waves);
    const size_t need = fixed_bytes(s, n_waves, mr) + (size_t) cap * n_waves * per_entry + 1024;
    if (p + need <= p_end) break;
    const size_t over = p + need - p_end;
    const unsigned long long cut = over / (per_entry * (size_t) n_waves) + 4;
    if (cut >= cap) return b;
    cap -= cut;
  }
  b.cap = (uint32_t) cap;
  b.max_regions = max_regions_for(s, n_waves, cap * n_waves);
  b.count_stride = b.max_regions;
  b.wave_count = carve<uint32_t>(p, (size_t) n_waves * sizeof(uint32_t));
  b.region_start = carve<unsigned long long>(p, (size_t) b.max_regions * sizeof(unsigned long long));
  b.region_count = carve<uint32_t>(p, (size_t) b.max_regions * sizeof(uint32_t));
  b.region_group = carve<uint32_t>(p, (size_t) b.max_regions * sizeof(uint32_t));
  b.region_index = carve<uint32_t>(p, (size_t) b.max_regions * sizeof(uint32_t));
  b.group_regions = carve<uint32_t>(p, (size_t) (kMaxGroups + 1) * sizeof(uint32_t));
  b.n_regions = carve<uint32_t>(p, 256);
  b.count = carve<uint32_t>(p, (size_t) b.n_tiles * b.max_regions * sizeof(uint32_t));
  b.tile_base = carve<unsigned long long>(p, ((size_t) b.n_tiles + 1) * sizeof(unsigned long long));
  b.slice_base = carve<uint32_t>(p, ((size_t) b.n_tiles + 1) * sizeof(uint32_t));
  if (b.two_level) {
    const size_t keys = (size_t) b.n_groups * kReplicas;
    b.a_count = carve<uint32_t>(p, keys * n_waves * sizeof(uint32_t));
    b.a_base = carve<unsigned long long>(p, (keys + 1) * sizeof(unsigned long long));
  }
  b.stream = carve<uint32_t>(p, (size_t) n_waves * b.cap * sizeof(uint32_t));
  if (b.two_level) b.grouped = carve<uint32_t>(p, (size_t) n_waves * b.cap * sizeof(uint32_t));
  b.sorted = carve<uint16_t>(p, (size_t) n_waves * b.cap * sizeof(uint16_t));
  b.enabled = 1;
  return b;
}

namespace {

template <bool kLevelA>
hipError_t launch_pass(const BinLayout &b, uint32_t n_keys_lds, uint32_t total_keys, uint32_t regions,
                       hipStream_t stream) {
  const size_t count_lds = (size_t) n_keys_lds * sizeof(uint32_t);
  hipLaunchKernelGGL((bin_count_kernel<kLevelA>), dim3(regions), dim3(kScatterThreads), count_lds, stream, b);
  hipLaunchKernelGGL((bin_scan_rows_kernel<kLevelA>), dim3(total_keys), dim3(256), 0, stream, b);
  hipLaunchKernelGGL((bin_scan_keys_kernel<kLevelA>), dim3(1), dim3(1024), 0, stream, b);
  if (kLevelA) hipLaunchKernelGGL(bin_group_regions_kernel, dim3(1), dim3(kMaxGroups), 0, stream, b);
  const size_t scatter_lds = ((size_t) 3 * n_keys_lds + 8 + 2 * kChunkEntries) * sizeof(uint32_t);
  if (scatter_lds > 64 * 1024) {  // 76 KiB at 1024 keys, 112 KiB at 4096 tiles; gfx950 has 160 KiB per workgroup
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bin_scatter_kernel<kLevelA>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) scatter_lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((bin_scatter_kernel<kLevelA>), dim3(regions), dim3(kScatterThreads), scatter_lds, stream, b);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_binned_scatter(const BinLayout &b, unsigned long long *hist, int w, int h,
                                 hipStream_t stream) {
  if (!b.enabled) return hipSuccess;
  hipError_t e;
  if (b.two_level) {
    // level-B rows of the count matrix are group_regions[g] long; columns beyond a group's regions are
    // neither written nor read
    e = launch_pass<true>(b, b.n_groups * kReplicas, b.n_groups * kReplicas, b.n_waves, stream);
    if (e != hipSuccess) return e;
    e = launch_pass<false>(b, kGroupTiles, b.n_tiles, b.max_regions, stream);
  } else {
    hipLaunchKernelGGL(bin_wave_regions_kernel, dim3((b.n_waves + 255u) / 256u), dim3(256), 0, stream, b);
    e = launch_pass<false>(b, b.n_tiles, b.n_tiles, b.n_waves, stream);
  }
  if (e != hipSuccess) return e;
  // upper bound on the number of slices: one partial slice per tile + the full ones
  const unsigned long long max_entries = (unsigned long long) b.n_waves * b.cap;
  const uint32_t slices = b.n_tiles + (uint32_t) (max_entries / b.slice_entries) + 1u;
  hipLaunchKernelGGL(bin_accumulate_kernel, dim3(slices), dim3(kAccThreads), 0, stream, b, hist, w, h);
  return hipGetLastError();
}

}  // namespace cb
