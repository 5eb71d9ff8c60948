// scatter.hip -- the deferred, tile-binned scatter (see kernels.h, BinLayout).
//
// Input: the pixel stream the REPLAY stage of draw_wave_kernel wrote -- per wave a segment of packed
// (row << 16 | col) words.  Output: the same increments added to the u64 histogram, with one
// 64-byte memory-side request per 8 pixels of a tile slice instead of one per increment.
//
// One level (canvases of up to 1024 tiles of 128 x 128 pixels; C2 / C3: exactly 1024):
//   regions     the stream is cut into regions of at most 32768 entries (a wave's segment = consecutive
//               regions); one small kernel makes the table
//   region_sort one workgroup per region: the region's entries are read ONCE (32 per thread, kept in
//               registers), counted per tile in LDS, the 1024 counts are scanned, the entries are ranked
//               with LDS atomics and written -- as 14-bit in-tile offsets -- into a sorted image of the
//               region in LDS, which goes out as one linear, fully coalesced block.  The only other thing
//               a region publishes is run_start[tile][region]: where each tile's run begins inside it.
//               No pass over the stream that only counts, no global prefix sums, no fragmentary writes
//               (the previous design placed every tile's entries contiguously in a global order: one pass
//               to count, two scans, and a scatter whose runs were 8 entries = 16 bytes long).
//   accumulate  one workgroup per (tile, slice of 4096 regions): each LANE walks one region's run of the
//               tile (~32 entries = 64 bytes, 16-byte loads), `ds_add_u32` into a 64 KiB LDS tile, then one
//               coalesced pass of device-scope atomics over the non-zero pixels.
//
// Two levels (up to 262144 tiles: 20000 x 20000 and beyond, or the planes of a fused render together).  A
// region sort has 1024 keys, so the stream is first partitioned into GROUPS of 1024 consecutive tiles by
// a counting sort over the wave segments (level A: count -> scan -> scatter of whole 4-byte words into
// `grouped`; the key is group * 4 + a replica derived from the entry's index, which spreads the LDS
// atomics).  Regions are then cut from each group's stretch of `grouped`, so every region holds tiles of
// one group, and the two kernels above run unchanged with the group's first tile as key 0.
//
// Two levels, CHUNKED (up to 64 groups: 20000 x 20000, the three planes of the 20000 x 15000 recipe): level A is
// done by the draw kernel as it writes -- a wave's segment is a set of chunks of 1024 words, each of one group
// (kernels.h, kChunkWords).  What is left of level A here is bookkeeping on CHUNKS: count them per (group, wave),
// scan, list them by group (chunk_count / group_scan_* / chunk_list), and cut regions of 32 chunks from each
// group's stretch of the list.  The region sort reads its entries through the list; `grouped` does not exist.
//
// Increments commute, so the histogram is the same whatever the order; nothing here depends on timing
// except the order of entries inside a run, which nothing reads.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "kernels.h"

// priority of the sort / gather waves (0..3; tools/gpu_wide_prio.sh sweeps it against draw_wide.hip's CB_WIDE_PRIO_*)
#ifndef CB_SCATTER_PRIO
#define CB_SCATTER_PRIO 0
#endif
// ... of the region sort while it reads its region / writes its image (few instructions, long waits: worth issuing
// early), while it ranks (most of its vector and LDS instructions), and of the gather
#ifndef CB_GATHER_MASKED_ADDS  // (0: never, 1: on canvases of more than 1024 tiles, 2: always -- bin_gather_accumulate_kernel)
#define CB_GATHER_MASKED_ADDS 1
#endif
#ifndef CB_GATHER_SKIP_BLOCKS
#define CB_GATHER_SKIP_BLOCKS 1
#endif
#ifndef CB_LEAN_BATCH
#define CB_LEAN_BATCH 4  // 16-byte loads a thread of the lean region sort keeps in flight
#endif
#ifndef CB_SORT_PRIO_IO
#define CB_SORT_PRIO_IO 3   // (swept beside the wide draw kernel's 2 / 1, tools/gpu_wide_prio.sh: -5..7 % per step against 0)
#endif
#ifndef CB_SORT_PRIO_RANK
#define CB_SORT_PRIO_RANK CB_SCATTER_PRIO
#endif
#ifndef CB_GATHER_PRIO
#define CB_GATHER_PRIO CB_SCATTER_PRIO
#endif

namespace cb {

namespace {

constexpr uint32_t kScatterThreads = 512;          // level A
constexpr uint32_t kChunkEntries = 8192;           // level A: 16 per thread
constexpr uint32_t kPerThread = kChunkEntries / kScatterThreads;
// threads of an accumulate workgroup: 1024 where a workgroup has thousands of runs to walk (one level: 0.1 ms of
// 1.5 at C3), 512 where a tile has few regions (two levels: 2.2 ms instead of 2.3 at 20000^2)
constexpr uint32_t kAccThreadsWide = 1024, kAccThreadsNarrow = 512;
constexpr uint32_t kGroupShift = 10;
static_assert((1u << kGroupShift) == kGroupTiles, "group = tile >> kGroupShift");
constexpr uint32_t kReplicas = kGroupReplicas;     // level-A keys per group (2..8 measured equal at 20000^2; 16: 5 % slower, 64: 15 %)
constexpr uint32_t kMaxGroups = 256;               // -> 262144 tiles (all planes of a fused render together)
constexpr uint32_t kMaxKeys = kMaxGroups * kGroupTiles;
constexpr uint32_t kRegionEntries = 32768;         // entries of one region sort: runs of ~32 entries per tile
constexpr uint32_t kSortThreads = 1024;
constexpr uint32_t kSortPerThread = kRegionEntries / kSortThreads;  // 32 entries in registers
constexpr uint32_t kSliceRegionsDefault = 8192;    // most regions one accumulate workgroup gathers from (slice_regions_for)
constexpr uint32_t kGroupRegionEntries = kRegionEntries - 8u;  // regions cut from a group's stretch (two levels)
static_assert(kRegionEntries <= 65535, "run_start holds 16-bit positions inside a region");

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
__device__ __forceinline__ uint32_t wave_count_of(const BinLayout &b, uint32_t w) {
  const uint32_t n = b.wave_count[w];  // written by the draw kernel; never beyond the segment
  return n < b.cap ? n : b.cap;
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

// The same for the region sort, whose 1024 threads run it once per region beside the draw kernel, where every vector
// instruction counts: the wave's scan with DPP adds (row shifts, then the two row broadcasts: 6 instructions instead of
// 6 LDS shuffles with their address arithmetic), and the 16 wave totals summed by each wave for itself, one per lane
// (instead of 16 reads, compares and selects in every thread).  blockDim.x == 1024; no total.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  // v_add with a DPP operand: lanes without a source (first lanes of a row; rows a broadcast does not reach) add 0
  v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ uint32_t sort_exclusive_scan(uint32_t v, uint32_t *wave_totals) {
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t inc = wave_inclusive_scan(v);
  if (lane == 63u) wave_totals[wid] = inc;
  __syncthreads();
  uint32_t wt = wave_totals[lane & 15u];
  wt = (lane < wid) ? wt : 0u;  // (wid <= 15: lanes of the first row only)
  // the sum of the first row, in its last lane
  wt += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wt, 0x111, 0x1, 0xf, false);
  wt += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wt, 0x112, 0x1, 0xf, false);
  wt += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wt, 0x114, 0x1, 0xf, false);
  wt += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wt, 0x118, 0x1, 0xf, false);
  const uint32_t before = (uint32_t) __builtin_amdgcn_readlane((int) wt, 15);
  return before + inc - v;
}

// ---- region tables ---------------------------------------------------------------------------------
//
// The stream is cut into regions of at most kRegionEntries entries.  One level: every wave's segment
// [w * cap, w * cap + count) on its own, all regions in group 0.  Two levels: every group's stretch of `grouped`
// (a_base, scanned) on its own, in pieces of kGroupRegionEntries (a little short of kRegionEntries: these
// regions start anywhere, and the sort reads from the 16-byte boundary below).  An OWNER is a wave or a group.
// bin_region_heads_kernel (one workgroup): owner_first[o] = first region of owner o (exclusive scan of the
// owners' region counts), n_regions, and the groups' extents.  bin_fill_regions_kernel: one thread per region.
__device__ __forceinline__ uint32_t owner_count(const BinLayout &b) { return b.two_level ? b.n_groups : b.n_waves; }
__device__ __forceinline__ void owner_extent(const BinLayout &b, uint32_t o, unsigned long long *begin,
                                             unsigned long long *entries, uint32_t *piece) {
  if (b.chunked) {  // in chunks: a group's stretch of chunk_list, regions of kRegionChunks chunks
    *begin = b.a_base[o];
    *entries = b.a_base[o + 1u] - *begin;
    *piece = kRegionChunks;
  } else if (b.two_level) {
    *begin = b.a_base[(size_t) o * kReplicas];
    *entries = b.a_base[(size_t) (o + 1u) * kReplicas] - *begin;
    *piece = kGroupRegionEntries;
  } else {
    *begin = (unsigned long long) o * b.cap;
    *entries = wave_count_of(b, o);
    *piece = kRegionEntries;
  }
}

__global__ void __launch_bounds__(1024) bin_region_heads_kernel(BinLayout b) {
  __shared__ uint32_t wave_totals[16];
  __shared__ unsigned long long all_entries;
  const uint32_t n_owners = owner_count(b);
  const uint32_t per = (n_owners + 1023u) / 1024u;
  const uint32_t o0 = threadIdx.x * per;
  if (threadIdx.x == 0) all_entries = 0ull;
  __syncthreads();
  uint32_t mine = 0;
  unsigned long long my_entries = 0;
  for (uint32_t k = 0; k < per; ++k) {
    if (o0 + k < n_owners) {
      unsigned long long begin, entries;
      uint32_t piece;
      owner_extent(b, o0 + k, &begin, &entries, &piece);
      mine += (uint32_t) ((entries + piece - 1u) / piece);
      my_entries += entries;
    }
  }
  if (my_entries) __hip_atomic_fetch_add(&all_entries, my_entries, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  uint32_t total = 0;
  uint32_t r = block_exclusive_scan(mine, wave_totals, &total);
  // (block_exclusive_scan has a barrier behind the atomics above)
  if (threadIdx.x == 0) {  // entries of the launch, for the slice size (n_regions[2..3] as one 64-bit word)
    // (chunked: chunks were counted -- taken as full, the figure only steers the slice size)
    *reinterpret_cast<unsigned long long *>(b.n_regions + 2) = b.chunked ? all_entries * kChunkWords : all_entries;
  }
  for (uint32_t k = 0; k < per; ++k) {
    const uint32_t o = o0 + k;
    if (o >= n_owners) break;
    unsigned long long begin, entries;
    uint32_t piece;
    owner_extent(b, o, &begin, &entries, &piece);
    const uint32_t regions = (uint32_t) ((entries + piece - 1u) / piece);
    b.owner_first[o] = r;
    if (b.two_level) {
      b.group_first[o] = r;
      b.group_regions[o] = regions;
    }
    r += regions;
  }
  if (threadIdx.x == 0) {
    b.owner_first[n_owners] = total;
    *b.n_regions = total;
    if (!b.two_level) {
      b.group_first[0] = 0u;
      b.group_regions[0] = total;
    }
  }
}

__global__ void __launch_bounds__(256) bin_fill_regions_kernel(BinLayout b) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= *b.n_regions) return;  // the grid is an upper bound
  uint32_t lo = 0, hi = owner_count(b);  // invariant: owner_first[lo] <= r < owner_first[hi]
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (b.owner_first[mid] <= r) {
      lo = mid;
    } else {
      hi = mid;
    }
  }
  unsigned long long begin, entries;
  uint32_t piece;
  owner_extent(b, lo, &begin, &entries, &piece);
  const unsigned long long offset = (unsigned long long) (r - b.owner_first[lo]) * piece;
  b.region_start[r] = begin + offset;
  b.region_count[r] = (uint32_t) ((entries - offset) < piece ? (entries - offset) : piece);
  b.region_group[r] = b.two_level ? lo : 0u;
}

// How many regions one accumulate workgroup gathers from.  A workgroup zeroes and flushes a 64 KiB tile, so it
// wants many regions; the GPU wants thousands of workgroups, and a canvas of few tiles (the reference's 1000 x
// 1000 default has 64) would otherwise give a few hundred.  pairs = (tile, region) pairs of the launch.
#ifndef CB_SLICE_TARGET
#define CB_SLICE_TARGET 12288
#endif
#ifndef CB_SLICE_TARGET_TWO_LEVEL
#define CB_SLICE_TARGET_TWO_LEVEL 6144
#endif
constexpr uint32_t kSliceTargetGroups = CB_SLICE_TARGET;
// ... on a canvas of more than 1024 tiles the tiles alone are thousands of workgroups, a tile sees few entries per
// slice, and what a further slice of a tile costs is a 64 KiB tile zeroed, scanned and flushed once more (C4: 24649
// tiles; 9.95 -> 9.45 ms per step with half the slices beyond one per tile, and no different with fewer still)
constexpr uint32_t kSliceTargetGroupsTwoLevel = CB_SLICE_TARGET_TWO_LEVEL;
static_assert(kSliceTargetGroupsTwoLevel <= kSliceTargetGroups, "the grid of the gather is sized by the larger one");
constexpr uint32_t kSliceRegionsMin = 512;  // one run per lane: fewer leave waves of the workgroup without work
constexpr uint32_t kSliceEntriesMin = 32768;  // ... and a workgroup should find a few entries per pixel of its tile
__device__ __forceinline__ uint32_t slice_regions_for(const BinLayout &b, unsigned long long pairs,
                                                      unsigned long long entries) {
  const unsigned long long target = b.two_level ? kSliceTargetGroupsTwoLevel : kSliceTargetGroups;
  unsigned long long s = (pairs + target - 1u) / target;
  if (s < kSliceRegionsMin) s = kSliceRegionsMin;
  // a thin stream (the launch that only drains: a partial region per wave): fewer, longer slices
  const unsigned long long by_entries = entries ? (kSliceEntriesMin * pairs + entries - 1u) / entries : b.slice_regions;
  if (s < by_entries) s = by_entries;
  if (s > b.slice_regions) s = b.slice_regions;
  return (uint32_t) s;
}

// slice_base[t] <- number of accumulate workgroups of tiles < t: tile t of group g takes
// ceil(group_regions[g] / S) of them, S = slice_regions_for(...) left in n_regions[1].  One workgroup.
__global__ void __launch_bounds__(1024) bin_slice_table_kernel(BinLayout b) {
  __shared__ uint32_t wave_totals[16];
  const uint32_t nk = b.n_tiles;
  const uint32_t per = (nk + 1023u) / 1024u;  // <= kMaxKeys / 1024
  const uint32_t t0 = threadIdx.x * per;
  unsigned long long pairs = 0;
  for (uint32_t g = 0; g < b.n_groups; ++g) {
    const uint32_t tiles = (nk - (g << kGroupShift)) < kGroupTiles ? (nk - (g << kGroupShift)) : kGroupTiles;
    pairs += (unsigned long long) tiles * b.group_regions[g];
  }
  const uint32_t S = slice_regions_for(b, pairs, *reinterpret_cast<const unsigned long long *>(b.n_regions + 2));
  if (threadIdx.x == 0) b.n_regions[1] = S;
  auto slices_of = [&](uint32_t t) { return (b.group_regions[t >> kGroupShift] + S - 1u) / S; };
  uint32_t mine = 0;
  for (uint32_t k = 0; k < per; ++k) {
    if (t0 + k < nk) mine += slices_of(t0 + k);
  }
  uint32_t total = 0;
  uint32_t run = block_exclusive_scan(mine, wave_totals, &total);
  for (uint32_t k = 0; k < per; ++k) {
    const uint32_t t = t0 + k;
    if (t < nk) {
      b.slice_base[t] = run;
      run += slices_of(t);
    }
  }
  if (threadIdx.x == 0) b.slice_base[nk] = total;
}

// ---- region sort -----------------------------------------------------------------------------------
//
// LDS: cnt[1024 + 16] | wave_totals[16] | image[kRegionEntries + 16] (u16).  The image is kept congruent to
// its place in `sorted` modulo 8 entries (16 bytes), so that it leaves with 16-byte stores wherever the region
// starts (regions of a group start anywhere).  The per-entry steps are branch-free: the slots of a thread that
// lie beyond the region count into cnt[1024] and are placed at image[kRegionEntries + 8], which nothing reads
// (a branch per entry costs the compiler its scalar registers, and the kernel its second workgroup per CU).
// The counters exist kCntReplicas times (a lane uses replica lane % kCntReplicas; a tile's run is the replicas'
// stretches one after another): the stream is far from uniform over the tiles -- at C3 it spreads like 162
// equally likely tiles, not 1024 -- and LDS atomics of one wave instruction on one address are served one
// after another.
constexpr uint32_t kDummyPlace = kRegionEntries + 8u;
// Two replicas of 1024 counters, or -- canvases of at most 256 tiles, where the atomics crowd on few counters
// (the reference's default 1000 x 1000 has 64) -- eight replicas of 256.
template <bool kFewTiles>
struct SortLds {
#ifndef CB_CNT_REPLICAS
#define CB_CNT_REPLICAS 2
#endif
  static constexpr uint32_t kReplicas = kFewTiles ? 8u : CB_CNT_REPLICAS;
  static constexpr uint32_t kStride = (kFewTiles ? 256u : kGroupTiles) + 16u;  // counters of a replica + the dummy
  static constexpr size_t kBytes =  // counters | wave_totals[16] | chunks of the region[32] {first word, words} | image
      (kReplicas * kStride + 16 + 2 * kRegionChunks + 4) * sizeof(uint32_t) + (kRegionEntries + 16) * sizeof(uint16_t);
};
constexpr uint32_t kFewTilesMax = 256;
#ifndef CB_SORT_GRID
#define CB_SORT_GRID 32768
#endif
constexpr uint32_t kSortGrid = CB_SORT_GRID;  // workgroups of the region sort (each takes every kSortGrid-th region)
// The lean instance takes kRunBatch CONSECUTIVE regions at a time and publishes a tile's run starts in them as ONE
// 16-byte store: run_start is [tile][region], a region's 1024 run starts lie a row apart, and as 2-byte stores each
// costs the fabric a sector of its own (C3: 44 MB of table, 1.2 GB written; C4: 87 MB, 2.3 GB).
#ifndef CB_RUN_BATCH
#define CB_RUN_BATCH 8
#endif
constexpr uint32_t kRunBatch = CB_RUN_BATCH;  // 8 (16-byte stores), or 1: a store of 2 bytes per tile and region

// 16 bytes of the stream, which the sort reads once (CB_SORT_NT_LOADS: as a non-temporal load)
#ifndef CB_SORT_NT_LOADS
#define CB_SORT_NT_LOADS 0
#endif
__device__ __forceinline__ uint4 stream_load16(const uint4 *p) {
#if CB_SORT_NT_LOADS
  typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));
  const U32x4 v = __builtin_nontemporal_load(reinterpret_cast<const U32x4 *>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}

// kPlain: one plane and the plain word row << 16 | col (every render that is not a fused multi-channel one):
// tile and offset with constant shifts instead of the layout's run-time fields.
template <bool kPlain>
__device__ __forceinline__ uint32_t sort_word(uint32_t e, uint32_t k0, const BinLayout &b) {
  if (kPlain) {
    const uint32_t key = __umul24(e >> (16 + kTileShift), b.tiles_x) + ((e & 0xffffu) >> kTileShift) - k0;
    return (key << 16) | ((e >> (16 - kTileShift)) & ((kTileSize - 1u) << kTileShift)) | (e & (kTileSize - 1u));
  }
  return ((tile_of(e, b) - k0) << 16) | offset_of(e, b);
}

// kChunked (BinLayout::chunked): the region is a list of at most kRegionChunks chunks of the stream (region_start:
// first entry of chunk_list, region_count: chunks) instead of a stretch of it; its image goes to sorted[r * 32768]
// and region_count[r] becomes its number of entries, which is what the gather reads.
// kLean: the instance for the FULL regions of a plain stream on a 16-byte boundary (see below); the other instance of
// such a launch then takes what is left (a wave's last, partial region).
template <bool kPlain, bool kFewTiles, bool kChunked = false, bool kLean = false>
__global__ void __launch_bounds__(kSortThreads, 8) bin_region_sort_kernel(BinLayout b, uint32_t skip_lean) {
  constexpr uint32_t kCntReplicas = SortLds<kFewTiles>::kReplicas, kCntStride = SortLds<kFewTiles>::kStride;
  constexpr uint32_t kDummyKey = kCntStride - 16u;  // the counter behind a replica's real ones
  // Beside the two-waves-per-SIMD draw kernel (draw_wide.hip) these waves share their SIMDs with draw waves that
  // never wait for memory: the scatter issues little and waits a lot, so it goes first when it can issue.
  __builtin_amdgcn_s_setprio(CB_SORT_PRIO_IO);
  extern __shared__ uint32_t lds[];
  uint32_t *cnt = lds + (threadIdx.x % kCntReplicas) * kCntStride;  // this lane's replica
  uint32_t *wave_totals = lds + kCntReplicas * kCntStride;
  uint2 *chunks = reinterpret_cast<uint2 *>(lds + kCntReplicas * kCntStride + 16);
  uint16_t *image = reinterpret_cast<uint16_t *>(lds + kCntReplicas * kCntStride + 16 + 2 * kRegionChunks + 4);  // (+ the words of the region)

  // The grid is kSortGrid workgroups striding over the regions: how many regions a launch has is only known on
  // the device, and a grid sized for the most it could have spends 11 ns on every workgroup that finds nothing
  // to do (0.25 ms of a steady launch, 0.7 ms of a drain launch's thin stream).  (A grid of just the 512
  // resident workgroups was 12 % slower: the two workgroups of a CU fall into step.)
  const uint32_t n_regions = *b.n_regions;
  constexpr uint32_t kBatchRegions = kLean ? kRunBatch : 1u;
  for (uint32_t r_batch = blockIdx.x * kBatchRegions; r_batch < n_regions; r_batch += gridDim.x * kBatchRegions) {
  // kLean: this thread's tile's run starts in the regions of the batch, the first one in the lowest half-word (a
  // region the batch or this instance does not have: 0 -- the general instance, launched after this one, writes its own)
  uint32_t runs[4] = {0u, 0u, 0u, 0u};
  const auto push_run = [&](uint32_t v) {
    runs[0] = __builtin_amdgcn_alignbit(runs[1], runs[0], 16);
    runs[1] = __builtin_amdgcn_alignbit(runs[2], runs[1], 16);
    runs[2] = __builtin_amdgcn_alignbit(runs[3], runs[2], 16);
    runs[3] = __builtin_amdgcn_alignbit(v, runs[3], 16);
  };
#pragma unroll 1
  for (uint32_t r = r_batch; r < r_batch + kBatchRegions; ++r) {
  if (kLean && r >= n_regions) {
    push_run(0u);
    continue;
  }
  uint32_t n = kChunked ? 0u : b.region_count[r];
  const unsigned long long start = kChunked ? (unsigned long long) r * kRegionEntries : b.region_start[r];
  const uint32_t *src = kChunked ? b.stream : (b.two_level ? b.grouped : b.stream) + start;
  const uint32_t k0 = b.region_group[r] << kGroupShift;
  const uint32_t nk = (b.n_tiles - k0) < kGroupTiles ? (b.n_tiles - k0) : kGroupTiles;
  __syncthreads();  // (a further region of this workgroup: the previous image has left)
  for (uint32_t t = threadIdx.x; t < kCntReplicas * kCntStride; t += kSortThreads) lds[t] = 0u;
  if (kChunked && threadIdx.x < 64u) {  // the first wave: the region's chunks, and the sum of their words
    const uint32_t n_chunks = b.region_count[r];
    // a chunk the region does not have: no words (its loads read the start of the stream and are ignored)
    const uint2 c = threadIdx.x < n_chunks && threadIdx.x < kRegionChunks ? b.chunk_list[b.region_start[r] + threadIdx.x]
                                                                          : make_uint2(0u, 0u);
    if (threadIdx.x < kRegionChunks) chunks[threadIdx.x] = c;
    uint32_t words = c.y;
#pragma unroll
    for (uint32_t d = 32; d >= 1; d >>= 1) words += __shfl_xor(words, d, 64);
    if (threadIdx.x == 0) chunks[kRegionChunks].x = words;
  }
  __syncthreads();
  if (kChunked) n = chunks[kRegionChunks].x;

  // A FULL region of a plain stream on a 16-byte boundary (ten of a wave's eleven at C3): no masks, no dummy key, and
  // every entry carries the LDS byte address of its tile's counter instead of the key, the counters count bytes of the
  // image, so that the ranking pass is ONE vector instruction per entry (the address out of the packed word).  10
  // vector instructions per entry all told instead of 36: alone on the GPU the sort is memory-bound either way, beside
  // the two-waves-per-SIMD draw kernel its vector instructions are what it costs.
  const bool lean_region = kPlain && n != 0u && (kChunked || (start & 7ull) == 0ull);
  if (!kLean && skip_lean != 0u && lean_region) continue;  // (the lean instance's)
  if (kLean && !lean_region) {
    push_run(0u);
    continue;
  }
  if constexpr (kLean) {
    // LDS byte addresses as integers (the counters of this lane's replica, less the group's first key; the image), so
    // that an entry's counter is ONE shift-and-add away from its key and the image takes positions as they are
    typedef __attribute__((address_space(3))) uint32_t *LdsWord;
    typedef __attribute__((address_space(3))) uint16_t *LdsHalf;
    const uint32_t lds0 = (uint32_t) reinterpret_cast<uintptr_t>(static_cast<void *>(lds));  // (a flat LDS address: the offset is its low word)
    const uint32_t cnt_at = lds0 + (uint32_t) ((threadIdx.x % kCntReplicas) * kCntStride * sizeof(uint32_t)) - (k0 << 2);
    const uint32_t image_at = (uint32_t) reinterpret_cast<uintptr_t>(static_cast<void *>(image));
    const uint4 *src4 = reinterpret_cast<const uint4 *>(src);
    const uint32_t tiles_x = b.tiles_x;
    constexpr uint32_t kBatch = kChunked ? 2u : CB_LEAN_BATCH;  // 16-byte loads in flight per thread (chunked: 64-bit addresses)
    // kMasked: a region that is not full (a wave's last): loads clamped to the region, the slots beyond it count into the
    // replica's dummy counter and are not placed -- two or three more instructions per entry, on a tenth of them
    const auto sort_region = [&](auto masked) {
    constexpr bool kMasked = decltype(masked)::value;
    const uint32_t dummy_at = lds0 + (uint32_t) (((threadIdx.x % kCntReplicas) * kCntStride + kDummyKey) * sizeof(uint32_t));
    const uint32_t last4 = (n - 1u) >> 2;
    const uint32_t slot0 = 4u * threadIdx.x;  // this thread's slots: 4096 G + slot0 + q, G = 0..7
    // kChunked: slot group G is the region's chunks 4 G .. 4 G + 3, this thread's 16 bytes the in_chunk-th of chunk
    // 4 G + my_chunk (the same chunk for the whole wave); a slot holds an entry if it lies below the chunk's words
    const uint32_t my_chunk = threadIdx.x >> 8, in_chunk = threadIdx.x & 255u;
    const uint32_t n_chunks = kChunked ? __builtin_amdgcn_readfirstlane(b.region_count[r]) : 0u;
    const auto beyond = [&](uint32_t G) {  // (wave-uniform) nothing of slot group G lies inside the region
      return kChunked ? 4u * G >= n_chunks : (kMasked && G * 4u * kSortThreads >= n);
    };
    uint32_t e[kSortPerThread];
    // 1. the entries, once: e = counter address << 16 | in-tile offset (7 vector instructions per entry); the counters
    //    count BYTES of the image (2 per entry), so that the ranking below gets an entry's place as an address
#pragma unroll
    for (uint32_t part = 0; part < kSortPerThread / 4u / kBatch; ++part) {
      uint4 v[kBatch];
      uint32_t words_g[kBatch];
#pragma unroll
      for (uint32_t j = 0; j < kBatch; ++j) {
        const uint32_t i4 = (part * kBatch + j) * kSortThreads + threadIdx.x;
        if (kChunked) {  // (a chunk the region does not have: {0, 0} -- the start of the stream, no words)
          const uint2 cd = chunks[4u * (part * kBatch + j) + my_chunk];
          words_g[j] = cd.y;
          v[j] = stream_load16(&src4[(size_t) (cd.x >> 2) + in_chunk]);
        } else {
          v[j] = stream_load16(&src4[kMasked ? (i4 < last4 ? i4 : last4) : i4]);
        }
      }
#pragma unroll
      for (uint32_t j = 0; j < kBatch; ++j) {
        // (the slots 4096 G .. 4096 G + 4095 of group G = part * kBatch + j: none inside the region -- nothing to count)
        if (beyond(part * kBatch + j)) break;
        asm volatile("" : "+v"(v[j].x), "+v"(v[j].y), "+v"(v[j].z), "+v"(v[j].w));
        const uint32_t words[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
          const uint32_t w = words[q];  // row << 16 | col; not chunked: col < 16384 (the host's condition for the instance)
          const uint32_t key = __umul24(w >> (16 + kTileShift), tiles_x) + __builtin_amdgcn_ubfe(w, kTileShift, 16 - kTileShift);
          uint32_t addr = cnt_at + (key << 2);  // < 2^16: at most 8 replicas of 272 or 2 of 1040 counters
          if (kChunked ? !(4u * in_chunk + q < words_g[j])
                       : (kMasked && !((part * kBatch + j) * 4u * kSortThreads + q + slot0 < n))) {
            addr = dummy_at;
          }
          __hip_atomic_fetch_add((LdsWord) (uintptr_t) addr, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          // bits 7..13 from the row, the rest as it is: bits 0..6 are the column's, 14 and 15 are 0, and what lies
          // above is cut off by the byte select that joins address and offset
          const uint32_t off = ((w >> (16 - kTileShift)) & ((kTileSize - 1u) << kTileShift)) |
                               (w & (kChunked ? kTileSize - 1u : ~((kTileSize - 1u) << kTileShift)));
          uint32_t packed = __builtin_amdgcn_perm(addr, off, 0x05040100u);  // addr << 16 | off & 0xffff
          asm volatile("" : "+v"(packed));  // formed HERE (else address and offset are kept apart until the ranking pass)
          e[4u * (part * kBatch + j) + q] = packed;
        }
      }
      asm volatile("" ::: "memory");  // (one batch after the other: the registers of a batch are free for the next)
    }
    __syncthreads();
    {  // 2. where each tile's run starts (byte positions; as below)
      const bool has_key = threadIdx.x < kDummyKey;
      uint32_t c[kCntReplicas], sum = 0;
#pragma unroll
      for (uint32_t k = 0; k < kCntReplicas; ++k) {
        c[k] = has_key ? lds[k * kCntStride + threadIdx.x] : 0u;
        sum += c[k];
      }
      uint32_t first = sort_exclusive_scan(sum, wave_totals);
      __syncthreads();
      if (kRunBatch == 8u) {
        push_run(first >> 1);  // (a thread beyond the group's tiles: the region's entries, which nobody reads)
      } else if (threadIdx.x < nk) {
        b.run_start[(size_t) threadIdx.x * b.max_regions + r] = (uint16_t) (first >> 1);
      }
#pragma unroll
      for (uint32_t k = 0; k < kCntReplicas; ++k) {
        if (has_key) lds[k * kCntStride + threadIdx.x] = image_at + first;
        first += c[k];
      }
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(CB_SORT_PRIO_RANK);
#pragma unroll
    for (uint32_t g = 0; g < kSortPerThread / 8u; ++g) {  // 3. rank and place, eight entries' atomics in flight
      if (beyond(2u * g)) break;  // (groups 2 g and 2 g + 1: beyond the region)
      uint32_t pos[8];
      const uint32_t words_a = kChunked ? chunks[4u * (2u * g) + my_chunk].y : 0u;
      const uint32_t words_b = kChunked ? chunks[4u * (2u * g + 1u) + my_chunk].y : 0u;
#pragma unroll
      for (uint32_t k = 0; k < 8u; ++k) {
        if (k == 4u && beyond(2u * g + 1u)) break;
        pos[k] = __hip_atomic_fetch_add((LdsWord) (uintptr_t) (e[8u * g + k] >> 16), 2u,
                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_rtn_u32
      }
#pragma unroll
      for (uint32_t k = 0; k < 8u; ++k) {
        // (slot of entry 8 g + k: group G = 2 g + k / 4, q = k % 4)
        const bool inside = kChunked ? 4u * in_chunk + (k & 3u) < (k < 4u ? words_a : words_b)
                                     : (!kMasked || (2u * g + k / 4u) * 4u * kSortThreads + (k & 3u) + slot0 < n);
        if (inside) *((LdsHalf) (uintptr_t) pos[k]) = (uint16_t) e[8u * g + k];
      }
      asm volatile("" ::: "memory");
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(CB_SORT_PRIO_IO);
    {  // 4. the image leaves as one linear block
      const uint4 *s4 = reinterpret_cast<const uint4 *>(image);
      uint4 *d4 = reinterpret_cast<uint4 *>(b.sorted + start);
      const uint32_t n8 = kMasked ? n >> 3 : kRegionEntries / 8u;
      for (uint32_t i = threadIdx.x; i < n8; i += kSortThreads) d4[i] = s4[i];
      if (kMasked && threadIdx.x < (n & 7u)) b.sorted[start + 8u * n8 + threadIdx.x] = image[8u * n8 + threadIdx.x];
      if (kChunked && threadIdx.x == 0) b.region_count[r] = n;  // from chunks to entries (every region is sorted once)
    }
    };  // sort_region
    if (!kChunked && n == kRegionEntries) {
      sort_region(std::false_type());
    } else {
      sort_region(std::true_type());
    }
  } else {
  // 1. the region's entries, once: e[k] = (key << 16) | in-tile offset, ~0 beyond the region; counts per tile.
  // Which thread takes which entry does not matter: 16-byte loads from the 16-byte boundary below the region
  // (a region of a group starts anywhere; `head` entries before it are masked, and regions that may start
  // off a boundary are cut 8 entries short so that head + n still fits the 32 per thread).
  uint32_t e[kSortPerThread];
  if constexpr (kChunked) {
    // 16-byte load i4 of the region = load i4 % 256 of its chunk i4 / 256 (a chunk: 1024 words on a 4 KiB boundary)
    static_assert(kChunkWords == 1024 && kSortThreads == 1024, "a load instruction of the workgroup covers four chunks");
    const uint4 *stream4 = reinterpret_cast<const uint4 *>(b.stream);
    // batches of two loads (the addresses are 64-bit here -- a chunk may lie anywhere in 16 GiB -- and four in
    // flight with their addresses cost the second workgroup of a CU)
#pragma unroll
    for (uint32_t pair = 0; pair < kSortPerThread / 8u; ++pair) {
      uint4 v[2];
#pragma unroll
      for (uint32_t j = 0; j < 2; ++j) {
        const uint32_t i4 = (pair * 2u + j) * kSortThreads + threadIdx.x;
        v[j] = stream4[(size_t) (chunks[i4 >> 8].x >> 2) + (i4 & 255u)];
      }
#pragma unroll
      for (uint32_t j = 0; j < 2; ++j) {
        asm volatile("" : "+v"(v[j].x), "+v"(v[j].y), "+v"(v[j].z), "+v"(v[j].w));
        const uint32_t k = pair * 2u + j;
        const uint32_t words = chunks[(k * kSortThreads + threadIdx.x) >> 8].y;
        const uint32_t i = (threadIdx.x & 255u) * 4u;  // index of the load's first word inside its chunk
        const uint32_t w0 = sort_word<kPlain>(v[j].x, k0, b), w1 = sort_word<kPlain>(v[j].y, k0, b);
        const uint32_t w2 = sort_word<kPlain>(v[j].z, k0, b), w3 = sort_word<kPlain>(v[j].w, k0, b);
        e[4 * k + 0] = (i < words) ? w0 : ~0u;
        e[4 * k + 1] = (i + 1u < words) ? w1 : ~0u;
        e[4 * k + 2] = (i + 2u < words) ? w2 : ~0u;
        e[4 * k + 3] = (i + 3u < words) ? w3 : ~0u;
      }
#pragma unroll
      for (uint32_t k = pair * 8u; k < (pair + 1u) * 8u; ++k) {
        const uint32_t key = e[k] >> 16;  // 0xffff beyond the chunk's words
        lds_inc(&cnt[key < kDummyKey ? key : kDummyKey]);
      }
    }
  } else {
    const uint32_t head = (uint32_t) (start & 3ull);
    const uint4 *src4 = reinterpret_cast<const uint4 *>(src - head);
    const uint32_t lim = head + n;
    const uint32_t last4 = lim ? (lim - 1u) >> 2 : 0u;  // loads are unconditional (index clamped): a load under a
                                             // branch is issued and waited for alone
    // two batches of four loads: eight in flight at once cost more registers than the second workgroup of
    // a CU is worth
#pragma unroll
    for (uint32_t half = 0; half < 2; ++half) {
      uint4 v[kSortPerThread / 8u];
#pragma unroll
      for (uint32_t j = 0; j < kSortPerThread / 8u; ++j) {
        const uint32_t i4 = (half * (kSortPerThread / 8u) + j) * kSortThreads + threadIdx.x;
        v[j] = src4[i4 < last4 ? i4 : last4];  // the last one may read up to three words past the region: inside the buffer
      }
#pragma unroll
      for (uint32_t j = 0; j < kSortPerThread / 8u; ++j) {
        // the four words as they are, whatever the masks below say (else the compiler splits the load and
        // hides its parts behind per-entry branches)
        asm volatile("" : "+v"(v[j].x), "+v"(v[j].y), "+v"(v[j].z), "+v"(v[j].w));
        const uint32_t k = half * (kSortPerThread / 8u) + j;
        const uint32_t i = (k * kSortThreads + threadIdx.x) * 4u;
        const uint32_t w0 = sort_word<kPlain>(v[j].x, k0, b), w1 = sort_word<kPlain>(v[j].y, k0, b);
        const uint32_t w2 = sort_word<kPlain>(v[j].z, k0, b), w3 = sort_word<kPlain>(v[j].w, k0, b);
        e[4 * k + 0] = (i >= head && i < lim) ? w0 : ~0u;
        e[4 * k + 1] = (i + 1u >= head && i + 1u < lim) ? w1 : ~0u;
        e[4 * k + 2] = (i + 2u >= head && i + 2u < lim) ? w2 : ~0u;
        e[4 * k + 3] = (i + 3u < lim) ? w3 : ~0u;
      }
#pragma unroll
      for (uint32_t k = half * (kSortPerThread / 2u); k < (half + 1u) * (kSortPerThread / 2u); ++k) {
        const uint32_t key = e[k] >> 16;  // 0xffff beyond the region
        lds_inc(&cnt[key < kDummyKey ? key : kDummyKey]);
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_s_setprio(CB_SORT_PRIO_RANK);
  // 2. where each tile's run starts: exclusive scan of the counts; published as run_start[tile][region]
  {
    const bool has_key = threadIdx.x < kDummyKey;  // (with few tiles most threads only take part in the scan)
    uint32_t c[kCntReplicas], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < kCntReplicas; ++k) {
      c[k] = has_key ? lds[k * kCntStride + threadIdx.x] : 0u;
      sum += c[k];
    }
    uint32_t first = sort_exclusive_scan(sum, wave_totals);
    __syncthreads();  // every count has been read
    if (threadIdx.x < nk) b.run_start[(size_t) threadIdx.x * b.max_regions + r] = (uint16_t) first;
#pragma unroll
    for (uint32_t k = 0; k < kCntReplicas; ++k) {
      if (has_key) lds[k * kCntStride + threadIdx.x] = first;
      first += c[k];
    }
  }
  __syncthreads();
  // 3. rank and place: the sorted image of the region, in LDS
  const uint32_t shift = (uint32_t) (start & 7ull);
#pragma unroll
  for (uint32_t k = 0; k < kSortPerThread; ++k) {
    const uint32_t key = e[k] >> 16;
    const bool real = key < kDummyKey;
    const uint32_t pos = __hip_atomic_fetch_add(&cnt[real ? key : kDummyKey], 1u, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_rtn_u32
    image[real ? shift + pos : kDummyPlace] = (uint16_t) e[k];
  }
  __syncthreads();
  __builtin_amdgcn_s_setprio(CB_SORT_PRIO_IO);
  // 4. the image leaves as one linear block: image[shift + i] -> sorted[start + i]
  uint16_t *dst = b.sorted + (start - shift);  // 16-byte aligned; index = position in the image
  const uint32_t lo = shift, hi = shift + n;
  const uint32_t body_lo = (lo + 7u) & ~7u, body_hi = hi & ~7u;
  if (body_lo < body_hi) {
    for (uint32_t i = lo + threadIdx.x; i < body_lo; i += kSortThreads) dst[i] = image[i];
    const uint4 *s4 = reinterpret_cast<const uint4 *>(image);
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
    for (uint32_t i = (body_lo >> 3) + threadIdx.x; i < (body_hi >> 3); i += kSortThreads) d4[i] = s4[i];
    for (uint32_t i = body_hi + threadIdx.x; i < hi; i += kSortThreads) dst[i] = image[i];
  } else {
    for (uint32_t i = lo + threadIdx.x; i < hi; i += kSortThreads) dst[i] = image[i];
  }
  if (kChunked && threadIdx.x == 0) b.region_count[r] = n;  // from chunks to entries (every region is sorted once)
  }  // (not the lean instance)
  }  // regions of the batch
  if (kLean && kRunBatch == 8u && threadIdx.x < (b.n_tiles < kGroupTiles ? b.n_tiles : kGroupTiles)) {
    *reinterpret_cast<uint4 *>(b.run_start + (size_t) threadIdx.x * b.max_regions + r_batch) =
        make_uint4(runs[0], runs[1], runs[2], runs[3]);
  }
  }  // batches of this workgroup
}

// ---- gather + accumulate ---------------------------------------------------------------------------

// One workgroup per (tile, slice of regions).  (Measured too: one workgroup of 1024 threads per PAIR of tiles,
// whose runs lie side by side -- a third fewer 128-byte lines fetched, half the workgroups per CU: no faster.)
template <uint32_t kAccThreads, bool kMaskedAdds>
__global__ void __launch_bounds__(kAccThreads) bin_gather_accumulate_kernel(BinLayout b,
                                                                            unsigned long long *hist,
                                                                            int w, int h) {
  __shared__ uint32_t tile[kTilePixels];  // 64 KiB
  __builtin_amdgcn_s_setprio(CB_GATHER_PRIO);  // (see bin_region_sort_kernel)
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
  const uint32_t g = t >> kGroupShift, kl = t & (kGroupTiles - 1u);
  const uint32_t k0 = g << kGroupShift;
  const uint32_t nk = (b.n_tiles - k0) < kGroupTiles ? (b.n_tiles - k0) : kGroupTiles;
  const uint32_t first = b.group_first[g], n_reg = b.group_regions[g];
  const uint32_t S = b.n_regions[1];  // regions per slice of this launch (bin_slice_table_kernel)
  const uint32_t r0 = (s - b.slice_base[t]) * S;
  const uint32_t r1 = (r0 + S < n_reg) ? r0 + S : n_reg;

  for (uint32_t p = threadIdx.x; p < (uint32_t) kTilePixels; p += kAccThreads) tile[p] = 0u;
  __syncthreads();
  // Each lane walks one region's run of this tile: [beg, end) inside the region's sorted image, read with
  // 16-byte loads from the 16-byte boundary below the run (entries outside the run are masked).
  const uint16_t *row0 = b.run_start + (size_t) kl * b.max_regions + first;
  const bool last_key = (kl + 1u == nk);  // its runs end where the region ends
  const uint16_t *row1 = row0 + b.max_regions;
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  // a run's description: where it starts (16-byte boundary below it), the entries to skip there, its end
  struct Run {
    const uint4 *p4;
    uint32_t lead, len;
  };
  const auto describe = [&](uint32_t base) -> Run {
    Run run = {reinterpret_cast<const uint4 *>(b.sorted), 0u, 0u};  // lanes without a run load (and ignore) the first bytes
    const uint32_t rr = base + lane;
    if (rr < r1) {
      // (chunked: region r's image is sorted[r * 32768 ...]; region_start there names its chunks)
      const unsigned long long rs = b.chunked ? (unsigned long long) (first + rr) * kRegionEntries : b.region_start[first + rr];
      const unsigned long long beg = rs + row0[rr];
      const unsigned long long end = rs + (last_key ? b.region_count[first + rr] : (uint32_t) row1[rr]);
      if (end != beg) {  // an empty run reads nothing
        run.lead = (uint32_t) (beg & 7ull);
        run.len = run.lead + (uint32_t) (end - beg);
        run.p4 = reinterpret_cast<const uint4 *>(b.sorted + (beg - run.lead));
      }
    }
    return run;
  };
  constexpr uint32_t kStep = (kAccThreads / 64u) * 64u;
  Run next = describe(r0 + wv * 64u);
  for (uint32_t base = r0 + wv * 64u; base < r1; base += kStep) {
    const Run run = next;
    next = describe(base + kStep);  // the next batch's descriptors are on their way while this batch is read
    const uint32_t lead = run.lead, len = run.len;
    const uint4 *p4 = run.p4;
    const uint32_t last8 = len ? (len - 1u) >> 3 : 0u;  // loads are unconditional, the index clamped to the run
    for (uint32_t at = 0; __ballot(at < len) != 0ull; at += 32u) {
      if (at < len) {
        // up to 32 entries = four 16-byte loads in flight per lane
        uint4 v[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
          const uint32_t i8 = (at >> 3) + j;
          v[j] = p4[i8 < last8 ? i8 : last8];
        }
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
          // (a block of eight that no lane of the wave has is skipped: runs average 32 entries behind their lead-in,
          // so most rounds end after one or two of their four blocks -- beside the draw kernel the gather's vector
          // instructions are what it costs)
          const uint32_t i0 = at + 8u * j;
          if (CB_GATHER_SKIP_BLOCKS && __ballot(i0 < len) == 0ull) break;
          const uint32_t words[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
          // a block that lies inside the run in every lane (the runs of a hot tile are long, and alike): no tests
          if (CB_GATHER_SKIP_BLOCKS && __ballot(i0 < lead || i0 + 8u > len) == 0ull) {
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q) lds_inc(&tile[(words[q >> 1] >> ((q & 1u) * 16u)) & 0xffffu]);
            continue;
          }
          if constexpr (kMaskedAdds) {
            // a block that the ends of some lane's run cut: which of its eight entries are the run's, as bits, and
            // every entry adds its bit -- an entry outside the run (a neighbouring tile's, or a clamped load's: any
            // offset lies inside the tile) adds 0.  Two vector instructions per entry and no scalar one, where a test
            // per entry is three and six (compare, compare, and, save the mask, branch, restore).  The scatter is 5 %
            // faster beside the draw launch for it, and the draw launch 1 % slower: kept where the scatter's chain is
            // the longer of the two (canvases of more than 1024 tiles, C4: -1 % per step; C3: +1 %).
            const int lo = (int) lead - (int) i0, hi = (int) len - (int) i0;  // the run's entries here: lo <= q < hi
            const uint32_t lo_c = (uint32_t) (lo < 0 ? 0 : (lo > 8 ? 8 : lo)), hi_c = (uint32_t) (hi < 0 ? 0 : (hi > 8 ? 8 : hi));
            const uint32_t valid = hi_c > lo_c ? ((1u << (hi_c - lo_c)) - 1u) << lo_c : 0u;
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q) {
              __hip_atomic_fetch_add(&tile[(words[q >> 1] >> ((q & 1u) * 16u)) & 0xffffu], (valid >> q) & 1u,
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          } else {
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q) {
              const uint32_t i = i0 + q;
              if (i >= lead && i < len) lds_inc(&tile[(words[q >> 1] >> ((q & 1u) * 16u)) & 0xffffu]);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  const uint32_t plane_tiles = b.tiles_x * b.tiles_y;
  const uint32_t plane = t / plane_tiles, tt = t - plane * plane_tiles;
  const uint32_t row_0 = (tt / b.tiles_x) << kTileShift;
  const uint32_t col_0 = (tt % b.tiles_x) << kTileShift;
  hist += (unsigned long long) plane * b.plane_pixels;
  for (uint32_t p = threadIdx.x; p < (uint32_t) kTilePixels; p += kAccThreads) {
    const uint32_t v = tile[p];
    if (v != 0u) {
      const uint32_t row = row_0 + (p >> kTileShift);
      const uint32_t col = col_0 + (p & (kTileSize - 1u));
      if (row < (uint32_t) h && col < (uint32_t) w) {  // always true for a recorded pixel
        __hip_atomic_fetch_add(hist + ((unsigned long long) row * (unsigned long long) w + col),
                               (unsigned long long) v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// ---- level A (two levels only): the stream, partitioned into groups of 1024 tiles -------------------
//
// A counting sort over the wave segments: count[key][wave] (LDS histogram per wave segment) -> exclusive
// prefix over waves per key, then over keys -> scatter of whole words into `grouped`, chunk by chunk
// through LDS so that consecutive lanes write consecutive places of a run.  key = group * kReplicas + a
// replica that is a function of the entry's index alone (the count and the scatter kernel evaluate it with
// different thread mappings).
__device__ __forceinline__ uint32_t group_key(const BinLayout &b, uint32_t e, uint32_t j) {
  return ((tile_of(e, b) >> kGroupShift) * kReplicas) | (j & (kReplicas - 1u));
}

__global__ void __launch_bounds__(kScatterThreads) group_count_kernel(BinLayout b) {
  extern __shared__ uint32_t lds[];  // [n_groups * kReplicas]
  const uint32_t r = blockIdx.x;
  const uint32_t n = wave_count_of(b, r);
  const uint32_t *src = b.stream + (size_t) r * b.cap;  // 16-byte aligned: cap is a multiple of 8
  const uint32_t nk = b.n_groups * kReplicas;
  for (uint32_t t = threadIdx.x; t < nk; t += blockDim.x) lds[t] = 0u;
  __syncthreads();
  const uint32_t n4 = n >> 2;
  const uint4 *src4 = reinterpret_cast<const uint4 *>(src);
  for (uint32_t i = threadIdx.x; i < n4; i += blockDim.x) {
    const uint4 v = src4[i];
    const uint32_t j = i << 2;
    lds_inc(&lds[group_key(b, v.x, j)]);
    lds_inc(&lds[group_key(b, v.y, j + 1u)]);
    lds_inc(&lds[group_key(b, v.z, j + 2u)]);
    lds_inc(&lds[group_key(b, v.w, j + 3u)]);
  }
  for (uint32_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) lds_inc(&lds[group_key(b, src[i], i)]);
  __syncthreads();
  // key-major: the scan over waves reads each key's row contiguously
  for (uint32_t t = threadIdx.x; t < nk; t += blockDim.x) b.a_count[(size_t) t * b.n_waves + r] = lds[t];
}

// ---- chunked level A: the chunks of the stream, listed by group ---------------------------------------
//
// chunk_desc[w][j] = group << 16 | words for the chunks j < wave_count[w] of wave w (written by the draw kernel).
// chunk_count: a_count[g][w] = chunks of group g in wave w; group_scan_rows / group_scan_keys make the exclusive
// prefixes (over waves per group, over groups); chunk_list: chunk_list[a_base[g] + a_count[g][w] + rank] =
// {first word of the chunk in the stream, words} -- the order of a (group, wave)'s chunks among themselves is
// whatever the atomics give, which nothing depends on.
__global__ void __launch_bounds__(64) chunk_count_kernel(BinLayout b) {
  __shared__ uint32_t cnt[kChunkedGroupsMax];
  const uint32_t w = blockIdx.x;
  cnt[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t n = b.wave_count[w] < b.chunks_per_wave ? b.wave_count[w] : b.chunks_per_wave;
  const uint32_t *desc = b.chunk_desc + (size_t) w * b.chunks_per_wave;
  for (uint32_t j = threadIdx.x; j < n; j += 64u) {
    const uint32_t g = desc[j] >> 16;
    if (g < b.n_groups) lds_inc(&cnt[g]);
  }
  __syncthreads();
  if (threadIdx.x < b.n_groups) b.a_count[(size_t) threadIdx.x * b.n_waves + w] = cnt[threadIdx.x];
}

__global__ void __launch_bounds__(64) chunk_list_kernel(BinLayout b) {
  __shared__ uint32_t cnt[kChunkedGroupsMax];
  const uint32_t w = blockIdx.x;
  cnt[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t n = b.wave_count[w] < b.chunks_per_wave ? b.wave_count[w] : b.chunks_per_wave;
  const uint32_t *desc = b.chunk_desc + (size_t) w * b.chunks_per_wave;
  for (uint32_t j = threadIdx.x; j < n; j += 64u) {
    const uint32_t d = desc[j], g = d >> 16;
    if (g >= b.n_groups) continue;  // (never written by the draw kernel)
    const uint32_t words = (d & 0xffffu) < kChunkWords ? (d & 0xffffu) : kChunkWords;
    const uint32_t rank = __hip_atomic_fetch_add(&cnt[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned long long at = b.a_base[g] + b.a_count[(size_t) g * b.n_waves + w] + rank;
    b.chunk_list[at] = make_uint2(w * b.cap + j * kChunkWords, words);  // all entries together are < 2^32
  }
}

// a_count[key][w] <- sum of a_count[key][w'] for w' < w; a_base[key] <- total of the key (scanned next).
__global__ void __launch_bounds__(256) group_scan_rows_kernel(BinLayout b) {
  __shared__ uint32_t wave_totals[4];
  const uint32_t len = b.n_waves;
  uint32_t *row = b.a_count + (size_t) blockIdx.x * b.n_waves;
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
  if (threadIdx.x == 0) b.a_base[blockIdx.x] = total;
}

// a_base[key] <- sum of totals of keys < key (a_base[n] <- grand total).  One workgroup.
__global__ void __launch_bounds__(1024) group_scan_keys_kernel(BinLayout b) {
  __shared__ unsigned long long part[1024];
  const uint32_t nk = b.chunked ? b.n_groups : b.n_groups * kReplicas;  // <= 1024
  const unsigned long long v = threadIdx.x < nk ? b.a_base[threadIdx.x] : 0ull;
  part[threadIdx.x] = v;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan (64-bit)
    const unsigned long long add = threadIdx.x >= d ? part[threadIdx.x - d] : 0ull;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  if (threadIdx.x < nk) b.a_base[threadIdx.x] = part[threadIdx.x] - v;
  if (threadIdx.x == 1023) b.a_base[nk] = part[1023];
}

__global__ void __launch_bounds__(kScatterThreads) group_scatter_kernel(BinLayout b) {
  extern __shared__ uint32_t lds[];
  // dynamic LDS: bounds[nk] {start of the key's run in the sorted chunk, its next place in the output} |
  // cnt[nk] | wave_totals[8] | sorted[chunk] {place in the output, word}
  const uint32_t nk = b.n_groups * kReplicas;
  uint2 *bounds = reinterpret_cast<uint2 *>(lds);
  uint32_t *cnt = lds + 2u * nk;
  uint32_t *wave_totals = cnt + nk;
  uint2 *sorted = reinterpret_cast<uint2 *>(wave_totals + 8);

  const uint32_t r = blockIdx.x;
  const uint32_t n = wave_count_of(b, r);
  if (n == 0) return;
  const uint32_t *src = b.stream + (size_t) r * b.cap;
  for (uint32_t t = threadIdx.x; t < nk; t += kScatterThreads) {
    // all entries together are < 2^32, so 32-bit places suffice
    bounds[t].y = (uint32_t) b.a_base[t] + b.a_count[(size_t) t * b.n_waves + r];
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
        key[k] = group_key(b, e[k], base + i);
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
    // 3. sort the chunk in LDS: destination place and word, key by key
#pragma unroll
    for (uint32_t k = 0; k < kPerThread; ++k) {
      if (key[k] != ~0u) {
        const uint2 bd = bounds[key[k]];
        sorted[bd.x + rank[k]] = make_uint2(bd.y + rank[k], e[k]);
      }
    }
    __syncthreads();
    // 4. write the runs out (consecutive lanes -> consecutive places of a run) and advance the places
    for (uint32_t i = threadIdx.x; i < taken_in_chunk; i += kScatterThreads) {
      const uint2 v = sorted[i];
      b.grouped[v.x] = v.y;
    }
    for (uint32_t t = threadIdx.x; t < nk; t += kScatterThreads) bounds[t].y += cnt[t];
    __syncthreads();
  }
}

// ---- workspace carving -----------------------------------------------------------------------------

struct Shape {
  bool ok;
  uint32_t tiles_x, n_tiles, two_level, n_groups, tiles_y;  // n_tiles: of all planes together
  uint32_t chunked;  // two levels, level A by the draw kernel (kernels.h, kChunkWords)
};

Shape shape_of(int w, int h, uint32_t planes) {
  Shape s = {false, 0, 0, 0, 0, 0, 0};
  if (w <= 0 || h <= 0 || w > 65536 || h > 65536 || planes < 1u) return s;
  const uint32_t tiles_x = ((uint32_t) w + kTileSize - 1u) >> kTileShift;
  const uint32_t tiles_y = ((uint32_t) h + kTileSize - 1u) >> kTileShift;
  const unsigned long long n = (unsigned long long) tiles_x * tiles_y * planes;
  if (n > (unsigned long long) kMaxKeys) return s;
  s.ok = true;
  s.tiles_x = tiles_x;
  s.tiles_y = tiles_y;
  s.n_tiles = (uint32_t) n;
  s.two_level = n > kGroupTiles ? 1u : 0u;
  if (const char *e = cb_debug_knob("CUDABROT_AMD_TWO_LEVEL")) {  // test knob: two levels on a small canvas
    if (atoi(e) != 0) s.two_level = 1u;
  }
  s.n_groups = (s.n_tiles + kGroupTiles - 1u) / kGroupTiles;
  s.chunked = (s.two_level && s.n_groups <= kChunkedGroupsMax) ? 1u : 0u;
  if (const char *e = cb_debug_knob("CUDABROT_AMD_CHUNKED")) {  // test knob: 0 = level A as a counting sort over the stream
    if (atoi(e) == 0) s.chunked = 0u;
  }
  return s;
}

// Upper bound on the number of regions for `entries` stream entries: every wave segment (one level) or
// group (two levels) ends with one partial region.
// (A multiple of kRunBatch: a row of run_start is written kRunBatch regions at a time, 16 bytes on a 16-byte boundary.)
uint32_t max_regions_for(const Shape &s, uint32_t n_waves, unsigned long long entries) {
  uint32_t n;
  if (s.chunked) {
    n = (uint32_t) (entries / kRegionEntries) + s.n_groups + 1u;  // entries = chunks * kChunkWords
  } else if (s.two_level) {
    n = (uint32_t) (entries / kGroupRegionEntries) + s.n_groups + 1u;
  } else {
    n = (uint32_t) (entries / kRegionEntries) + n_waves + 1u;
  }
  return (n + kRunBatch - 1u) / kRunBatch * kRunBatch;
}

// Bytes of everything but the per-entry buffers, for a region table of max_regions.
size_t fixed_bytes(const Shape &s, uint32_t n_waves, uint32_t max_regions) {
  const size_t rows = s.n_tiles < kGroupTiles ? s.n_tiles : kGroupTiles;
  size_t b = 0;
  b += round_up((size_t) n_waves * sizeof(uint32_t), 256);                          // wave_count
  b += round_up((size_t) max_regions * sizeof(unsigned long long), 256);            // region_start
  b += 2 * round_up((size_t) max_regions * sizeof(uint32_t), 256);                  // region_count/group
  b += 2 * round_up((size_t) kMaxGroups * sizeof(uint32_t), 256);                  // group_first/regions
  b += round_up(((size_t) (n_waves > kMaxGroups ? n_waves : kMaxGroups) + 1) * sizeof(uint32_t), 256);  // owner_first
  b += 256;                                                                         // n_regions
  b += round_up(rows * max_regions * sizeof(uint16_t), 256);                        // run_start
  b += round_up(((size_t) s.n_tiles + 1) * sizeof(uint32_t), 256);                  // slice_base
  if (s.two_level) {
    const size_t keys = (size_t) s.n_groups * kReplicas;  // (chunked: n_groups rows are used)
    b += round_up(keys * n_waves * sizeof(uint32_t), 256);                          // a_count
    b += round_up((keys + 1) * sizeof(unsigned long long), 256);                    // a_base
  }
  if (s.chunked) b += round_up((size_t) max_regions * kRegionEntries * sizeof(uint16_t), 256);  // sorted: an image per region
  return b + 1024;
}

// per entry of the stream (chunked: the stream, and per chunk of 1024 a descriptor and a list entry, counted as
// 1 byte per entry; `sorted` is sized by the regions, fixed_bytes)
size_t bytes_per_entry(const Shape &s) {
  if (s.chunked) return sizeof(uint32_t) + 1;
  return sizeof(uint32_t) + sizeof(uint16_t) + (s.two_level ? sizeof(uint32_t) : 0);
}

template <typename T>
T *carve(uintptr_t &p, size_t bytes) {
  T *r = reinterpret_cast<T *>(p);
  p += round_up(bytes, 256);
  return r;
}

uint32_t slice_regions_setting() {
  if (const char *e = cb_debug_knob("CUDABROT_AMD_SLICE")) {  // tuning knob: regions per accumulate workgroup
    const long v = atol(e);
    if (v >= 1 && v <= (1l << 24)) return (uint32_t) v;
  }
  return kSliceRegionsDefault;
}

}  // namespace

size_t bin_workspace_bytes(int w, int h, uint32_t n_waves, double entries_per_wave, int n_planes) {
  const Shape s = shape_of(w, h, (uint32_t) (n_planes > 0 ? n_planes : 1));
  if (!s.ok || n_waves == 0) return 0;
  if (entries_per_wave < 2.0 * kMinRegionEntries) entries_per_wave = 2.0 * kMinRegionEntries;
  if (s.chunked) entries_per_wave += (double) (s.n_groups + 3u) * kChunkWords;  // a partly filled chunk per group, and the room a burst asks for
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
  b.slice_regions = slice_regions_setting();
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
  // The region table (and with it run_start) is sized by the entries, which are sized by what is left:
  // shrink cap until everything fits.
  const size_t per_entry = bytes_per_entry(s);
  unsigned long long cap = (p_end - p) / (per_entry * (size_t) n_waves);
  const unsigned long long cap_limit = 0xffffffffull / n_waves;  // all entries together < 2^32
  if (cap > cap_limit) cap = cap_limit;
  for (;;) {
    cap &= ~7ull;  // segments start on 16-byte boundaries of the 16-bit sorted image too
    if (s.chunked) cap &= ~(unsigned long long) (kChunkWords - 1u);  // a segment is a whole number of chunks
    if (cap < kMinRegionEntries) return b;
    const uint32_t mr = max_regions_for(s, n_waves, cap * n_waves);
    const size_t need = fixed_bytes(s, n_waves, mr) + (size_t) cap * n_waves * per_entry + 1024;
    if (p + need <= p_end) break;
    const size_t over = p + need - p_end;
    const unsigned long long cut = over / (per_entry * (size_t) n_waves) + (s.chunked ? kChunkWords : 8u);
    if (cut >= cap) return b;
    cap -= cut;
  }
  b.cap = (uint32_t) cap;
  b.max_regions = max_regions_for(s, n_waves, cap * n_waves);
  const size_t rows = b.n_tiles < kGroupTiles ? b.n_tiles : kGroupTiles;
  b.wave_count = carve<uint32_t>(p, (size_t) n_waves * sizeof(uint32_t));
  b.region_start = carve<unsigned long long>(p, (size_t) b.max_regions * sizeof(unsigned long long));
  b.region_count = carve<uint32_t>(p, (size_t) b.max_regions * sizeof(uint32_t));
  b.region_group = carve<uint32_t>(p, (size_t) b.max_regions * sizeof(uint32_t));
  b.group_first = carve<uint32_t>(p, (size_t) kMaxGroups * sizeof(uint32_t));
  b.group_regions = carve<uint32_t>(p, (size_t) kMaxGroups * sizeof(uint32_t));
  b.owner_first = carve<uint32_t>(p, ((size_t) (n_waves > kMaxGroups ? n_waves : kMaxGroups) + 1) * sizeof(uint32_t));
  b.n_regions = carve<uint32_t>(p, 256);
  b.chunked = s.chunked;
  b.chunks_per_wave = s.chunked ? b.cap / kChunkWords : 0u;
  b.run_start = carve<uint16_t>(p, rows * b.max_regions * sizeof(uint16_t));
  b.slice_base = carve<uint32_t>(p, ((size_t) b.n_tiles + 1) * sizeof(uint32_t));
  if (b.two_level) {
    const size_t keys = (size_t) b.n_groups * kReplicas;
    b.a_count = carve<uint32_t>(p, keys * n_waves * sizeof(uint32_t));
    b.a_base = carve<unsigned long long>(p, (keys + 1) * sizeof(unsigned long long));
  }
  b.stream = carve<uint32_t>(p, (size_t) n_waves * b.cap * sizeof(uint32_t));
  if (b.chunked) {
    b.chunk_desc = carve<uint32_t>(p, (size_t) n_waves * b.chunks_per_wave * sizeof(uint32_t));
    b.chunk_list = carve<uint2>(p, (size_t) n_waves * b.chunks_per_wave * sizeof(uint2));
    b.sorted = carve<uint16_t>(p, (size_t) b.max_regions * kRegionEntries * sizeof(uint16_t));
  } else {
    if (b.two_level) b.grouped = carve<uint32_t>(p, (size_t) n_waves * b.cap * sizeof(uint32_t));
    b.sorted = carve<uint16_t>(p, (size_t) n_waves * b.cap * sizeof(uint16_t));
  }
  b.enabled = 1;
  return b;
}

// One wave that sleeps: the head of the scatter's chain.  The chain runs beside the NEXT draw launch, on the other stream,
// and both are released by the same event; whichever reaches the dispatcher first gets the empty CUs.  If that is this
// chain's first kernel (2048 one-wave workgroups, a few registers each, all over the GPU), the draw's waves -- 128
// registers each, two per SIMD -- land behind them, and what they leave free on a SIMD is no longer one block of 256
// registers: the region sort's workgroup (four waves of 64 on every SIMD of a CU) then fits nowhere until draw waves
// exit, i.e. the sort runs AFTER the draw instead of beside it (C4: a step of 12.1 ms instead of 9.4, every other step).
#ifndef CB_CHAIN_DELAY_US
#define CB_CHAIN_DELAY_US 40
#endif
__global__ void __launch_bounds__(64) chain_delay_kernel(uint32_t ticks) {
  const unsigned long long t0 = wall_clock64();  // 100 MHz
  for (uint32_t turn = 0; turn < 4096u && wall_clock64() - t0 < ticks; ++turn) __builtin_amdgcn_s_sleep(32);
}

hipError_t launch_binned_scatter(const BinLayout &b, unsigned long long *hist, int w, int h,
                                 hipStream_t stream) {
  if (!b.enabled) return hipSuccess;
  if (CB_CHAIN_DELAY_US) hipLaunchKernelGGL(chain_delay_kernel, dim3(1), dim3(64), 0, stream, CB_CHAIN_DELAY_US * 100u);
  if (b.chunked) {
    hipLaunchKernelGGL(chunk_count_kernel, dim3(b.n_waves), dim3(64), 0, stream, b);
    hipLaunchKernelGGL(group_scan_rows_kernel, dim3(b.n_groups), dim3(256), 0, stream, b);
    hipLaunchKernelGGL(group_scan_keys_kernel, dim3(1), dim3(1024), 0, stream, b);
    hipLaunchKernelGGL(chunk_list_kernel, dim3(b.n_waves), dim3(64), 0, stream, b);
  } else if (b.two_level) {
    const uint32_t nk = b.n_groups * kReplicas;
    hipLaunchKernelGGL(group_count_kernel, dim3(b.n_waves), dim3(kScatterThreads), nk * sizeof(uint32_t), stream, b);
    hipLaunchKernelGGL(group_scan_rows_kernel, dim3(nk), dim3(256), 0, stream, b);
    hipLaunchKernelGGL(group_scan_keys_kernel, dim3(1), dim3(1024), 0, stream, b);
    const size_t scatter_lds = ((size_t) 3 * nk + 8 + 2 * kChunkEntries) * sizeof(uint32_t);
    if (scatter_lds > 64 * 1024) {  // 76 KiB at 1024 keys; gfx950 has 160 KiB per workgroup
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(group_scatter_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int) scatter_lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(group_scatter_kernel, dim3(b.n_waves), dim3(kScatterThreads), scatter_lds, stream, b);
  }
  hipLaunchKernelGGL(bin_region_heads_kernel, dim3(1), dim3(1024), 0, stream, b);
  hipLaunchKernelGGL(bin_fill_regions_kernel, dim3((b.max_regions + 255u) / 256u), dim3(256), 0, stream, b);
  hipLaunchKernelGGL(bin_slice_table_kernel, dim3(1), dim3(1024), 0, stream, b);
  const bool plain = b.n_planes == 1u && b.e_row_shift == 16u && b.e_col_mask == 0xffffu && b.e_row_mask == 0xffffu &&
                     b.e_chan_mask == 0u;
  const bool few = b.n_tiles <= kFewTilesMax;
  uint32_t skip_lean = 0u;
#ifndef CB_SORT_LDS_PAD  // (a measuring switch: more LDS per sort workgroup than it needs -- 16384 leaves room for ONE per CU)
#define CB_SORT_LDS_PAD 0
#endif
#ifndef CB_GATHER_LDS_PAD  // (likewise for the gather, whose 64 KiB tile is static)
#define CB_GATHER_LDS_PAD 0
#endif
  const auto launch_sort = [&](auto kernel, size_t lds_bytes, uint32_t batch = 1u) -> hipError_t {
    lds_bytes += CB_SORT_LDS_PAD;
    // per call: the attribute belongs to the current device
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes);  // ~74 KiB of the 160 per CU
    if (e != hipSuccess) return e;
    const uint32_t batches = (b.max_regions + batch - 1u) / batch;  // (the lean instance: kRunBatch regions at a time)
    const uint32_t grid = batches < kSortGrid ? batches : kSortGrid;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kSortThreads), lds_bytes, stream, b, skip_lean);
    return hipSuccess;
  };
  hipError_t se;
  if (b.chunked) {
#ifndef CB_NO_LEAN_SORT
    se = plain ? launch_sort(bin_region_sort_kernel<true, false, true, true>, SortLds<false>::kBytes, kRunBatch)
#else
    se = plain ? launch_sort(bin_region_sort_kernel<true, false, true>, SortLds<false>::kBytes)
#endif
               : launch_sort(bin_region_sort_kernel<false, false, true>, SortLds<false>::kBytes);
#ifndef CB_NO_LEAN_SORT
  } else if (plain && !b.two_level && b.tiles_x <= 128u) {  // (columns below 16384: the lean instance's word)
    // the full regions first (lean instance: 10 vector instructions per entry), then the waves' last, partial ones
    se = few ? launch_sort(bin_region_sort_kernel<true, true, false, true>, SortLds<true>::kBytes, kRunBatch)
             : launch_sort(bin_region_sort_kernel<true, false, false, true>, SortLds<false>::kBytes, kRunBatch);
    // (a one-level plain stream: every region starts on a 16-byte boundary of its wave's segment -- cap is a multiple
    // of 8 entries -- so the general instance has nothing left)
    if (se == hipSuccess && (b.cap & 7u) != 0u) {
      skip_lean = 1u;
      se = few ? launch_sort(bin_region_sort_kernel<true, true>, SortLds<true>::kBytes)
               : launch_sort(bin_region_sort_kernel<true, false>, SortLds<false>::kBytes);
    }
#endif
  } else if (few) {
    se = plain ? launch_sort(bin_region_sort_kernel<true, true>, SortLds<true>::kBytes)
               : launch_sort(bin_region_sort_kernel<false, true>, SortLds<true>::kBytes);
  } else {
    se = plain ? launch_sort(bin_region_sort_kernel<true, false>, SortLds<false>::kBytes)
               : launch_sort(bin_region_sort_kernel<false, false>, SortLds<false>::kBytes);
  }
  if (se != hipSuccess) return se;
  // upper bound on the accumulate workgroups: every tile's last, partial slice + the full ones
  const unsigned long long rows = b.n_tiles < kGroupTiles ? b.n_tiles : kGroupTiles;
  const unsigned long long by_cap = rows * b.max_regions / b.slice_regions;  // slices of the largest size
  const unsigned long long slices = b.n_tiles + (by_cap > kSliceTargetGroups ? by_cap : kSliceTargetGroups) + 1ull;
  // 1024 threads per workgroup on every canvas.  (Canvases of more than 1024 tiles had 512, for two workgroups per CU
  // with the GPU to themselves; beside the draw launch there is LDS for ONE 64 KiB tile per CU either way, and a
  // workgroup of 1024 threads has twice the runs in flight: C4 +3 %, tools/gpu_define_sweep.sh.)
#ifndef CB_GATHER_NARROW_TWO_LEVEL
#define CB_GATHER_NARROW_TWO_LEVEL 0
#endif
  const bool masked = CB_GATHER_MASKED_ADDS == 2 || (CB_GATHER_MASKED_ADDS == 1 && b.two_level);
  const auto launch_gather = [&](auto kernel, uint32_t threads) {
    hipLaunchKernelGGL(kernel, dim3((uint32_t) slices), dim3(threads), CB_GATHER_LDS_PAD, stream, b, hist, w, h);
  };
  if (b.two_level && CB_GATHER_NARROW_TWO_LEVEL) {
    if (masked) launch_gather(bin_gather_accumulate_kernel<kAccThreadsNarrow, true>, kAccThreadsNarrow);
    else launch_gather(bin_gather_accumulate_kernel<kAccThreadsNarrow, false>, kAccThreadsNarrow);
  } else {
    if (masked) launch_gather(bin_gather_accumulate_kernel<kAccThreadsWide, true>, kAccThreadsWide);
    else launch_gather(bin_gather_accumulate_kernel<kAccThreadsWide, false>, kAccThreadsWide);
  }
  return hipGetLastError();
}

}  // namespace cb
