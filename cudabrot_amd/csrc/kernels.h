// kernels.h -- internal launch interface between the C ABI (capi.hip) and the device code.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cudabrot_amd.h"

namespace cb {

// Number of GF(2) jump matrices kept on the device: A^(2^(67+b)), b in [0,64).
constexpr int kSeqJumpMatrices = 64;
constexpr int kMatrixWords = 160 * 5;

// Host: fills out[kSeqJumpMatrices][kMatrixWords] (xorwow_host.cpp).
void build_sequence_jump_matrices(uint32_t *out);
// Host: seeded XORWOW state before any jump (rocrand_xorwow.h:104-123): x[0..4], d.
void seed_state(uint64_t seed, uint32_t x[5], uint32_t *d);

// ---- deferred, tile-binned scatter (scatter.hip) ------------------------------------------------
//
// Random u64 atomics run at ~24 G/s on MI355X whatever the schedule (one 64-byte memory-side request
// each), which caps every configuration of this path.  With a workspace, the REPLAY stage therefore
// does not touch the histogram: each wave appends the visited pixels as packed (row << 16 | col)
// words to its own region of a stream in HBM (coalesced stores).  After the draw kernel the stream is
// cut into REGIONS of 32768 entries; every region is sorted by 128x128-pixel tile on its own, inside one
// workgroup's LDS (no counting pass over the stream, no global prefix sums: the only thing a region
// publishes is where each tile's run starts inside it), and one workgroup per (tile, slice of regions)
// then gathers that tile's runs from the regions, accumulates them in an LDS histogram and adds the tile
// to the u64 histogram with coalesced atomics: one 64-byte request per 8 pixels per slice instead of one
// per increment.
constexpr int kTileShift = 7;
constexpr int kTileSize = 1 << kTileShift;          // 128 x 128 pixels
constexpr int kTilePixels = kTileSize * kTileSize;  // 16384 u32 counters = 64 KiB of LDS
constexpr uint32_t kGroupTiles = 1024;              // keys of one region sort; more tiles: two levels
constexpr uint32_t kMinRegionEntries = 4096;        // below this per wave the workspace is not used
constexpr uint32_t kGroupReplicas = 4;              // level-A keys per group of the two-level sort (scatter.hip)
// Two levels, CHUNKED (up to kChunkedGroupsMax groups of 1024 tiles): level A happens where the words are born.
// A wave's segment of the stream is cut into chunks of kChunkWords words; REPLAY appends a word to the current
// chunk of its GROUP (cursor and limit of every group in the wave's LDS; a full chunk is followed by the next
// free one of the segment) and notes the group of every chunk it opens in chunk_desc.  The scatter then sorts
// REGIONS OF kRegionChunks CHUNKS of one group (lists of chunks instead of stretches of a grouped copy of the
// stream): the stream is written once and read once, where the counting sort of level A read it twice more and
// wrote it once more.
constexpr uint32_t kChunkWords = 1024;
constexpr uint32_t kRegionChunks = 32;              // 32 * 1024 = the 32768 entries of a region sort
constexpr uint32_t kChunkedGroupsMax = 64;          // cursors: 64 x {next word, end of chunk} = 512 B of LDS per wave

struct BinLayout {
  uint32_t enabled;     // 0: REPLAY adds to the histogram directly
  uint32_t n_waves;     // segments of the stream (= waves of the draw kernel)
  uint32_t cap;         // entries per wave segment (multiple of 8)
  uint32_t n_tiles;     // n_planes * tiles_x * tiles_y
  uint32_t tiles_x;
  uint32_t slice_regions;  // regions one accumulate workgroup gathers from
  // more than kGroupTiles tiles: the stream is first partitioned into groups of 1024 consecutive tiles
  // (level A: count -> scan -> scatter of whole words into `grouped`), and the regions are cut from each
  // group's stretch of `grouped`, so that a region holds tiles of one group only; scatter.hip
  uint32_t two_level;
  uint32_t n_groups;    // 1 with one level
  // two levels and n_groups <= kChunkedGroupsMax: the stream is chunked by group as it is written (above);
  // wave_count then holds the chunks a wave has opened, a_count / a_base count and place CHUNKS, and `grouped`
  // does not exist
  uint32_t chunked;
  uint32_t chunks_per_wave;  // cap / kChunkWords
  uint32_t max_regions;   // size of the region table and row length of run_start
  // Layout of a stream word: col in the low bits, row above it (e_row_shift), and -- fused multi-channel
  // renders only -- the index of the channel (plane) the point goes to above both (e_chan_shift,
  // e_chan_mask; 0 / 0 with one plane).  One channel: row << 16 | col.  The planes are sorted as ONE
  // canvas of n_planes * tiles_y tile rows: tile index = (plane * tiles_y + tile row) * tiles_x + tile
  // column, n_tiles = n_planes * tiles_x * tiles_y, so one sort -> accumulate serves all planes.
  uint32_t e_row_shift, e_col_mask, e_row_mask, e_chan_shift, e_chan_mask;
  uint32_t n_planes, tiles_y;
  unsigned long long plane_pixels;  // w * h: distance between the planes of the histogram
  uint32_t *wave_count;           // [n_waves]            entries written by each wave
  uint32_t *stream;               // [n_waves][cap]       packed row << 16 | col
  uint32_t *a_count;              // [n_groups*replicas][n_waves]  level A counts, then prefix over waves
  unsigned long long *a_base;     // [n_groups*replicas + 1]  level A exclusive prefix over keys
  uint32_t *grouped;              // [n_waves * cap]      the stream, grouped (two levels)
  unsigned long long *region_start;  // [max_regions]     regions of the (grouped) stream: first entry ...
  uint32_t *region_count;         // [max_regions]        ... entries (<= 32768) ...
  uint32_t *region_group;         // [max_regions]        ... and group (0 with one level)
  uint32_t *owner_first;          // [max(n_waves, groups) + 1]  first region of every wave (one level) / group
  uint32_t *group_first;          // [n_groups]           a group's regions are consecutive: the first ...
  uint32_t *group_regions;        // [n_groups]           ... and how many
  uint32_t *n_regions;            // [1] (+ the slice size and the entries of the launch behind it)
  uint32_t *chunk_desc;           // [n_waves][chunks_per_wave]  chunked: group << 16 | words in the chunk
  uint2 *chunk_list;              // [n_waves * chunks_per_wave]  chunked: {first word, words} of every chunk, by group
  uint16_t *run_start;            // [min(n_tiles, 1024)][max_regions]  where tile k0 + i's run starts in a region
  uint32_t *slice_base;           // [n_tiles + 1]        exclusive prefix of accumulate workgroups per tile
  uint16_t *sorted;               // [n_waves * cap]      in-tile offsets, every region sorted by tile in place
};

// Bytes of a workspace for launches that write about entries_per_wave stream entries per wave (0 if
// the canvas cannot use one: a side above 65536 or more than 262144 tiles over all planes).
size_t bin_workspace_bytes(int w, int h, uint32_t n_waves, double entries_per_wave, int n_planes = 1);
// Carves `bytes` at `workspace` into a BinLayout (enabled = 0 if it is too small or the canvas does
// not qualify).
BinLayout make_bin_layout(void *workspace, size_t bytes, int w, int h, uint32_t n_waves, int n_channels = 0);
// region sort -> gather + accumulate on `stream`, after the draw kernel that filled the stream.
hipError_t launch_binned_scatter(const BinLayout &b, unsigned long long *hist, int w, int h,
                                 hipStream_t stream);

struct DrawArgs {
  // canvas (cudabrot.cu:46-58) + exact-reciprocal fast path
  double min_real, min_imag, delta_real, delta_imag, inv_delta_real, inv_delta_imag;
  int w, h, pow2_real, pow2_imag;
  // RN(1 / delta) for any delta: the replay burst's quotient estimate (draw_wave.hip, CB_REPLAY_BIN_DIV)
  double rcp_delta_real, rcp_delta_imag;
  // The replay burst's wave-constant operands, made on the host (capi.hip, make_args) so that the kernel reads
  // them from its argument segment with scalar loads where they are used: neither vector instructions to
  // derive them nor scalar registers to hold them across the other stages.
  //   2 min (the tests R >= 2 min_re on doubled coordinates);
  //   dyadic pixels: scale = 0.5 / delta, offset = -(min / delta)  (pixel = fma(R, scale, offset));
  //   else:          scale = delta, offset = min                  (pixel = (R / 2 - offset) / scale)
  double replay_min2_real, replay_min2_imag;
  double replay_scale_real, replay_scale_imag, replay_offset_real, replay_offset_imag;
  double replay_bound_w, replay_bound_h;  // (double) w, (double) h (draw_wide.hip)
  // Likewise the LONG stage's constants (from max_iter, min_iter and the stage split): iterations left to
  // the stage, the length of an orbit's last, shorter chunk and the l_rem that marks it (~0: none), and the
  // largest l_rem at which an escape is accepted (max_iter - min_iter)
  uint32_t long_steps, tail_steps, tail_value;
  int accept_rem;
  int fast_mid;  // the usual split: MID ends at or before min_iter and the LONG stage follows
  // The LONG stage's chunks test for escape on every tenth step only (draw_wave.hip, iterate_chunk2_sparse):
  // allowed when every escape inside a full LONG chunk is accepted, i.e. min_iter <= long_start.  long_start:
  // first iteration of the LONG stage; tail_start: first iteration of an orbit's last, shorter chunk (which
  // tests every step).
  int sparse_long, long_start, tail_start;
  // what |Z|^2 is compared with at a sparse test step: 16 - 2^-10 (draw_wave.hip, kSparseThreshold).  Any value
  // up to that gives the same result -- a lower one only sends more lanes through the exact decision
  // (CUDABROT_AMD_SPARSE_THRESHOLD: how the tests drive that path, which otherwise runs once in 3e8 test steps)
  double sparse_threshold;
  // iteration control (cudabrot.cu:62-67)
  int max_iter, min_iter;
  // stage split of draw_wave_kernel (plan_stages): HEAD runs iterations [0, head_steps), MID the
  // next mid_steps, LONG the rest
  int head_steps, mid_steps;
  uint32_t n_threads;
  uint32_t samples_per_thread;
  unsigned long long *hist;
  uint32_t *states;  // six planes of n_threads
  cb_counters *counters;
  BinLayout bin;
  // timed variant only, may be null: 8 words per wave {HW_ID, XCC_ID, start, end (100 MHz clock),
  // cycles in HEAD, LONG, REPLAY, total}
  unsigned long long *wave_dump;
  // 1: retire orbits found exactly periodic (default); 0: iterate every sample to max_iter like the
  // reference does (CB_KERNEL_FULL_ITERATE, for measuring the iterate loop against its roofline)
  int check_periodic;
  // Carry-over of in-flight work (may be null).  Draining a launch is expensive: the deepest orbits
  // take hundreds of chunks with almost every lane idle.  With a carry buffer a launch stops when
  // its samples are drawn and leaves queues and orbit slots in the buffer for the next launch; a
  // launch with drain != 0 (and normally no samples) finishes everything.  kCarryWordsPerWave u64
  // words per wave, zeroed by the caller before the first launch.
  unsigned long long *carry;
  int drain;
  // the reference's RENDER_BURNING_SHIP variant (cudabrot.cu:15-17); read by draw_simple_kernel, the
  // wave kernel has a build of its own for it (launch_draw_wave_ship)
  int burning_ship;
  // Fused multi-channel render: n_channels > 0 windows [chan_min[j], chan_max[j]) of
  // the escape index; max_iter / min_iter above are then the largest max and the smallest min, and an
  // orbit is replayed once into every channel whose window holds its escape index.  hist is
  // n_channels planes of plane_pixels counters.
  int n_channels;
  int chan_min[CB_MAX_CHANNELS], chan_max[CB_MAX_CHANNELS];
  unsigned long long plane_pixels;
  // The interior map (tools/interior_map.c; null: none): one bit per cell of side 2^-level of the c-plane, re in
  // [-2, 0.5), |im| in [0, 1.25), set where EVERY sample of the cell provably never escapes under the reference's
  // iteration.  draw_wide_kernel's MID stage looks a survivor of HEAD up and retires a marked one as never-escaping
  // instead of iterating it until its orbit repeats.  interior_shift = level - 1 (the kernel's coordinates are doubled).
  const unsigned char *interior_map;
  uint32_t interior_shift, interior_cols, interior_rows;
};

constexpr uint32_t kCarryHeaderWords = 8;
constexpr uint32_t kCarryQueueWords = (2 * 128 + 4 * 96 + 2 * 192 + 192 / 2);  // = sizeof(WaveQueues) / 8
constexpr uint32_t kCarryLanePlanes = 19;
constexpr uint32_t kCarryWordsPerWave = kCarryHeaderWords + kCarryQueueWords + kCarryLanePlanes * 64;
// Behind the per-wave records of a carry buffer: the progress board of the draw kernel's waves, 16 words
// per SIMD (one per wave slot), indexed by (XCC, SE, SH, CU, SIMD).  The waves that share a SIMD post how
// many samples they still have to draw and take their issue priority from their rank (draw_wave.hip).
constexpr uint32_t kSchedKeys = 1u << 16;
constexpr uint32_t kSchedWords = kSchedKeys * 16u;  // u32 words

constexpr uint32_t kDrawBlockThreads = 256;  // 4 waves per workgroup
inline uint32_t draw_wave_count(uint32_t n_threads) {
  return ((n_threads + kDrawBlockThreads - 1u) / kDrawBlockThreads) * (kDrawBlockThreads / 64u);
}

hipError_t launch_rng_init(uint64_t seed, uint64_t first_subsequence, uint32_t n_threads,
                           uint32_t *d_states, const uint32_t *d_matrices, hipStream_t stream);
hipError_t launch_draw_simple(const DrawArgs &a, hipStream_t stream);
hipError_t launch_draw_wave(const DrawArgs &a, bool timed, hipStream_t stream);
hipError_t launch_draw_wave_ship(const DrawArgs &a, bool timed, hipStream_t stream);
// draw_wide.hip: the same path from two waves per SIMD (four orbits per lane in the LONG stage, software-pipelined
// HEAD and REPLAY), so that the scatter of the previous launch fits beside it.  draw_wide_takes: the launches it
// is built for (one-level workspace, one channel, the usual stage split, a carry buffer, whole workgroups of 512
// subsequences); everything else is draw_wave_kernel's.  Same results, own carry format.
bool draw_wide_takes(const DrawArgs &a);
hipError_t launch_draw_wide(const DrawArgs &a, bool timed, hipStream_t stream);
hipError_t launch_draw_wide_ship(const DrawArgs &a, bool timed, hipStream_t stream);

// Steps per chunk of the LONG stage; the stage split is chosen so that no chunk straddles min_iter.
// The exact-periodicity check compares z with a saved point at chunk boundaries only, so a cycle of period
// p is seen p / gcd(p, chunk) chunks after the save.  The periods that matter are mostly multiples of 3
// (by area of the set outside the cardioid and the period-2 disc: 3, 4, 6, 12, 9, 5, 8, 15, 10 ...), so
// 30 = 2 * 3 * 5 finds them sooner than 32: 9 % fewer executed iterations at C3, the draw launch 5 %
// shorter (24 and 36 measured too: tools/gpu_chunk_sweep.sh).  With the sparse escape tests (draw_wave.hip,
// iterate_chunk2_sparse) a chunk of 30 had become so short that the bookkeeping between chunks was a sixth
// of it: 60 = 4 * 3 * 5 executes 5 % more iterations and is 5 % faster (tools/gpu_draw_sweep.sh).
#ifndef CB_CHUNK
#define CB_CHUNK 60
#endif
constexpr int kChunk = CB_CHUNK;
void plan_stages(int max_iter, int min_iter, int *head_steps, int *mid_steps);

}  // namespace cb
