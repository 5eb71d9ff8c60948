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
// partitioned by 128x128-pixel tile with a counting sort (count per (wave, tile) -> exclusive scan
// -> scatter of 14-bit in-tile offsets, no atomics, deterministic), and one workgroup per tile
// accumulates its bucket in an LDS histogram and adds the tile to the u64 histogram with coalesced
// atomics: one 64-byte request per 8 pixels per launch instead of one per increment.
constexpr int kTileShift = 7;
constexpr int kTileSize = 1 << kTileShift;          // 128 x 128 pixels
constexpr int kTilePixels = kTileSize * kTileSize;  // 16384 u32 counters = 64 KiB of LDS
constexpr uint32_t kMaxTiles = 4096;                // canvases up to 8192 x 8192
constexpr uint32_t kMinRegionEntries = 4096;        // below this per wave the workspace is not used

struct BinLayout {
  uint32_t enabled;     // 0: REPLAY adds to the histogram directly
  uint32_t n_waves;     // regions in the stream (= waves of the draw kernel)
  uint32_t cap;         // entries per region (multiple of 4)
  uint32_t n_tiles;     // tiles_x * tiles_y
  uint32_t tiles_x;
  uint32_t slice_entries;  // entries one accumulate workgroup takes
  uint32_t *wave_count;           // [n_waves]            entries written by each wave
  uint32_t *stream;               // [n_waves][cap]       packed row << 16 | col
  uint32_t *count;                // [n_tiles][n_waves]   counts, then exclusive prefix over waves
  unsigned long long *tile_base;  // [n_tiles + 1]        exclusive prefix over tiles
  uint32_t *slice_base;           // [n_tiles + 1]        exclusive prefix of accumulate slices
  uint16_t *sorted;               // [n_waves * cap]      in-tile offsets grouped by tile
};

// Fixed (entry-independent) bytes of a workspace and bytes per stream entry.
size_t bin_fixed_bytes(uint32_t n_waves, uint32_t n_tiles);
constexpr size_t kBinBytesPerEntry = sizeof(uint32_t) + sizeof(uint16_t);
// Carves `bytes` at `workspace` into a BinLayout (enabled = 0 if it is too small or the canvas does
// not qualify: more than kMaxTiles tiles or a side above 65536).
BinLayout make_bin_layout(void *workspace, size_t bytes, int w, int h, uint32_t n_waves);
// count -> scan -> scatter -> accumulate on `stream`, after the draw kernel that filled the stream.
hipError_t launch_binned_scatter(const BinLayout &b, unsigned long long *hist, int w, int h,
                                 hipStream_t stream);

struct DrawArgs {
  // canvas (cudabrot.cu:46-58) + exact-reciprocal fast path
  double min_real, min_imag, delta_real, delta_imag, inv_delta_real, inv_delta_imag;
  int w, h, pow2_real, pow2_imag;
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
};

constexpr uint32_t kCarryHeaderWords = 8;
constexpr uint32_t kCarryQueueWords = (2 * 128 + 4 * 128 + 2 * 192);  // = sizeof(WaveQueues) / 8
constexpr uint32_t kCarryLanePlanes = 18;
constexpr uint32_t kCarryWordsPerWave = kCarryHeaderWords + kCarryQueueWords + kCarryLanePlanes * 64;

constexpr uint32_t kDrawBlockThreads = 256;  // 4 waves per workgroup
inline uint32_t draw_wave_count(uint32_t n_threads) {
  return ((n_threads + kDrawBlockThreads - 1u) / kDrawBlockThreads) * (kDrawBlockThreads / 64u);
}

hipError_t launch_rng_init(uint64_t seed, uint64_t first_subsequence, uint32_t n_threads,
                           uint32_t *d_states, const uint32_t *d_matrices, hipStream_t stream);
hipError_t launch_draw_simple(const DrawArgs &a, hipStream_t stream);
hipError_t launch_draw_wave(const DrawArgs &a, bool timed, hipStream_t stream);
hipError_t launch_draw_wave_ship(const DrawArgs &a, bool timed, hipStream_t stream);

// Steps per chunk of the LONG stage; the stage split is chosen so that no chunk straddles min_iter.
constexpr int kChunk = 32;
void plan_stages(int max_iter, int min_iter, int *head_steps, int *mid_steps);

}  // namespace cb
