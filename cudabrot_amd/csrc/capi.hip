// capi.hip -- implementation of include/cudabrot_amd.h (the C ABI).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "kernels.h"

namespace {

#define CB_TRY(expr)                       \
  do {                                     \
    hipError_t cb_e_ = (expr);             \
    if (cb_e_ != hipSuccess) return (int) cb_e_; \
  } while (0)

// One copy of the jump matrices per device, created on first use and kept for the process lifetime.
std::mutex g_matrix_mutex;
const uint32_t *g_matrices[64] = {nullptr};

int device_matrices(const uint32_t **out) {
  int dev = 0;
  CB_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return (int) hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(g_matrix_mutex);
  if (!g_matrices[dev]) {
    std::vector<uint32_t> host((size_t) cb::kSeqJumpMatrices * cb::kMatrixWords);
    cb::build_sequence_jump_matrices(host.data());
    uint32_t *d = nullptr;
    CB_TRY(hipMalloc(&d, host.size() * sizeof(uint32_t)));
    hipError_t e = hipMemcpy(d, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void) hipFree(d);
      return (int) e;
    }
    g_matrices[dev] = d;
  }
  *out = g_matrices[dev];
  return 0;
}

// ---- the interior map: data that decides what the kernels compute --------------------------------------------------
//
// One bit per cell of side 2^-level of the c-plane: every sample of a marked cell provably never escapes (made and PROVEN
// on the CPU by tools/interior_map.c; DrawArgs::interior_map).  It is EMBEDDED in this library and in the binary (maps.S:
// .incbin of the file `make` unpacks and whose sha256 it checks against the digest kept in the tree) -- there is no file
// to find, lose or swap at run time.  One copy per device, made on first use.  CUDABROT_AMD_INTERIOR_MAP=<file> (a test
// knob, behind CUDABROT_AMD_DEBUG=1) takes another map of the same format -- header, level and EXACT length are checked;
// anything else is an error, never a fallback.
extern "C" {
extern const unsigned char cb_embedded_interior_map[], cb_embedded_interior_map_end[];
}

enum MapKind { kMapInterior = 0, kMapKinds };
struct DeviceMap {
  const unsigned char *d_bytes;
  uint32_t level, cols, rows;
};
std::mutex g_map_mutex;
DeviceMap g_maps[64][kMapKinds];
int g_map_state[64][kMapKinds] = {{0}};  // 0: not tried, 1: there, -1: failed (reported once)
std::atomic<int> g_interior_level{0};    // cb_debug_interior_map_level: the level of the map the last launch used (0: none)

// header: magic, level, columns, rows (u32 each); then the payload, whose length the header implies
bool parse_map(MapKind, const unsigned char *bytes, size_t n, DeviceMap *out, size_t *payload) {
  if (n < 16) return false;
  uint32_t h[4];
  memcpy(h, bytes, 16);
  const uint32_t level = h[1];
  if (h[0] != 0x4d494243u /* "CBIM" */ || level < 8u || level > 15u || h[2] != (5u << level) / 2u ||
      h[3] != (5u << level) / 4u) {  // 2.5 and 1.25 * 2^level
    return false;
  }
  *payload = ((size_t) h[2] * h[3] + 7) / 8;
  if (n != 16 + *payload) return false;  // not a byte more or less
  *out = DeviceMap{nullptr, level, h[2], h[3]};
  return true;
}

const DeviceMap *device_map(MapKind kind) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(g_map_mutex);
  if (g_map_state[dev][kind] == 0) {
    g_map_state[dev][kind] = -1;
    const unsigned char *bytes = cb_embedded_interior_map;
    size_t n = (size_t) (cb_embedded_interior_map_end - cb_embedded_interior_map);
    std::vector<unsigned char> file;
    const char *knob = cb_debug_knob("CUDABROT_AMD_INTERIOR_MAP");
    if (knob) {  // (a test knob: another map)
      FILE *f = fopen(knob, "rb");
      if (f) {
        unsigned char buf[1 << 16];
        size_t got;
        while ((got = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + got);
        fclose(f);
      }
      bytes = file.data();
      n = file.size();
    }
    DeviceMap m;
    size_t payload = 0;
    if (!parse_map(kind, bytes, n, &m, &payload)) {
      fprintf(stderr, "cudabrot_amd: %s is not a map of the kind it stands for (header, level or length): refused\n",
              knob ? knob : "the embedded map");
      return nullptr;
    }
    const unsigned char *src = bytes + 16;
    unsigned char *d = nullptr;
    if (hipMalloc(&d, payload) != hipSuccess) return nullptr;
    if (hipMemcpy(d, src, payload, hipMemcpyHostToDevice) != hipSuccess) {
      (void) hipFree(d);
      return nullptr;
    }
    m.d_bytes = d;
    g_maps[dev][kind] = m;
    g_map_state[dev][kind] = 1;
  }
  return g_map_state[dev][kind] == 1 ? &g_maps[dev][kind] : nullptr;
}

// The interior map for a launch of the wave-scheduled kernels (their MID stage in its one-piece form looks samples up;
// the lock-step kernel and the other forms ignore it): where orbits may be retired early at all -- not the Burning
// Ship, not the full-iterate variants (CUDABROT_AMD_NO_INTERIOR_MAP=1, a test knob: never).  0, or an error: a map that
// should be there and is not is never passed over in silence.
bool wants_interior_map(bool ship, int base_variant) {
  return !(ship || base_variant == CB_KERNEL_FULL_ITERATE || base_variant == CB_KERNEL_SIMPLE ||
           cb_debug_knob("CUDABROT_AMD_TIMED_FULL") != nullptr || cb_debug_knob("CUDABROT_AMD_NO_INTERIOR_MAP") != nullptr);
}
int attach_interior_map(cb::DrawArgs &a, bool ship, int base_variant) {
  g_interior_level.store(0, std::memory_order_relaxed);
  if (!wants_interior_map(ship, base_variant)) return 0;
  const DeviceMap *m = device_map(kMapInterior);
  if (!m) return (int) hipErrorInvalidValue;
  a.interior_map = m->d_bytes;
  a.interior_shift = m->level - 1u;
  a.interior_cols = m->cols;
  a.interior_rows = m->rows;
  g_interior_level.store((int) m->level, std::memory_order_relaxed);
  return 0;
}

// x / delta == x * (1 / delta) bit for bit iff delta is a (normal) power of two.
bool exact_reciprocal(double delta, double *inv) {
  int e = 0;
  if (!(delta > 0.0) || !isfinite(delta)) return false;
  if (frexp(delta, &e) != 0.5) return false;
  const double r = 1.0 / delta;
  if (!isfinite(r) || r == 0.0 || fpclassify(delta) != FP_NORMAL || fpclassify(r) != FP_NORMAL) {
    return false;
  }
  *inv = r;
  return true;
}

// Diagnostic only: which draw kernel the last cb_draw_buddhabrot* call of this process launched (cb_debug_last_draw_kernel).
std::atomic<int> g_last_draw_kernel{0};

// Diagnostic only (CUDABROT_AMD_WAVE_DUMP=<file> with the timed kernel variant): per-wave records of
// the last launch, see DrawArgs::wave_dump.
unsigned long long *g_wave_dump = nullptr;

cb::DrawArgs make_args(const cb_fractal_dimensions *dims, const cb_iteration_control *it,
                       cb_pixel *d_hist, void *d_states, uint32_t n_threads,
                       uint32_t samples_per_thread, cb_counters *d_counters, void *d_workspace,
                       size_t workspace_bytes, void *d_carry, int n_channels = 0) {
  cb::DrawArgs a;
  memset(&a, 0, sizeof(a));
  a.min_real = dims->min_real;
  a.min_imag = dims->min_imag;
  a.delta_real = dims->delta_real;
  a.delta_imag = dims->delta_imag;
  a.inv_delta_real = 0.0;
  a.inv_delta_imag = 0.0;
  a.pow2_real = exact_reciprocal(dims->delta_real, &a.inv_delta_real) ? 1 : 0;
  a.pow2_imag = exact_reciprocal(dims->delta_imag, &a.inv_delta_imag) ? 1 : 0;
  a.rcp_delta_real = 1.0 / dims->delta_real;  // only ever an estimate: any value (inf, nan) is safe
  a.rcp_delta_imag = 1.0 / dims->delta_imag;
  a.replay_min2_real = dims->min_real + dims->min_real;
  a.replay_min2_imag = dims->min_imag + dims->min_imag;
  if (a.pow2_real && a.pow2_imag) {
    a.replay_scale_real = 0.5 * a.inv_delta_real;
    a.replay_scale_imag = 0.5 * a.inv_delta_imag;
    a.replay_offset_real = -(dims->min_real * a.inv_delta_real);
    a.replay_offset_imag = -(dims->min_imag * a.inv_delta_imag);
  } else {
    a.replay_scale_real = dims->delta_real;
    a.replay_scale_imag = dims->delta_imag;
    a.replay_offset_real = dims->min_real;
    a.replay_offset_imag = dims->min_imag;
  }
  a.w = dims->w;
  a.h = dims->h;
  a.replay_bound_w = (double) dims->w;
  a.replay_bound_h = (double) dims->h;
  a.max_iter = it->max_escape_iterations;
  a.min_iter = it->min_escape_iterations;
  cb::plan_stages(a.max_iter, a.min_iter, &a.head_steps, &a.mid_steps);
  {
    const int long_steps = a.max_iter - (a.head_steps + a.mid_steps);
    a.long_steps = long_steps > 0 ? (uint32_t) long_steps : 0u;
    a.tail_steps = a.long_steps % (uint32_t) cb::kChunk;
    a.tail_value = a.tail_steps ? a.tail_steps : ~0u;
    a.accept_rem = a.max_iter - a.min_iter;
    a.fast_mid = (a.min_iter >= a.head_steps + a.mid_steps && long_steps > 0) ? 1 : 0;
    a.long_start = a.head_steps + a.mid_steps;
    a.tail_start = a.long_start + (int) (a.long_steps - a.tail_steps);
    a.sparse_long = (long_steps > 0 && a.min_iter <= a.long_start && cb_debug_knob("CUDABROT_AMD_DENSE_TESTS") == nullptr) ? 1 : 0;
    a.sparse_threshold = 16.0 - 0x1p-10;
    if (const char *e = cb_debug_knob("CUDABROT_AMD_SPARSE_THRESHOLD")) {  // test knob: only ever lower
      const double v = atof(e);
      if (v > 0.0 && v < a.sparse_threshold) a.sparse_threshold = v;
    }
  }
  a.n_threads = n_threads;
  a.samples_per_thread = samples_per_thread;
  a.hist = reinterpret_cast<unsigned long long *>(d_hist);
  a.states = reinterpret_cast<uint32_t *>(d_states);
  a.counters = d_counters;
  a.bin = cb::make_bin_layout(d_workspace, workspace_bytes, dims->w, dims->h,
                              cb::draw_wave_count(n_threads), n_channels);
  a.wave_dump = g_wave_dump;
  a.check_periodic = 1;
  a.carry = reinterpret_cast<unsigned long long *>(d_carry);
  a.drain = (d_carry != nullptr && samples_per_thread == 0) ? 1 : 0;
  return a;
}

// A draw call that launches nothing (no samples and nothing to drain) must still leave the workspace in the
// state cb_flush_scatter expects -- an empty stream -- or the flush would re-add the previous launch's stream
// (or read the counts of a workspace no kernel has written yet).
int empty_stream_if_nothing_launches(const cb::DrawArgs &a, hipStream_t stream) {
  const bool launches = a.n_threads != 0 && (a.samples_per_thread != 0 || (a.carry != nullptr && a.drain != 0));
  if (launches || !a.bin.enabled) return 0;
  return (int) hipMemsetAsync(a.bin.wave_count, 0, (size_t) a.bin.n_waves * sizeof(uint32_t), stream);
}

// Stream entries a launch is expected to produce: visited in-canvas points per sample are 0.4 (max_iter
// 100) to 1.7 (max_iter 20000) on the full canvas; anything beyond the estimate falls back to atomics.
constexpr double kEntriesPerSample = 2.5;
// Passes fused into one launch by cb_renderer.  A launch has fixed costs (the scatter's table kernels, the seams
// of the pipeline, the spread of the waves' end times): 128 instead of 64 is + 6 % on the reference's default
// canvas, + 7 % at max_iter 2000, + 2.5 % at 20000 (4096^2), + 0.5 % at 20000^2; 256 would not fit the stream's
// 32-bit entry indices.  Costs workspace: 2 x 23.7 GiB at 4096^2.
constexpr uint32_t kRendererPassesPerLaunch = 128;

}  // namespace

struct cb_renderer {
  int device;
  cb_fractal_dimensions dims;
  cb_iteration_control iterations;
  uint32_t n_threads;
  cb_pixel *d_hist;
  uint32_t *d_states;
  cb_counters *d_counters;
  // Scatter workspaces, allocated on first use.  Two of them, so that the flush of launch n (on
  // flush_stream) overlaps the draw kernel of launch n+1 (on stream): the flush is HBM/LDS work, the
  // draw kernel is fp64 work.
  void *d_workspace[2];
  size_t workspace_bytes;
  int workspace_tried;
  int next_workspace;
  bool flush_pending[2];
  hipEvent_t draw_done[2], flush_done[2];
  hipStream_t stream, flush_stream;
  // In-flight work carried from launch to launch (cb_draw_buddhabrot's d_carry); drained lazily by
  // finish() before anything reads the histogram or the counters.
  void *d_carry;
  bool carry_pending;
  int carry_variant;
  // fused multi-channel render: n_channels > 0 windows, d_hist holds that many planes
  int n_channels;
  cb_iteration_control windows[CB_MAX_CHANNELS];
  // the level of the interior map this renderer's last launch used (0: none) -- its own record: the process-wide
  // cb_debug_interior_map_level is whichever rank's launch came last
  int interior_level;
};

namespace {

// Adds one launch (or, with passes == 0, the drain of the carried work) and its flush to the
// renderer's streams.
int enqueue_launch(cb_renderer *r, uint32_t passes, int kernel_variant) {
  const bool wave = (kernel_variant & ~(CB_KERNEL_FLAG_BURNING_SHIP | CB_KERNEL_FLAG_DRAIN)) != CB_KERNEL_SIMPLE;
  const bool deferred = r->d_workspace[0] && wave;
  const int k = r->next_workspace;
  if (deferred && r->flush_pending[k]) {
    // workspace k is free again once the flush that read it has finished
    CB_TRY(hipStreamWaitEvent(r->stream, r->flush_done[k], 0));
    r->flush_pending[k] = false;
  }
  int rc;
  if (r->n_channels > 0) {
    rc = cb_draw_buddhabrot_channels(&r->dims, r->d_hist, r->windows, r->n_channels, r->d_states,
                                     r->n_threads, passes * CB_SAMPLES_PER_THREAD, r->d_counters,
                                     kernel_variant, deferred ? r->d_workspace[k] : nullptr,
                                     r->workspace_bytes, r->d_carry, r->stream);
  } else {
    rc = cb_draw_buddhabrot(&r->dims, r->d_hist, &r->iterations, r->d_states, r->n_threads,
                            passes * CB_SAMPLES_PER_THREAD, r->d_counters, kernel_variant,
                            deferred ? r->d_workspace[k] : nullptr, r->workspace_bytes,
                            wave ? r->d_carry : nullptr, r->stream);
  }
  if (rc) return rc;
  {
    const bool ship = (kernel_variant & CB_KERNEL_FLAG_BURNING_SHIP) != 0;
    const int base = kernel_variant & ~(CB_KERNEL_FLAG_BURNING_SHIP | CB_KERNEL_FLAG_DRAIN);
    const DeviceMap *m = wants_interior_map(ship, base) ? device_map(kMapInterior) : nullptr;
    r->interior_level = m ? (int) m->level : 0;
  }
  if (wave && r->d_carry) {
    r->carry_pending = passes != 0;
    r->carry_variant = kernel_variant;
  }
  if (deferred) {
    CB_TRY(hipEventRecord(r->draw_done[k], r->stream));
    CB_TRY(hipStreamWaitEvent(r->flush_stream, r->draw_done[k], 0));
    if (r->n_channels > 0) {
      rc = cb_flush_scatter_channels(&r->dims, r->d_hist, r->n_channels, r->n_threads, r->d_workspace[k],
                                     r->workspace_bytes, r->flush_stream);
    } else {
      rc = cb_flush_scatter(&r->dims, r->d_hist, r->n_threads, r->d_workspace[k], r->workspace_bytes,
                            r->flush_stream);
    }
    if (rc) return rc;
    CB_TRY(hipEventRecord(r->flush_done[k], r->flush_stream));
    r->flush_pending[k] = true;
    r->next_workspace = k ^ 1;
  }
  return 0;
}

int sync_streams(cb_renderer *r) {
  CB_TRY(hipStreamSynchronize(r->stream));
  CB_TRY(hipStreamSynchronize(r->flush_stream));
  r->flush_pending[0] = r->flush_pending[1] = false;
  return 0;
}

// Completes the work earlier launches left in the carry buffer: after this the histogram and the
// counters account for every sample drawn so far.
int finish(cb_renderer *r) {
  if (r->carry_pending) {
    int rc = enqueue_launch(r, 0, r->carry_variant);
    if (rc) return rc;
  }
  int rc = sync_streams(r);
  if (rc) return rc;
  // A kernel that saw one of its invariants broken (a queue ring overrun, a replay that does not end) has
  // lost or duplicated samples: the histogram must not be taken for a result.
  unsigned long long status = 0;
  CB_TRY(hipMemcpyAsync(&status, &r->d_counters->status, sizeof(status), hipMemcpyDeviceToHost, r->stream));
  CB_TRY(hipStreamSynchronize(r->stream));
  return status ? CB_ERROR_KERNEL_INVARIANT : 0;
}

__global__ void __launch_bounds__(256) add_histogram_kernel(unsigned long long *dst,
                                                            const unsigned long long *src, size_t n) {
  const size_t stride = (size_t) gridDim.x * blockDim.x;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] += src[i];
}

// ncclReduce of the renderers' histograms onto renderers[0] (one device each).  librccl is opened on
// first use; the symbols are the ones rccl.h declares.
std::mutex g_rccl_mutex;
decltype(&ncclCommDestroy) g_comm_destroy = nullptr;
std::vector<int> g_comm_devices;   // the devices of the cached communicators ...
std::vector<ncclComm_t> g_comms;   // ... one per device, kept until one of their renderers goes (below)

// The cached communicators end with the first renderer of their set that is destroyed (cb_renderer_destroy): nothing of
// RCCL's is left to the teardown of the process.
void release_communicators_with(int device) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  bool mine = false;
  for (int d : g_comm_devices) mine = mine || d == device;
  if (!mine || !g_comm_destroy) return;
  for (ncclComm_t c : g_comms) (void) g_comm_destroy(c);
  g_comms.clear();
  g_comm_devices.clear();
}

int rccl_reduce_to_root(cb_renderer *const *renderers, int n, size_t count) {
  static void *lib = nullptr;
  static decltype(&ncclCommInitAll) comm_init_all = nullptr;
  static decltype(&ncclGroupStart) group_start = nullptr;
  static decltype(&ncclGroupEnd) group_end = nullptr;
  static decltype(&ncclReduce) reduce = nullptr;
  decltype(&ncclCommDestroy) &comm_destroy = g_comm_destroy;
  std::vector<int> &comm_devices = g_comm_devices;
  std::vector<ncclComm_t> &comms = g_comms;
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (!lib) {
    lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return (int) hipErrorSharedObjectInitFailed;
    comm_init_all = reinterpret_cast<decltype(comm_init_all)>(dlsym(lib, "ncclCommInitAll"));
    comm_destroy = reinterpret_cast<decltype(&ncclCommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    group_start = reinterpret_cast<decltype(group_start)>(dlsym(lib, "ncclGroupStart"));
    group_end = reinterpret_cast<decltype(group_end)>(dlsym(lib, "ncclGroupEnd"));
    reduce = reinterpret_cast<decltype(reduce)>(dlsym(lib, "ncclReduce"));
    if (!comm_init_all || !comm_destroy || !group_start || !group_end || !reduce) {
      return (int) hipErrorSharedObjectSymbolNotFound;
    }
  }
  std::vector<int> devices(n);
  for (int k = 0; k < n; ++k) devices[k] = renderers[k]->device;
  // One communicator per set of devices, kept while its renderers live: creating it costs far more than the reduce (a
  // bootstrap over all ranks), and a renderer set is reduced at every checkpoint.
  if (comm_devices != devices) {
    for (ncclComm_t c : comms) (void) comm_destroy(c);
    comms.assign((size_t) n, nullptr);
    comm_devices.clear();
    if (comm_init_all(comms.data(), n, devices.data()) != ncclSuccess) {
      comms.clear();
      return (int) hipErrorUnknown;
    }
    comm_devices = devices;
  }
  ncclResult_t nr = group_start();
  for (int k = 0; k < n && nr == ncclSuccess; ++k) {
    if (hipSetDevice(devices[k]) != hipSuccess) {
      nr = ncclUnhandledCudaError;
      break;
    }
    // in place on the root; u64 counters, integer sum: the result does not depend on the order
    nr = reduce(renderers[k]->d_hist, renderers[k]->d_hist, count, ncclUint64, ncclSum, 0, comms[k],
                renderers[k]->stream);
  }
  const ncclResult_t ne = group_end();
  int rc = (nr == ncclSuccess && ne == ncclSuccess) ? 0 : (int) hipErrorUnknown;
  for (int k = 0; k < n; ++k) {
    (void) hipSetDevice(devices[k]);
    const hipError_t e = hipStreamSynchronize(renderers[k]->stream);
    if (e != hipSuccess && rc == 0) rc = (int) e;
  }
  if (rc != 0) {  // a communicator that has failed is not reused
    for (ncclComm_t c : comms) (void) comm_destroy(c);
    comms.clear();
    comm_devices.clear();
  }
  return rc;
}

}  // namespace

extern "C" {

int cb_abi_version(void) { return CB_ABI_VERSION; }

int cb_debug_last_draw_kernel(void) { return g_last_draw_kernel.load(std::memory_order_relaxed); }
int cb_debug_interior_map_level(void) { return g_interior_level.load(std::memory_order_relaxed); }
int cb_renderer_interior_map_level(const cb_renderer *r) { return r ? r->interior_level : 0; }

const char *cb_error_string(int code) {
  if (code == 0) return "no error";
  if (code == CB_ERROR_KERNEL_INVARIANT) {
    return "the draw kernel reported a broken internal invariant (cb_counters.status): samples were lost";
  }
  return hipGetErrorString((hipError_t) code);
}

int cb_initialize_rng(uint64_t seed, uint64_t first_subsequence, uint32_t n_threads, void *d_states,
                      void *stream) {
  if (n_threads == 0) return 0;
  if (!d_states) return (int) hipErrorInvalidValue;
  const uint32_t *mats = nullptr;
  int rc = device_matrices(&mats);
  if (rc) return rc;
  return (int) cb::launch_rng_init(seed, first_subsequence, n_threads,
                                   reinterpret_cast<uint32_t *>(d_states), mats,
                                   reinterpret_cast<hipStream_t>(stream));
}

size_t cb_carry_bytes(uint32_t n_threads) {
  return (size_t) cb::draw_wave_count(n_threads) * cb::kCarryWordsPerWave * sizeof(unsigned long long) +
         (size_t) cb::kSchedWords * sizeof(uint32_t);
}

size_t cb_scatter_workspace_bytes_channels(const cb_fractal_dimensions *dims, int n_channels, uint32_t n_threads,
                                           uint32_t samples_per_thread) {
  if (!dims || dims->w <= 0 || dims->h <= 0 || n_threads == 0 || samples_per_thread == 0) return 0;
  if (n_channels < 1 || n_channels > CB_MAX_CHANNELS) return 0;
  const uint32_t n_waves = cb::draw_wave_count(n_threads);
  // every plane of a fused launch receives its own share of the visited points
  const double entries = (double) n_threads * (double) samples_per_thread * kEntriesPerSample * (n_channels > 1 ? 1.5 : 1.0);
  return cb::bin_workspace_bytes(dims->w, dims->h, n_waves, entries / n_waves, n_channels);
}

size_t cb_scatter_workspace_bytes(const cb_fractal_dimensions *dims, uint32_t n_threads,
                                  uint32_t samples_per_thread) {
  return cb_scatter_workspace_bytes_channels(dims, 1, n_threads, samples_per_thread);
}

int cb_draw_buddhabrot(const cb_fractal_dimensions *dims, cb_pixel *d_hist,
                       const cb_iteration_control *iterations, void *d_states, uint32_t n_threads,
                       uint32_t samples_per_thread, cb_counters *d_counters, int kernel_variant,
                       void *d_workspace, size_t workspace_bytes, void *d_carry, void *stream) {
  if (!dims || !iterations || !d_hist || !d_states) return (int) hipErrorInvalidValue;
  if (dims->w <= 0 || dims->h <= 0) return (int) hipErrorInvalidValue;
  const bool ship = (kernel_variant & CB_KERNEL_FLAG_BURNING_SHIP) != 0;
  const int base_variant = kernel_variant & ~(CB_KERNEL_FLAG_BURNING_SHIP | CB_KERNEL_FLAG_DRAIN);
  if (base_variant == CB_KERNEL_SIMPLE) {  // the baseline kernel: atomics, every launch complete
    d_workspace = nullptr;
    d_carry = nullptr;
  }
  cb::DrawArgs a = make_args(dims, iterations, d_hist, d_states, n_threads, samples_per_thread,
                             d_counters, d_workspace, workspace_bytes, d_carry);
  a.burning_ship = ship ? 1 : 0;
  if ((kernel_variant & CB_KERNEL_FLAG_DRAIN) && a.carry) a.drain = 1;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  {
    int rc = empty_stream_if_nothing_launches(a, s);
    if (rc) return rc;
  }
  const auto wave = ship ? cb::launch_draw_wave_ship : cb::launch_draw_wave;
  {
    const int rc = attach_interior_map(a, ship, base_variant);
    if (rc) return rc;
  }
  // the two-waves-per-SIMD kernel where it applies (CUDABROT_AMD_NO_WIDE=1, a test knob: never)
  const bool wide = cb::draw_wide_takes(a) && cb_debug_knob("CUDABROT_AMD_NO_WIDE") == nullptr;
  const auto wide_launch = ship ? cb::launch_draw_wide_ship : cb::launch_draw_wide;
  g_last_draw_kernel.store(base_variant == CB_KERNEL_SIMPLE ? 3 : (wide ? 2 : 1), std::memory_order_relaxed);
  switch (base_variant) {
    case CB_KERNEL_DEFAULT:
      return (int) (wide ? wide_launch(a, false, s) : wave(a, false, s));
    case CB_KERNEL_FULL_ITERATE:
      a.check_periodic = 0;
      return (int) (wide ? wide_launch(a, false, s) : wave(a, false, s));
    case CB_KERNEL_TIMED:
      if (cb_debug_knob("CUDABROT_AMD_TIMED_FULL")) a.check_periodic = 0;  // diagnostic: stage clocks of the full-iterate form
      return (int) (wide ? wide_launch(a, true, s) : wave(a, true, s));
    case CB_KERNEL_SIMPLE:
      return (int) cb::launch_draw_simple(a, s);
    default:
      return (int) hipErrorInvalidValue;
  }
}

int cb_flush_scatter(const cb_fractal_dimensions *dims, cb_pixel *d_hist, uint32_t n_threads,
                     void *d_workspace, size_t workspace_bytes, void *stream) {
  if (!dims || !d_hist || dims->w <= 0 || dims->h <= 0) return (int) hipErrorInvalidValue;
  // the carve is a pure function of these arguments, so it is the one the draw call used
  const cb::BinLayout b = cb::make_bin_layout(d_workspace, workspace_bytes, dims->w, dims->h,
                                              cb::draw_wave_count(n_threads));
  return (int) cb::launch_binned_scatter(b, reinterpret_cast<unsigned long long *>(d_hist), dims->w,
                                         dims->h, reinterpret_cast<hipStream_t>(stream));
}

int cb_draw_buddhabrot_channels(const cb_fractal_dimensions *dims, cb_pixel *d_hist,
                                const cb_iteration_control *windows, int n_channels, void *d_states,
                                uint32_t n_threads, uint32_t samples_per_thread, cb_counters *d_counters,
                                int kernel_variant, void *d_workspace, size_t workspace_bytes,
                                void *d_carry, void *stream) {
  if (!dims || !windows || !d_hist || !d_states) return (int) hipErrorInvalidValue;
  if (dims->w <= 0 || dims->h <= 0 || n_channels < 1 || n_channels > CB_MAX_CHANNELS) {
    return (int) hipErrorInvalidValue;
  }
  const bool ship = (kernel_variant & CB_KERNEL_FLAG_BURNING_SHIP) != 0;
  const int base_variant = kernel_variant & ~(CB_KERNEL_FLAG_BURNING_SHIP | CB_KERNEL_FLAG_DRAIN);
  if (base_variant != CB_KERNEL_DEFAULT && base_variant != CB_KERNEL_FULL_ITERATE) {
    return (int) hipErrorInvalidValue;  // the wave-scheduled kernel only
  }
  // the kernel iterates to the largest max; what escapes before the smallest min is in no window
  cb_iteration_control hull = windows[0];
  for (int j = 1; j < n_channels; ++j) {
    if (windows[j].max_escape_iterations > hull.max_escape_iterations) {
      hull.max_escape_iterations = windows[j].max_escape_iterations;
    }
    if (windows[j].min_escape_iterations < hull.min_escape_iterations) {
      hull.min_escape_iterations = windows[j].min_escape_iterations;
    }
  }
  cb::DrawArgs a = make_args(dims, &hull, d_hist, d_states, n_threads, samples_per_thread, d_counters,
                             d_workspace, workspace_bytes, d_carry, n_channels);
  a.burning_ship = ship ? 1 : 0;
  if ((kernel_variant & CB_KERNEL_FLAG_DRAIN) && a.carry) a.drain = 1;
  a.n_channels = n_channels;
  a.plane_pixels = (unsigned long long) dims->w * (unsigned long long) dims->h;
  for (int j = 0; j < n_channels; ++j) {
    a.chan_min[j] = windows[j].min_escape_iterations;
    a.chan_max[j] = windows[j].max_escape_iterations;
  }
  if (base_variant == CB_KERNEL_FULL_ITERATE) a.check_periodic = 0;
  {
    int rc = empty_stream_if_nothing_launches(a, reinterpret_cast<hipStream_t>(stream));
    if (rc) return rc;
  }
  {
    const int rc = attach_interior_map(a, ship, base_variant);
    if (rc) return rc;
  }
  const auto wave = ship ? cb::launch_draw_wave_ship : cb::launch_draw_wave;
  g_last_draw_kernel.store(1, std::memory_order_relaxed);
  return (int) wave(a, false, reinterpret_cast<hipStream_t>(stream));
}

int cb_flush_scatter_channels(const cb_fractal_dimensions *dims, cb_pixel *d_hist, int n_channels,
                              uint32_t n_threads, void *d_workspace, size_t workspace_bytes,
                              void *stream) {
  if (!dims || !d_hist || dims->w <= 0 || dims->h <= 0 || n_channels < 1 || n_channels > CB_MAX_CHANNELS) {
    return (int) hipErrorInvalidValue;
  }
  // every word carries the index of its plane and the planes are sorted as one canvas: one pass
  const cb::BinLayout b = cb::make_bin_layout(d_workspace, workspace_bytes, dims->w, dims->h,
                                              cb::draw_wave_count(n_threads), n_channels);
  return (int) cb::launch_binned_scatter(b, reinterpret_cast<unsigned long long *>(d_hist), dims->w, dims->h,
                                         reinterpret_cast<hipStream_t>(stream));
}

int cb_renderer_create(cb_renderer **out, int device, const cb_fractal_dimensions *dims,
                       const cb_iteration_control *iterations, uint64_t seed,
                       uint64_t first_subsequence, uint32_t n_threads) {
  return cb_renderer_create_channels(out, device, dims, iterations, 0, seed, first_subsequence, n_threads);
}

int cb_renderer_create_channels(cb_renderer **out, int device, const cb_fractal_dimensions *dims,
                                const cb_iteration_control *iterations, int n_channels, uint64_t seed,
                                uint64_t first_subsequence, uint32_t n_threads) {
  if (!out || !dims || !iterations || dims->w <= 0 || dims->h <= 0 || n_threads == 0 || n_channels < 0 ||
      n_channels > CB_MAX_CHANNELS) {
    return (int) hipErrorInvalidValue;
  }
  *out = nullptr;
  CB_TRY(hipSetDevice(device));  // cudabrot.cu:155
  cb_renderer *r = new (std::nothrow) cb_renderer;
  if (!r) return (int) hipErrorOutOfMemory;
  memset(r, 0, sizeof(*r));
  r->device = device;
  r->dims = *dims;
  r->iterations = *iterations;
  r->n_channels = n_channels;
  for (int j = 0; j < n_channels; ++j) r->windows[j] = iterations[j];
  r->n_threads = n_threads;
  const size_t hist_bytes = (size_t) dims->w * (size_t) dims->h * sizeof(cb_pixel) * (size_t) (n_channels ? n_channels : 1);
  hipError_t e = hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->flush_stream, hipStreamNonBlocking);
  for (int k = 0; k < 2 && e == hipSuccess; ++k) {
    e = hipEventCreateWithFlags(&r->draw_done[k], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->flush_done[k], hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipMalloc(&r->d_hist, hist_bytes);                  // cudabrot.cu:168
  if (e == hipSuccess) e = hipMemsetAsync(r->d_hist, 0, hist_bytes, r->stream);  // cudabrot.cu:169
  if (e == hipSuccess) e = hipMalloc(&r->d_states, cb_rng_state_bytes(n_threads));  // :177
  if (e == hipSuccess) e = hipMalloc(&r->d_counters, sizeof(cb_counters));
  if (e == hipSuccess) e = hipMalloc(&r->d_carry, cb_carry_bytes(n_threads));
  if (e == hipSuccess) e = hipMemsetAsync(r->d_carry, 0, cb_carry_bytes(n_threads), r->stream);
  if (e == hipSuccess) e = hipMemsetAsync(r->d_counters, 0, sizeof(cb_counters), r->stream);
  int rc = (int) e;
  if (!rc) rc = cb_initialize_rng(seed, first_subsequence, n_threads, r->d_states, r->stream);
  if (!rc) rc = (int) hipStreamSynchronize(r->stream);  // cudabrot.cu:181
  if (rc) {
    cb_renderer_destroy(r);
    return rc;
  }
  *out = r;
  return 0;
}

namespace {

// 50 samples per thread per reference pass (cudabrot.cu:34,390), at most kRendererPassesPerLaunch passes per launch.
uint32_t max_passes_per_launch() {
  static const uint32_t v = [] {
    const char *e = cb_debug_knob("CUDABROT_AMD_PASSES_PER_LAUNCH");  // experiment knob
    const long n = e ? atol(e) : 0;
    return (n >= 1 && n <= 4096) ? (uint32_t) n : kRendererPassesPerLaunch;
  }();
  return v;
}

// What a renderer allocates on first use, because it depends on the kernel variant: the two scatter
// workspaces (tens of GB on a large canvas -- hipMalloc of that size can take seconds).
void prepare_for_variant(cb_renderer *r, int kernel_variant) {
  if ((kernel_variant & ~(CB_KERNEL_FLAG_BURNING_SHIP | CB_KERNEL_FLAG_DRAIN)) == CB_KERNEL_TIMED && cb_debug_knob("CUDABROT_AMD_WAVE_DUMP") &&
      !g_wave_dump) {
    const size_t bytes = (size_t) cb::draw_wave_count(r->n_threads) * 8 * sizeof(unsigned long long);
    if (hipMalloc(&g_wave_dump, bytes) != hipSuccess || hipMemset(g_wave_dump, 0, bytes) != hipSuccess) {
      g_wave_dump = nullptr;
    }
  }
  if (!r->workspace_tried && (kernel_variant & ~(CB_KERNEL_FLAG_BURNING_SHIP | CB_KERNEL_FLAG_DRAIN)) != CB_KERNEL_SIMPLE &&
      cb_debug_knob("CUDABROT_AMD_NO_WORKSPACE") == nullptr) {
    // scatter workspace for the largest launch render_passes makes; on any failure: direct atomics
    r->workspace_tried = 1;
    size_t want = cb_scatter_workspace_bytes_channels(&r->dims, r->n_channels > 0 ? r->n_channels : 1, r->n_threads,
                                                      max_passes_per_launch() * CB_SAMPLES_PER_THREAD);
    size_t free_b = 0, total_b = 0;
    if (want && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      if (want > free_b / 4) want = free_b / 4;
      if (hipMalloc(&r->d_workspace[0], want) == hipSuccess &&
          hipMalloc(&r->d_workspace[1], want) == hipSuccess) {
        r->workspace_bytes = want;
      } else {
        (void) hipGetLastError();
        (void) hipFree(r->d_workspace[0]);
        (void) hipFree(r->d_workspace[1]);
        r->d_workspace[0] = r->d_workspace[1] = nullptr;
      }
    }
  }
}

}  // namespace

int cb_renderer_prepare(cb_renderer *r, int kernel_variant) {
  if (!r) return (int) hipErrorInvalidValue;
  CB_TRY(hipSetDevice(r->device));
  prepare_for_variant(r, kernel_variant);
  return (int) hipDeviceSynchronize();
}

int cb_renderer_render_passes(cb_renderer *r, uint32_t passes, int kernel_variant) {
  if (!r) return (int) hipErrorInvalidValue;
  CB_TRY(hipSetDevice(r->device));
  const uint32_t max_passes_per_launch = ::max_passes_per_launch();
  prepare_for_variant(r, kernel_variant);
  if (r->carry_pending && r->carry_variant != kernel_variant) {
    int rc = finish(r);  // a different kernel variant cannot take over the carried work
    if (rc) return rc;
  }
  while (passes > 0) {
    const uint32_t now = passes < max_passes_per_launch ? passes : max_passes_per_launch;
    int rc = enqueue_launch(r, now, kernel_variant);
    if (rc) return rc;
    passes -= now;
  }
  // The draw launches are complete when this returns (the fence at the end); the scatter of the last one may
  // still be running on its own stream, so that the first draw launch of the next call starts beside it --
  // a caller that renders in batches (the binary: every ~0.2 s) keeps the pipeline of enqueue_launch going
  // across calls.  Everything that reads the histogram or the counters goes through finish(), which completes
  // the carried orbits and waits for both streams.
  if (g_wave_dump) {  // diagnostic: write the per-wave records of the last launch
    (void) hipStreamSynchronize(r->stream);
    const size_t n = (size_t) cb::draw_wave_count(r->n_threads) * 8;
    std::vector<unsigned long long> host(n);
    if (hipMemcpy(host.data(), g_wave_dump, n * 8, hipMemcpyDeviceToHost) == hipSuccess) {
      if (FILE *f = fopen(cb_debug_knob("CUDABROT_AMD_WAVE_DUMP"), "wb")) {
        fwrite(host.data(), 8, n, f);
        fclose(f);
      }
    }
    (void) hipFree(g_wave_dump);
    g_wave_dump = nullptr;
  }
  return (int) hipStreamSynchronize(r->stream);  // cudabrot.cu:487
}

int cb_renderer_finish(cb_renderer *r) {
  if (!r) return (int) hipErrorInvalidValue;
  CB_TRY(hipSetDevice(r->device));
  return finish(r);
}

int cb_renderer_read_histogram(cb_renderer *r, cb_pixel *host_out) {
  if (!r || !host_out) return (int) hipErrorInvalidValue;
  CB_TRY(hipSetDevice(r->device));
  {
    int rc = finish(r);
    if (rc) return rc;
  }
  const size_t bytes = (size_t) r->dims.w * (size_t) r->dims.h * sizeof(cb_pixel) *
                       (size_t) (r->n_channels ? r->n_channels : 1);
  CB_TRY(hipMemcpyAsync(host_out, r->d_hist, bytes, hipMemcpyDeviceToHost, r->stream));
  return (int) hipStreamSynchronize(r->stream);
}

int cb_renderer_grayscale_image(cb_renderer *r, double gamma, int mode, uint16_t *host_gray_be,
                                uint64_t *max_out, double *scale_out) {
  return cb_renderer_grayscale_plane(r, 0, gamma, mode, host_gray_be, max_out, scale_out);
}

int cb_renderer_grayscale_plane(cb_renderer *r, int plane, double gamma, int mode, uint16_t *host_gray_be,
                                uint64_t *max_out, double *scale_out) {
  if (!r || !host_gray_be || plane < 0 || plane >= (r->n_channels ? r->n_channels : 1)) {
    return (int) hipErrorInvalidValue;
  }
  CB_TRY(hipSetDevice(r->device));
  {
    int rc = finish(r);
    if (rc) return rc;
  }
  const size_t bytes = (size_t) r->dims.w * (size_t) r->dims.h * sizeof(uint16_t);
  uint16_t *d_gray = nullptr;
  CB_TRY(hipMalloc(reinterpret_cast<void **>(&d_gray), bytes));
  int rc = cb_tone_map_device(r->d_hist + (size_t) plane * (size_t) r->dims.w * (size_t) r->dims.h, r->dims.w,
                              r->dims.h, gamma, mode, d_gray, max_out, scale_out, r->stream);
  if (rc == 0) rc = (int) hipMemcpyAsync(host_gray_be, d_gray, bytes, hipMemcpyDeviceToHost, r->stream);
  if (rc == 0) rc = (int) hipStreamSynchronize(r->stream);
  (void) hipFree(d_gray);
  return rc;
}

int cb_renderer_write_histogram(cb_renderer *r, const cb_pixel *host_in) {
  if (!r || !host_in) return (int) hipErrorInvalidValue;
  CB_TRY(hipSetDevice(r->device));
  {
    int rc = finish(r);
    if (rc) return rc;
  }
  const size_t bytes = (size_t) r->dims.w * (size_t) r->dims.h * sizeof(cb_pixel) *
                       (size_t) (r->n_channels ? r->n_channels : 1);
  CB_TRY(hipMemcpyAsync(r->d_hist, host_in, bytes, hipMemcpyHostToDevice, r->stream));
  return (int) hipStreamSynchronize(r->stream);
}

int cb_renderer_read_counters(cb_renderer *r, cb_counters *host_out) {
  if (!r || !host_out) return (int) hipErrorInvalidValue;
  CB_TRY(hipSetDevice(r->device));
  {
    int rc = finish(r);
    if (rc && rc != CB_ERROR_KERNEL_INVARIANT) return rc;  // the counters are how a caller reads the status
  }
  CB_TRY(hipMemcpyAsync(host_out, r->d_counters, sizeof(cb_counters), hipMemcpyDeviceToHost,
                        r->stream));
  return (int) hipStreamSynchronize(r->stream);
}

int cb_renderer_read_rng_states(cb_renderer *r, void *host_out) {
  if (!r || !host_out) return (int) hipErrorInvalidValue;
  CB_TRY(hipSetDevice(r->device));
  {
    int rc = finish(r);  // afterwards no orbit is in flight: states + histogram are a complete checkpoint
    if (rc) return rc;
  }
  CB_TRY(hipMemcpyAsync(host_out, r->d_states, cb_rng_state_bytes(r->n_threads), hipMemcpyDeviceToHost,
                        r->stream));
  return (int) hipStreamSynchronize(r->stream);
}

int cb_renderer_write_rng_states(cb_renderer *r, const void *host_in) {
  if (!r || !host_in) return (int) hipErrorInvalidValue;
  CB_TRY(hipSetDevice(r->device));
  {
    int rc = finish(r);
    if (rc) return rc;
  }
  CB_TRY(hipMemcpyAsync(r->d_states, host_in, cb_rng_state_bytes(r->n_threads), hipMemcpyHostToDevice,
                        r->stream));
  return (int) hipStreamSynchronize(r->stream);
}

cb_pixel *cb_renderer_device_histogram(cb_renderer *r) { return r ? r->d_hist : nullptr; }

void cb_renderer_destroy(cb_renderer *r) {
  if (!r) return;
  (void) hipSetDevice(r->device);
  if (r->stream) (void) hipStreamSynchronize(r->stream);
  if (r->flush_stream) (void) hipStreamSynchronize(r->flush_stream);
  release_communicators_with(r->device);  // (cached by cb_renderers_reduce for a set this renderer may belong to)
  (void) hipFree(r->d_hist);
  (void) hipFree(r->d_states);
  (void) hipFree(r->d_counters);
  (void) hipFree(r->d_carry);
  (void) hipFree(r->d_workspace[0]);
  (void) hipFree(r->d_workspace[1]);
  for (int k = 0; k < 2; ++k) {
    if (r->draw_done[k]) (void) hipEventDestroy(r->draw_done[k]);
    if (r->flush_done[k]) (void) hipEventDestroy(r->flush_done[k]);
  }
  if (r->stream) (void) hipStreamDestroy(r->stream);
  if (r->flush_stream) (void) hipStreamDestroy(r->flush_stream);
  delete r;
}

// The one exchange of the multi-GPU path: renderers[0] += renderers[1..n).  Renderers on
// n distinct devices: one ncclReduce(ncclUint64, ncclSum, root 0) over xGMI.  RCCL is loaded on first
// use (dlopen), so single-GPU users of the library never touch it.  Renderers that all share one device
// (a rehearsal of the sharded path on a one-GPU box): an add kernel.
int cb_renderers_reduce(cb_renderer *const *renderers, int n) {
  if (!renderers || n < 1) return (int) hipErrorInvalidValue;
  for (int k = 0; k < n; ++k) {
    cb_renderer *r = renderers[k];
    if (!r || r->dims.w != renderers[0]->dims.w || r->dims.h != renderers[0]->dims.h ||
        r->n_channels != renderers[0]->n_channels) {
      return (int) hipErrorInvalidValue;
    }
    CB_TRY(hipSetDevice(r->device));
    const int rc = finish(r);
    if (rc) return rc;
  }
  cb_renderer *root = renderers[0];
  const size_t count = (size_t) root->dims.w * (size_t) root->dims.h * (size_t) (root->n_channels ? root->n_channels : 1);
  if (n == 1) {
    // CUDABROT_AMD_FORCE_RCCL=1 (test knob): a reduce over one rank, to exercise the RCCL calls where
    // only one device exists
    return cb_debug_knob("CUDABROT_AMD_FORCE_RCCL") ? rccl_reduce_to_root(renderers, 1, count) : 0;
  }
  bool same = true, distinct = true;
  for (int a = 0; a < n; ++a) {
    if (renderers[a]->device != root->device) same = false;
    for (int b = a + 1; b < n; ++b) {
      if (renderers[a]->device == renderers[b]->device) distinct = false;
    }
  }
  if (same) {
    CB_TRY(hipSetDevice(root->device));
    for (int k = 1; k < n; ++k) {
      hipLaunchKernelGGL(add_histogram_kernel, dim3(256 * 8), dim3(256), 0, root->stream,
                         reinterpret_cast<unsigned long long *>(root->d_hist),
                         reinterpret_cast<const unsigned long long *>(renderers[k]->d_hist), count);
      CB_TRY(hipGetLastError());
    }
    return (int) hipStreamSynchronize(root->stream);
  }
  if (!distinct) return (int) hipErrorInvalidValue;
  return rccl_reduce_to_root(renderers, n, count);
}

}  // extern "C"
