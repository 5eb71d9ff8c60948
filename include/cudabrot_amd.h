/*
 * cudabrot_amd.h -- C ABI of the MI355X-native Buddhabrot hot path (libcudabrot_amd.so).
 *
 * The reference (yalue/cudabrot, one file: cudabrot.cu) has no FFI or plugin interface; its only
 * internal boundary is the pair of kernel launches its host driver makes (last
 * row).  The entry points below are exactly that boundary, as a C ABI: plain pointers and sizes, no
 * C++ or torch types.  Each cites the reference interface it replaces.
 *
 * Conventions
 *  - every function returns 0 on success or a hipError_t value (>0); cb_error_string() names it,
 *    the way the reference prints cudaGetErrorString (cudabrot.cu:134-141);
 *  - pointers named d_* are DEVICE pointers on the current HIP device, `stream` is a hipStream_t
 *    passed as void* (NULL = the default stream); launches are asynchronous on that stream;
 *  - the library never falls back to the CPU: without a usable HIP device every call fails.
 */
#ifndef CUDABROT_AMD_H_
#define CUDABROT_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CB_ABI_VERSION 1

/* Reference constants (cudabrot.cu:20,23,34,37). */
#define CB_DEFAULT_BLOCK_SIZE 512
#define CB_DEFAULT_BLOCK_COUNT 512
#define CB_DEFAULT_THREADS (CB_DEFAULT_BLOCK_SIZE * CB_DEFAULT_BLOCK_COUNT)
#define CB_SAMPLES_PER_THREAD 50
#define CB_DEFAULT_RNG_SEED 1337ull

/* `Pixel` (cudabrot.cu:43), widened to 64 bits as BASELINE.json's north_star asks: identical to the
 * reference's uint32_t counts whenever no count reaches 2^32. */
typedef uint64_t cb_pixel;

/* `FractalDimensions` (cudabrot.cu:46-58): same fields, same order, same layout (56 bytes). */
typedef struct {
  int w;
  int h;
  double min_real;
  double min_imag;
  double max_real;
  double max_imag;
  double delta_real;
  double delta_imag;
} cb_fractal_dimensions;

/* `IterationControl` (cudabrot.cu:62-67). */
typedef struct {
  int max_escape_iterations;
  int min_escape_iterations;
} cb_iteration_control;

/* Exact workload counters, accumulated on the device; the reference has none. */
typedef struct {
  uint64_t samples;        /* starting points drawn (4 XORWOW outputs each)              */
  uint64_t rejected;       /* inside the main cardioid / period-2 bulb (cudabrot.cu:398) */
  uint64_t never_escaped;  /* IterateMandelbrot returned max (cudabrot.cu:407)           */
  uint64_t too_fast;       /* escaped before min_escape_iterations (cudabrot.cu:408)     */
  uint64_t recorded;       /* orbits replayed (cudabrot.cu:412)                          */
  uint64_t iterate_steps;  /* z<-z^2+c iterations of IterateMandelbrot (cudabrot.cu:326),
                              as the reference would execute them                        */
  uint64_t replay_steps;   /* iterations of IterateAndRecord (cudabrot.cu:352)           */
  uint64_t increments;     /* histogram increments (cudabrot.cu:312)                     */
  uint64_t skipped_steps;  /* part of iterate_steps NOT executed: the orbit was found to be
                              exactly periodic, or its cell is proven never-escaping (the
                              interior map)                                               */
  uint64_t status;         /* 0 = ok; nonzero = internal invariant violated (CB_STATUS_*) */
  /* CB_KERNEL_TIMED only (else 0): shader-clock cycles summed over waves, per stage and in total */
  uint64_t cycles_head, cycles_long, cycles_replay, cycles_total;
  /* CB_KERNEL_TIMED only, 100 MHz constant clock, LAST launch only meaningful if counters were zeroed
   * before it: ~(earliest wave start), latest wave end, sum of wave lifetimes */
  uint64_t rt_not_first_start, rt_last_end, rt_wave_life_sum;
} cb_counters;

#define CB_STATUS_QUEUE_OVERFLOW 1u
#define CB_STATUS_REPLAY_RUNAWAY 2u
#define CB_STATUS_INTERIOR_MAP 4u /* a sample of a cell proven never-escaping (interior map) escaped */
#define CB_STATUS_CARRY_FOREIGN 8u /* the carry buffer holds work of the OTHER draw kernel (another workspace size or
                                      none picks another kernel: see cb_draw_buddhabrot): it was not resumed */
/* Returned (instead of a hipError_t) by cb_renderer_finish and by everything that reads a renderer's histogram
 * or image when cb_counters.status is nonzero: a draw kernel saw one of its internal invariants broken and has
 * lost samples, so the histogram is not a result.  cb_renderer_read_counters still succeeds and shows the flags. */
#define CB_ERROR_KERNEL_INVARIANT 100001

/* Kernel variants of cb_draw_buddhabrot. */
#define CB_KERNEL_DEFAULT 0 /* wave-scheduled four-stage kernel (the product path)                */
#define CB_KERNEL_SIMPLE 1  /* one lane = one reference thread, lock-step; a validation baseline  */
#define CB_KERNEL_TIMED 2   /* the default kernel with per-stage s_memtime stamps (diagnostic build) */
#define CB_KERNEL_FULL_ITERATE 3 /* the default kernel without the exact-periodicity early-out: every
                                    sample is iterated to max_iter as the reference does (same result;
                                    for measuring the iterate loop against the fp64 roofline)        */

/* Fused multi-channel renders: at most this many (max, min) windows per launch. */
#define CB_MAX_CHANNELS 4

/* OR-ed into a kernel variant: the reference's RENDER_BURNING_SHIP build (cudabrot.cu:15-17 -- there a
 * compile-time switch): |real|, |imag| before every step and no cardioid / bulb shortcut. */
#define CB_KERNEL_FLAG_BURNING_SHIP 0x100

/* OR-ed into a kernel variant, with a carry buffer: this launch also completes every orbit in flight
 * (its own and the carried ones) instead of leaving them for the next launch -- the last launch of a
 * render.  A launch with samples_per_thread = 0 does only that. */
#define CB_KERNEL_FLAG_DRAIN 0x200

/* RecomputePixelDeltas (cudabrot.cu:505-527).  Returns 1 and fills delta_* if the canvas is valid,
 * else 0 and, if msg is not NULL, *msg points at the reference's message for the failed check. */
int cb_recompute_pixel_deltas(cb_fractal_dimensions *dims, const char **msg);

/* Bytes of device memory cb_initialize_rng / cb_draw_buddhabrot need at d_states for n_threads
 * generator states (replaces `block_size * block_count * sizeof(curandState_t)`, cudabrot.cu:177-178).
 * Layout: six uint32 planes [x0 | x1 | x2 | x3 | x4 | d], each n_threads long. */
size_t cb_rng_state_bytes(uint32_t n_threads);

/* InitializeRNG<<<...>>>(seed, states) (cudabrot.cu:146-149,179): state t becomes the XORWOW
 * generator rocrand_init(seed, first_subsequence + t, 0).  The reference always passes
 * first_subsequence = 0; rank r of a multi-GPU run passes r * n_threads. */
int cb_initialize_rng(uint64_t seed, uint64_t first_subsequence, uint32_t n_threads, void *d_states,
                      void *stream);

/* Suggested size of the scatter workspace of cb_draw_buddhabrot for launches of this shape (0 if the canvas
 * cannot use one: more than 262144 tiles of 128x128 pixels, or a side above 65536).  Any size works:
 * increments that do not fit are added with direct atomics, the result is the same. */
size_t cb_scatter_workspace_bytes(const cb_fractal_dimensions *dims, uint32_t n_threads,
                                  uint32_t samples_per_thread);
/* ... of cb_draw_buddhabrot_channels with n_channels planes (their tiles are sorted as one canvas). */
size_t cb_scatter_workspace_bytes_channels(const cb_fractal_dimensions *dims, int n_channels, uint32_t n_threads,
                                           uint32_t samples_per_thread);

/* DrawBuddhabrot<<<block_count, block_size>>>(dimensions, data, iterations, states)
 * (cudabrot.cu:379-414,485-486) for n_threads threads, samples_per_thread samples each (the
 * reference: 50 per launch; k reference passes in one launch = 50*k).  Adds to d_hist (w*h
 * cb_pixel, row-major, row 0 = min_imag), advances d_states, and adds to d_counters (may be NULL).
 * Without a workspace (d_workspace NULL) every increment is a device-scope atomic on d_hist (the
 * reference's += of cudabrot.cu:312, made atomic) and d_hist is complete when the launch is.
 * With one, the increments that fit are DEFERRED: the kernel streams the visited pixels into the
 * workspace and the caller must then run cb_flush_scatter on the same workspace (stream-ordered
 * after this call) before it reads d_hist or reuses the workspace; what does not fit is added
 * atomically at once, so the sum is the same for any workspace size.
 * d_carry (may be NULL: then the launch completes every sample it draws, like the reference's) is
 * cb_carry_bytes(n_threads) of device memory, zeroed before the first call, that carries orbits
 * still in flight from one launch to the next: finishing the deepest orbits of a launch takes
 * hundreds of iterations with almost every lane idle, so with a carry buffer the launch stops when
 * its samples are drawn.  A final call with samples_per_thread == 0 (followed by its
 * cb_flush_scatter) completes the carried work; only then do d_hist and d_counters account for
 * every sample drawn.  Same geometry, iteration control, kernel variant AND workspace (its size, or none)
 * for all calls sharing a carry buffer: the library has two draw kernels with carry records of their own and
 * picks one by the shape of the launch, the workspace included.  A call that finds the other kernel's work
 * in the buffer does not resume it and says so: cb_counters.status gets CB_STATUS_CARRY_FOREIGN (cb_renderer
 * keeps all of this consistent by itself). */
int cb_draw_buddhabrot(const cb_fractal_dimensions *dims, cb_pixel *d_hist,
                       const cb_iteration_control *iterations, void *d_states, uint32_t n_threads,
                       uint32_t samples_per_thread, cb_counters *d_counters, int kernel_variant,
                       void *d_workspace, size_t workspace_bytes, void *d_carry, void *stream);

/* Bytes of the carry buffer of cb_draw_buddhabrot for n_threads threads. */
size_t cb_carry_bytes(uint32_t n_threads);

/* Second half of the scatter: partitions the pixel stream a cb_draw_buddhabrot call left in
 * d_workspace by 128x128-pixel tile (sorts of regions of 32768 entries) and adds every tile to d_hist from an LDS
 * histogram with coalesced atomics.  Same dims, n_threads, d_workspace and workspace_bytes as that
 * call.  A no-op for a workspace the draw call could not use.  Precondition: the workspace was last written by
 * a cb_draw_buddhabrot call with these arguments (a call that launches nothing -- no samples, nothing to
 * drain -- leaves an empty stream); flushing a workspace no draw call has touched is undefined.
 * Meant to run on ANOTHER stream than the next draw call, beside it (two workspaces, as cb_renderer does): its kernels
 * fit the CUs beside the two-waves-per-SIMD draw kernel.  The first of them is one wave that sleeps 40 us, so that a
 * draw launch released by the same event reaches the empty CUs first (otherwise the region
 * sort cannot be resident beside the draw's waves and runs after them). */
int cb_flush_scatter(const cb_fractal_dimensions *dims, cb_pixel *d_hist, uint32_t n_threads,
                     void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- Fused multi-channel render ---------------------------------------------- *
 *
 * The reference's colour recipe (generate_hires_color_image.sh:27-59) runs the program once per
 * channel with different -m / -c.  All runs draw the same sample stream, so one pass can serve them:
 * every sample is iterated once up to the largest max, and its orbit is replayed once into every
 * channel j whose window windows[j].min <= k < windows[j].max holds the escape index k; the planes are
 * then sorted and accumulated together, as one taller canvas.  d_hist is
 * n_channels planes of w*h counters, plane j = what cb_draw_buddhabrot would add with windows[j].
 * The wave-scheduled kernel only (variant flags as above); cb_flush_scatter_channels after each
 * launch that was given a workspace (sized by cb_scatter_workspace_bytes_channels).
 * Counters of a fused launch: samples, rejected, never_escaped (against the largest max) and
 * iterate_steps as for one run with the largest max; too_fast = orbits that escaped but whose index
 * lies in no window; recorded = orbits in at least one window; replay_steps counts every replay (a first,
 * unrecorded one finds the escape index, then one per channel the orbit belongs to); increments = the
 * histogram increments of all planes together. */
int cb_draw_buddhabrot_channels(const cb_fractal_dimensions *dims, cb_pixel *d_hist,
                                const cb_iteration_control *windows, int n_channels, void *d_states,
                                uint32_t n_threads, uint32_t samples_per_thread, cb_counters *d_counters,
                                int kernel_variant, void *d_workspace, size_t workspace_bytes,
                                void *d_carry, void *stream);
int cb_flush_scatter_channels(const cb_fractal_dimensions *dims, cb_pixel *d_hist, int n_channels,
                              uint32_t n_threads, void *d_workspace, size_t workspace_bytes,
                              void *stream);

/* ---- Renderer: SetupCUDA + RenderImage + the -s buffer, as an owned object ---------------------- */

typedef struct cb_renderer cb_renderer;

/* SetupCUDA (cudabrot.cu:153-189): selects `device`, allocates and zeroes the histogram, allocates
 * and initialises n_threads generator states for subsequences [first_subsequence, +n_threads). */
int cb_renderer_create(cb_renderer **out, int device, const cb_fractal_dimensions *dims,
                       const cb_iteration_control *iterations, uint64_t seed,
                       uint64_t first_subsequence, uint32_t n_threads);
/* The same for a fused multi-channel render: n_channels (1..CB_MAX_CHANNELS) windows; the histogram is
 * n_channels planes of w*h counters, in read/write_histogram too (n_channels = 0: cb_renderer_create). */
int cb_renderer_create_channels(cb_renderer **out, int device, const cb_fractal_dimensions *dims,
                                const cb_iteration_control *windows, int n_channels, uint64_t seed,
                                uint64_t first_subsequence, uint32_t n_threads);
/* `passes` iterations of the loop body of RenderImage (cudabrot.cu:483-487), fused into as few
 * launches as possible; returns after the device has finished the draw launches (the scatter of the last one
 * may still be running, beside which the next call's first launch starts).  Orbits still in flight are
 * carried to the next call; cb_renderer_finish (called by the read/write functions below) completes
 * them and waits for everything. */
int cb_renderer_render_passes(cb_renderer *r, uint32_t passes, int kernel_variant);
/* Optional, before the first cb_renderer_render_passes: allocates now what that call would allocate for
 * this kernel variant (the scatter workspaces: tens of GB on a large canvas), so that a caller who times
 * the pass loop -- like the reference's "passes took" line, cudabrot.cu:499-500 -- does not time hipMalloc. */
int cb_renderer_prepare(cb_renderer *r, int kernel_variant);
/* Completes all carried work: afterwards the device histogram and counters account for every sample
 * of every pass rendered so far (needed before using cb_renderer_device_histogram directly). */
int cb_renderer_finish(cb_renderer *r);
/* The cudaMemcpy of cudabrot.cu:496-497: host_out receives w*h cb_pixel. */
int cb_renderer_read_histogram(cb_renderer *r, cb_pixel *host_out);
/* The H2D copy of LoadInProgressBuffer (cudabrot.cu:256-257): REPLACES the device histogram. */
int cb_renderer_write_histogram(cb_renderer *r, const cb_pixel *host_in);
int cb_renderer_read_counters(cb_renderer *r, cb_counters *host_out);
/* True-resume checkpoint: the generator states as an opaque blob of
 * cb_rng_state_bytes(n_threads) bytes.  The reference's -s buffer holds the histogram only, so a
 * resumed run replays seed 1337 from the start (cudabrot.cu:215-258,179); histogram + these states
 * continue the sample stream instead.  Both finish carried work first. */
int cb_renderer_read_rng_states(cb_renderer *r, void *host_out);
int cb_renderer_write_rng_states(cb_renderer *r, const void *host_in);
/* Device pointer of the histogram, for a caller-side RCCL reduce. */
cb_pixel *cb_renderer_device_histogram(cb_renderer *r);
/* The one exchange of the multi-GPU path (the reference has no multi-GPU code): rank r
 * renders with first_subsequence = r * n_threads on its own device, then renderers[0] += renderers[1..n)
 * -- one ncclReduce(ncclUint64, ncclSum, root 0) over xGMI when the renderers sit on n distinct devices
 * (RCCL is loaded on first use), an add kernel when they all share one (rehearsal on a one-GPU box).
 * Finishes carried work of every renderer first. */
int cb_renderers_reduce(cb_renderer *const *renderers, int n);
/* CleanupGlobals (cudabrot.cu:112-119). */
void cb_renderer_destroy(cb_renderer *r);

/* ---- Output stage (host side of the reference) -------------------------------------------------- */

/* SetGrayscalePixels (cudabrot.cu:425-468) on a host histogram: gray_out receives w*h host-endian
 * uint16; *max_out and *scale_out are the two numbers of the "Max value" line (cudabrot.cu:437). */
void cb_set_grayscale_pixels(const cb_pixel *hist, int w, int h, double gamma, uint16_t *gray_out,
                             uint64_t *max_out, double *scale_out);
/* SaveImage (cudabrot.cu:548-577): byte-swaps gray in place and writes the binary PGM.  Returns 0,
 * or 1/2/3 = open / header / pixel-data failure (the reference prints and carries on). */
int cb_save_image(const char *path, uint16_t *gray, int w, int h);

/* The value of one pixel under SetGrayscalePixels: count scaled by 65535/max, gamma-corrected
 * (cudabrot.cu:443-449) or plainly scaled when gamma <= 0 (cudabrot.cu:462-466), as uint16. */
uint16_t cb_tone_value(uint64_t count, uint64_t max, double gamma);
/* SaveImage for pixels that are already big-endian (the output of the device tone map). */
int cb_save_image_be(const char *path, const uint16_t *gray_be, int w, int h);

/* ---- Output stage on the device ---------------------------------------------- */

#define CB_TONE_AUTO 0        /* table when max < 2^24, thresholds above */
#define CB_TONE_LUT 1         /* host-evaluated table over every count in [0, max] */
#define CB_TONE_THRESHOLDS 2  /* host-evaluated smallest count per output value + binary search */

/* SetGrayscalePixels (cudabrot.cu:425-468) + the byte swap of SaveImage (cudabrot.cu:566-570) on a
 * DEVICE histogram: d_gray_be receives w*h big-endian uint16 (the PGM body).  The map itself is
 * evaluated by the host (cb_tone_value) into a table the device looks up, so the bytes equal
 * cb_set_grayscale_pixels + cb_save_image's.  Synchronises `stream`. */
int cb_tone_map_device(const cb_pixel *d_hist, int w, int h, double gamma, int mode,
                       uint16_t *d_gray_be, uint64_t *max_out, double *scale_out, void *stream);
/* The same for a renderer's histogram (finishes carried work first); host_gray_be receives the
 * w*h big-endian pixels: 2 bytes per pixel cross PCIe instead of 8. */
int cb_renderer_grayscale_image(cb_renderer *r, double gamma, int mode, uint16_t *host_gray_be,
                                uint64_t *max_out, double *scale_out);

/* ... for plane `plane` of a multi-channel renderer. */
int cb_renderer_grayscale_plane(cb_renderer *r, int plane, double gamma, int mode, uint16_t *host_gray_be,
                                uint64_t *max_out, double *scale_out);

const char *cb_error_string(int code);
int cb_abi_version(void);

/* Diagnostics.  The library's test and tuning knobs are environment variables named CUDABROT_AMD_* (DESIGN.md
 * section 7 lists them; none is needed in normal use, all are result-neutral).  They are read through this one
 * gate: the value of `name`, or NULL unless CUDABROT_AMD_DEBUG=1 is set too -- a stray variable in a user's
 * environment cannot change the path the product takes.  (The reference has no such knobs.) */
const char *cb_debug_knob(const char *name);
/* Which draw kernel the last cb_draw_buddhabrot* call of this process launched (the renderer's calls included):
 * 0 none yet, 1 draw_wave_kernel (four waves per SIMD), 2 draw_wide_kernel (two waves per SIMD, runs beside the
 * scatter), 3 the lock-step baseline.  The kernels give identical results; tests use this to know what they covered. */
int cb_debug_last_draw_kernel(void);
/* The level of the interior map the last cb_draw_buddhabrot call of this process used (cells of side 2^-level of the
 * c-plane whose samples provably never escape: the draw kernel retires them without iterating; made and proven by
 * tools/interior_map.c, embedded in the library), 0 if it used none.  Results do not depend on it. */
int cb_debug_interior_map_level(void);
/* The same for ONE renderer's last launch (with several ranks in a process the process-wide figure above is whichever
 * rank's launch came last; the map itself is copied to every device once per process, on its first launch there). */
int cb_renderer_interior_map_level(const cb_renderer *r);

#ifdef __cplusplus
}
#endif
#endif
