"""ctypes binding of include/cudabrot_amd.h -- one Python name per C entry point, nothing more.

Names follow the reference's own (cudabrot.cu): FractalDimensions (:46-58), IterationControl (:62-67),
InitializeRNG (:146-149) -> initialize_rng, DrawBuddhabrot (:379-414) -> draw_buddhabrot,
RecomputePixelDeltas (:505-527), SetGrayscalePixels (:454-468), SaveImage (:548-577); Renderer is
SetupCUDA + RenderImage (:153-189, :471-501) as an object.

There is no fallback path: the shared library must exist (``make`` or ``__graft_entry__.build()``) or
this module raises at import, and every device call raises :class:`CudabrotError` on a HIP error.
"""

import ctypes as C
import os

import numpy as np

CB_DEFAULT_THREADS = 512 * 512  # cudabrot.cu:20,23
CB_SAMPLES_PER_THREAD = 50  # cudabrot.cu:34
CB_DEFAULT_RNG_SEED = 1337  # cudabrot.cu:37
CB_KERNEL_DEFAULT = 0
CB_KERNEL_SIMPLE = 1
CB_KERNEL_TIMED = 2
CB_KERNEL_FULL_ITERATE = 3
CB_TONE_AUTO, CB_TONE_LUT, CB_TONE_THRESHOLDS = 0, 1, 2
CB_KERNEL_FLAG_BURNING_SHIP = 0x100
CB_KERNEL_FLAG_DRAIN = 0x200
# cb_counters.status bits (include/cudabrot_amd.h)
CB_STATUS_QUEUE_OVERFLOW, CB_STATUS_REPLAY_RUNAWAY, CB_STATUS_INTERIOR_MAP, CB_STATUS_CARRY_FOREIGN = 1, 2, 4, 8

_HERE = os.path.dirname(os.path.abspath(__file__))


def library_path():
    return os.path.join(_HERE, "libcudabrot_amd.so")


class CudabrotError(RuntimeError):
    """A C-ABI call returned a nonzero hipError_t."""

    def __init__(self, code, what):
        self.code = int(code)
        super().__init__("%s failed: HIP error %d (%s)" % (what, self.code, _error_string(self.code)))


class FractalDimensions(C.Structure):
    """cb_fractal_dimensions == FractalDimensions (cudabrot.cu:46-58)."""

    _fields_ = [
        ("w", C.c_int),
        ("h", C.c_int),
        ("min_real", C.c_double),
        ("min_imag", C.c_double),
        ("max_real", C.c_double),
        ("max_imag", C.c_double),
        ("delta_real", C.c_double),
        ("delta_imag", C.c_double),
    ]

    @classmethod
    def make(cls, w, h, min_real=-2.0, max_real=2.0, min_imag=-2.0, max_imag=2.0):
        d = cls(w, h, min_real, min_imag, max_real, max_imag, 0.0, 0.0)
        ok, msg = recompute_pixel_deltas(d)
        if not ok:
            raise ValueError(msg)
        return d


class IterationControl(C.Structure):
    """cb_iteration_control == IterationControl (cudabrot.cu:62-67)."""

    _fields_ = [("max_escape_iterations", C.c_int), ("min_escape_iterations", C.c_int)]


class Counters(C.Structure):
    """cb_counters: exact device-side workload counters."""

    _fields_ = [
        (n, C.c_uint64)
        for n in (
            "samples",
            "rejected",
            "never_escaped",
            "too_fast",
            "recorded",
            "iterate_steps",
            "replay_steps",
            "increments",
            "skipped_steps",
            "status",
            "cycles_head",
            "cycles_long",
            "cycles_replay",
            "cycles_total",
            "rt_not_first_start",
            "rt_last_end",
            "rt_wave_life_sum",
        )
    ]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def _share_torch_hip_runtime():
    """One HIP runtime per process.

    The PyTorch wheel bundles its own libamdhip64.so (SONAME libamdhip64.so.7) and links it by the
    unversioned file name, so a process that loads /opt/rocm's copy through this library and then
    uses torch.cuda ends up with two HIP/HSA runtimes, and the second one finds no GPU.  Loading
    torch's copy first (globally) makes this library's DT_NEEDED ``libamdhip64.so.7`` resolve to it,
    and a later ``import torch`` finds the same file.  Without torch installed (or with
    CUDABROT_AMD_SYSTEM_HIP=1) the system runtime is used, as the `cudabrot` binary always does.
    """
    if os.environ.get("CUDABROT_AMD_SYSTEM_HIP") == "1":
        return None
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        return None
    C.CDLL(cand, mode=C.RTLD_GLOBAL)
    return cand


def _load():
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            "cudabrot_amd: %s is missing -- build it with `make` (or __graft_entry__.build()); "
            "there is no CPU fallback" % path
        )
    _share_torch_hip_runtime()
    lib_ = C.CDLL(path)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    dims_p, it_p, cnt_p = C.POINTER(FractalDimensions), C.POINTER(IterationControl), C.POINTER(Counters)
    sigs = {
        "cb_abi_version": (i32, []),
        "cb_error_string": (C.c_char_p, [i32]),
        "cb_debug_knob": (C.c_char_p, [C.c_char_p]),
        "cb_debug_last_draw_kernel": (i32, []),
        "cb_debug_interior_map_level": (i32, []),
        "cb_renderer_interior_map_level": (i32, [vp]),
        "cb_recompute_pixel_deltas": (i32, [dims_p, C.POINTER(C.c_char_p)]),
        "cb_rng_state_bytes": (C.c_size_t, [u32]),
        "cb_initialize_rng": (i32, [u64, u64, u32, vp, vp]),
        "cb_scatter_workspace_bytes": (C.c_size_t, [dims_p, u32, u32]),
        "cb_scatter_workspace_bytes_channels": (C.c_size_t, [dims_p, i32, u32, u32]),
        "cb_draw_buddhabrot": (i32, [dims_p, vp, it_p, vp, u32, u32, vp, i32, vp, C.c_size_t, vp, vp]),
        "cb_carry_bytes": (C.c_size_t, [u32]),
        "cb_draw_buddhabrot_channels": (i32, [dims_p, vp, it_p, i32, vp, u32, u32, vp, i32, vp, C.c_size_t, vp, vp]),
        "cb_flush_scatter_channels": (i32, [dims_p, vp, i32, u32, vp, C.c_size_t, vp]),
        "cb_renderer_finish": (i32, [vp]),
        "cb_flush_scatter": (i32, [dims_p, vp, u32, vp, C.c_size_t, vp]),
        "cb_renderer_create": (i32, [C.POINTER(vp), i32, dims_p, it_p, u64, u64, u32]),
        "cb_renderer_create_channels": (i32, [C.POINTER(vp), i32, dims_p, it_p, i32, u64, u64, u32]),
        "cb_renderer_grayscale_plane": (i32, [vp, i32, C.c_double, i32, vp, C.POINTER(u64), C.POINTER(C.c_double)]),
        "cb_renderer_render_passes": (i32, [vp, u32, i32]),
        "cb_renderer_prepare": (i32, [vp, i32]),
        "cb_renderer_read_histogram": (i32, [vp, vp]),
        "cb_renderer_write_histogram": (i32, [vp, vp]),
        "cb_renderer_read_counters": (i32, [vp, cnt_p]),
        "cb_renderer_read_rng_states": (i32, [vp, vp]),
        "cb_renderer_write_rng_states": (i32, [vp, vp]),
        "cb_renderer_device_histogram": (vp, [vp]),
        "cb_renderers_reduce": (i32, [C.POINTER(vp), i32]),
        "cb_renderer_destroy": (None, [vp]),
        "cb_set_grayscale_pixels": (None, [vp, i32, i32, C.c_double, vp, C.POINTER(u64), C.POINTER(C.c_double)]),
        "cb_save_image": (i32, [C.c_char_p, vp, i32, i32]),
        "cb_save_image_be": (i32, [C.c_char_p, vp, i32, i32]),
        "cb_tone_value": (C.c_uint16, [u64, u64, C.c_double]),
        "cb_tone_map_device": (i32, [vp, i32, i32, C.c_double, i32, vp, C.POINTER(u64), C.POINTER(C.c_double), vp]),
        "cb_renderer_grayscale_image": (i32, [vp, C.c_double, i32, vp, C.POINTER(u64), C.POINTER(C.c_double)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib_, name)  # AttributeError here = the library does not export the ABI
        fn.restype = res
        fn.argtypes = args
    return lib_


lib = _load()
EXPORTED_SYMBOLS = (
    "cb_abi_version cb_error_string cb_debug_knob cb_debug_last_draw_kernel cb_debug_interior_map_level cb_renderer_interior_map_level cb_recompute_pixel_deltas cb_rng_state_bytes cb_initialize_rng "
    "cb_scatter_workspace_bytes cb_scatter_workspace_bytes_channels cb_carry_bytes cb_draw_buddhabrot cb_flush_scatter cb_renderer_create "
    "cb_renderer_render_passes cb_renderer_finish "
    "cb_renderer_read_histogram "
    "cb_renderer_write_histogram cb_renderer_read_counters cb_renderer_device_histogram "
    "cb_renderer_destroy cb_set_grayscale_pixels cb_save_image cb_save_image_be cb_tone_value "
    "cb_tone_map_device cb_renderer_grayscale_image cb_renderer_read_rng_states cb_renderer_write_rng_states "
    "cb_draw_buddhabrot_channels cb_flush_scatter_channels cb_renderer_create_channels cb_renderer_grayscale_plane cb_renderers_reduce "
    "cb_renderer_prepare"
).split()


def _error_string(code):
    return lib.cb_error_string(int(code)).decode()


def _check(code, what):
    if code != 0:
        raise CudabrotError(code, what)


def recompute_pixel_deltas(dims):
    """RecomputePixelDeltas (cudabrot.cu:505-527) -> (ok, message-or-None); fills dims.delta_*."""
    msg = C.c_char_p()
    ok = lib.cb_recompute_pixel_deltas(C.byref(dims), C.byref(msg))
    return bool(ok), (None if ok else msg.value.decode())


def rng_state_bytes(n_threads):
    return int(lib.cb_rng_state_bytes(n_threads))


def initialize_rng(seed, first_subsequence, n_threads, d_states, stream=0):
    """InitializeRNG (cudabrot.cu:146-149,179) on caller-owned device memory (integer pointers)."""
    _check(lib.cb_initialize_rng(seed, first_subsequence, n_threads, d_states, stream), "cb_initialize_rng")


def scatter_workspace_bytes(dims, n_threads, samples_per_thread, n_channels=1):
    """Suggested scatter-workspace size for launches of this shape (0: the canvas cannot use one);
    n_channels > 1: for a fused multi-channel launch of that many planes."""
    return int(lib.cb_scatter_workspace_bytes_channels(C.byref(dims), n_channels, n_threads, samples_per_thread))


def carry_bytes(n_threads):
    """Size of the carry buffer (in-flight orbits handed from launch to launch)."""
    return int(lib.cb_carry_bytes(n_threads))


def draw_buddhabrot(dims, d_hist, iterations, d_states, n_threads, samples_per_thread, d_counters=0,
                    kernel_variant=CB_KERNEL_DEFAULT, stream=0, d_workspace=0, workspace_bytes=0, d_carry=0):
    """DrawBuddhabrot (cudabrot.cu:379-414,485-486) on caller-owned device memory; asynchronous.
    With a workspace the increments go through the deferred tile-binned scatter, else direct atomics.
    With a (zeroed) carry buffer, orbits still in flight are handed to the next call; a last call with
    samples_per_thread=0 completes them."""
    _check(
        lib.cb_draw_buddhabrot(C.byref(dims), d_hist, C.byref(iterations), d_states, n_threads,
                               samples_per_thread, d_counters, kernel_variant, d_workspace, workspace_bytes,
                               d_carry, stream),
        "cb_draw_buddhabrot",
    )


def flush_scatter(dims, d_hist, n_threads, d_workspace, workspace_bytes, stream=0):
    """Adds the pixel stream a draw_buddhabrot call deferred into the workspace to the histogram."""
    _check(lib.cb_flush_scatter(C.byref(dims), d_hist, n_threads, d_workspace, workspace_bytes, stream),
           "cb_flush_scatter")


class Renderer:
    """SetupCUDA + RenderImage (cudabrot.cu:153-189, 471-501) over the C ABI's cb_renderer."""

    def __init__(self, dims, iterations, device=0, seed=CB_DEFAULT_RNG_SEED, first_subsequence=0,
                 n_threads=CB_DEFAULT_THREADS):
        """iterations: an IterationControl, or -- fused multi-channel render (N2) -- a list of (max, min)
        windows; read_histogram then returns one plane per window ([k, h, w])."""
        self.dims = dims
        self.iterations = iterations
        self.n_threads = n_threads
        self._h = C.c_void_p()
        if isinstance(iterations, IterationControl):
            self.n_channels = 0
            _check(
                lib.cb_renderer_create(C.byref(self._h), device, C.byref(dims), C.byref(iterations), seed,
                                       first_subsequence, n_threads),
                "cb_renderer_create",
            )
        else:
            self.n_channels = len(iterations)
            arr = (IterationControl * len(iterations))(*[IterationControl(int(m), int(c)) for m, c in iterations])
            _check(
                lib.cb_renderer_create_channels(C.byref(self._h), device, C.byref(dims), arr, len(iterations), seed,
                                                first_subsequence, n_threads),
                "cb_renderer_create_channels",
            )

    def prepare(self, kernel_variant=CB_KERNEL_DEFAULT):
        """Allocate now what the first render_passes would (the scatter workspaces)."""
        _check(lib.cb_renderer_prepare(self._h, kernel_variant), "cb_renderer_prepare")

    def render_passes(self, passes, kernel_variant=CB_KERNEL_DEFAULT):
        _check(lib.cb_renderer_render_passes(self._h, passes, kernel_variant), "cb_renderer_render_passes")

    def finish(self):
        """Complete the orbits carried between launches (the read functions do this themselves)."""
        _check(lib.cb_renderer_finish(self._h), "cb_renderer_finish")

    def read_histogram(self):
        planes = self.n_channels or 1
        out = np.empty(planes * self.dims.w * self.dims.h, dtype=np.uint64)
        _check(lib.cb_renderer_read_histogram(self._h, out.ctypes.data), "cb_renderer_read_histogram")
        if self.n_channels:
            return out.reshape(planes, self.dims.h, self.dims.w)
        return out.reshape(self.dims.h, self.dims.w)

    def read_rng_states(self):
        """True-resume checkpoint (N3): the generator states as bytes (six u32 planes of n_threads)."""
        out = np.empty(rng_state_bytes(self.n_threads), dtype=np.uint8)
        _check(lib.cb_renderer_read_rng_states(self._h, out.ctypes.data), "cb_renderer_read_rng_states")
        return out

    def write_rng_states(self, blob):
        a = np.ascontiguousarray(blob, dtype=np.uint8).reshape(-1)
        if a.size != rng_state_bytes(self.n_threads):
            raise ValueError("generator state blob does not match n_threads")
        _check(lib.cb_renderer_write_rng_states(self._h, a.ctypes.data), "cb_renderer_write_rng_states")

    def grayscale_image(self, gamma, mode=0, plane=0):
        """Device tone map (N1) -> (big-endian u16 image [h,w] = the PGM body, max count, scale)."""
        gray = np.empty((self.dims.h, self.dims.w), dtype=">u2")
        mx, scale = C.c_uint64(), C.c_double()
        _check(
            lib.cb_renderer_grayscale_plane(self._h, int(plane), float(gamma), int(mode), gray.ctypes.data,
                                            C.byref(mx), C.byref(scale)),
            "cb_renderer_grayscale_plane",
        )
        return gray, int(mx.value), float(scale.value)

    def write_histogram(self, hist):
        a = np.ascontiguousarray(hist, dtype=np.uint64).reshape(-1)
        if a.size != (self.n_channels or 1) * self.dims.w * self.dims.h:
            raise ValueError("histogram size does not match the canvas")
        _check(lib.cb_renderer_write_histogram(self._h, a.ctypes.data), "cb_renderer_write_histogram")

    def read_counters(self):
        c = Counters()
        _check(lib.cb_renderer_read_counters(self._h, C.byref(c)), "cb_renderer_read_counters")
        return c

    @property
    def device_histogram(self):
        return lib.cb_renderer_device_histogram(self._h)

    def close(self):
        if self._h:
            lib.cb_renderer_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def set_grayscale_pixels(hist, gamma):
    """SetGrayscalePixels (cudabrot.cu:454-468) -> (u16 image [h,w] host-endian, max count, scale)."""
    a = np.ascontiguousarray(hist, dtype=np.uint64)
    h, w = a.shape
    gray = np.empty((h, w), dtype=np.uint16)
    mx, scale = C.c_uint64(), C.c_double()
    lib.cb_set_grayscale_pixels(a.ctypes.data, w, h, float(gamma), gray.ctypes.data, C.byref(mx), C.byref(scale))
    return gray, int(mx.value), float(scale.value)


def draw_buddhabrot_channels(dims, d_hist, windows, d_states, n_threads, samples_per_thread, d_counters=0,
                             kernel_variant=CB_KERNEL_DEFAULT, stream=0, d_workspace=0, workspace_bytes=0, d_carry=0):
    """Fused multi-channel launch (N2): windows = [(max_iter, min_iter), ...]; d_hist = len(windows) planes."""
    arr = (IterationControl * len(windows))(*[IterationControl(int(m), int(c)) for m, c in windows])
    _check(
        lib.cb_draw_buddhabrot_channels(C.byref(dims), d_hist, arr, len(windows), d_states, n_threads,
                                        samples_per_thread, d_counters, kernel_variant, d_workspace, workspace_bytes,
                                        d_carry, stream),
        "cb_draw_buddhabrot_channels",
    )


def flush_scatter_channels(dims, d_hist, n_channels, n_threads, d_workspace, workspace_bytes, stream=0):
    _check(
        lib.cb_flush_scatter_channels(C.byref(dims), d_hist, n_channels, n_threads, d_workspace, workspace_bytes, stream),
        "cb_flush_scatter_channels",
    )


def renderers_reduce(renderers):
    """renderers[0] += renderers[1:] (cb_renderers_reduce: RCCL across devices, an add kernel on one device)."""
    arr = (C.c_void_p * len(renderers))(*[r._h for r in renderers])
    _check(lib.cb_renderers_reduce(arr, len(renderers)), "cb_renderers_reduce")


def tone_value(count, max_count, gamma):
    """One pixel of SetGrayscalePixels (cudabrot.cu:443-449,462-466)."""
    return int(lib.cb_tone_value(int(count), int(max_count), float(gamma)))


def tone_map_device(d_hist, w, h, gamma, d_gray_be, mode=0, stream=0):
    """cb_tone_map_device on caller-owned device memory (integer pointers) -> (max count, scale)."""
    mx, scale = C.c_uint64(), C.c_double()
    _check(
        lib.cb_tone_map_device(d_hist, w, h, float(gamma), int(mode), d_gray_be, C.byref(mx), C.byref(scale), stream),
        "cb_tone_map_device",
    )
    return int(mx.value), float(scale.value)


def save_image(path, gray):
    """SaveImage (cudabrot.cu:548-577); ``gray`` is copied (the C call byte-swaps in place)."""
    g = np.array(gray, dtype=np.uint16, order="C")
    h, w = g.shape
    return int(lib.cb_save_image(os.fsencode(path), g.ctypes.data, w, h))
