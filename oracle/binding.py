"""ctypes binding of oracle/liboracle.so (buddha_oracle.h) -- TEST INFRASTRUCTURE ONLY."""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_DIR = os.path.join(_HERE, "_ref")


class Dims(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("min_real", C.c_double), ("min_imag", C.c_double),
                ("max_real", C.c_double), ("max_imag", C.c_double), ("delta_real", C.c_double),
                ("delta_imag", C.c_double)]


class Iters(C.Structure):
    _fields_ = [("max_escape_iterations", C.c_int), ("min_escape_iterations", C.c_int)]


class Xorwow(C.Structure):
    _fields_ = [("d", C.c_uint32), ("x", C.c_uint32 * 5)]


COUNTER_NAMES = ("samples", "rejected", "never_escaped", "too_fast", "recorded", "iterate_steps",
                 "replay_steps", "increments")


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_NAMES]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_NAMES}


def build(force=False):
    """Compile liboracle.so (gcc) if missing."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)


def _load():
    build()
    lib_ = C.CDLL(LIB_PATH)
    u64, vp, i32, dbl = C.c_uint64, C.c_void_p, C.c_int, C.c_double
    lib_.orc_recompute_pixel_deltas.restype = i32
    lib_.orc_recompute_pixel_deltas.argtypes = [C.POINTER(Dims)]
    lib_.orc_xorwow_init.argtypes = [u64, u64, u64, C.POINTER(Xorwow)]
    lib_.orc_xorwow_init_range.argtypes = [u64, u64, u64, vp]
    lib_.orc_xorwow_next.restype = C.c_uint32
    lib_.orc_xorwow_next.argtypes = [C.POINTER(Xorwow)]
    lib_.orc_uniform_double.restype = dbl
    lib_.orc_uniform_double.argtypes = [C.POINTER(Xorwow)]
    lib_.orc_xorwow_sequence_jump_matrix.argtypes = [i32, vp]
    lib_.orc_xorwow_jump_matrix.argtypes = [i32, vp]
    lib_.orc_in_main_cardioid.restype = i32
    lib_.orc_in_main_cardioid.argtypes = [dbl, dbl]
    lib_.orc_in_order2_bulb.restype = i32
    lib_.orc_in_order2_bulb.argtypes = [dbl, dbl]
    lib_.orc_iterate_mandelbrot.restype = i32
    lib_.orc_iterate_mandelbrot.argtypes = [dbl, dbl, i32]
    lib_.orc_iterate_and_record.restype = u64
    lib_.orc_iterate_and_record.argtypes = [C.POINTER(Dims), vp, dbl, dbl, C.POINTER(u64)]
    lib_.orc_draw_buddhabrot.argtypes = [C.POINTER(Dims), vp, C.POINTER(Iters), vp, u64, i32, C.POINTER(Counters)]
    lib_.orc_draw_buddhabrot_omp.restype = i32
    lib_.orc_draw_buddhabrot_omp.argtypes = [C.POINTER(Dims), vp, C.POINTER(Iters), vp, u64, i32,
                                             C.POINTER(Counters), i32]
    lib_.orc_fnv1a_pixels.restype = u64
    lib_.orc_fnv1a_pixels.argtypes = [vp, u64]
    lib_.orc_set_grayscale_pixels.restype = u64
    lib_.orc_set_grayscale_pixels.argtypes = [vp, i32, i32, dbl, vp, C.POINTER(dbl)]
    lib_.orc_encode_pgm.restype = C.c_size_t
    lib_.orc_encode_pgm.argtypes = [vp, i32, i32, vp]
    return lib_


lib = _load()

STATE_DTYPE = np.dtype([("d", "<u4"), ("x", "<u4", (5,))])


def make_dims(w, h, min_real=-2.0, max_real=2.0, min_imag=-2.0, max_imag=2.0):
    d = Dims(w, h, min_real, min_imag, max_real, max_imag, 0.0, 0.0)
    if not lib.orc_recompute_pixel_deltas(C.byref(d)):
        raise ValueError("invalid canvas")
    return d


def init_states(seed, first_subsequence, n_threads):
    """n_threads XORWOW states for subsequences first_subsequence.. (cudabrot.cu:148)."""
    st = np.zeros(n_threads, dtype=STATE_DTYPE)
    lib.orc_xorwow_init_range(seed, first_subsequence, n_threads, st.ctypes.data)
    return st


def rng_u32(seed, subsequence, n):
    st = Xorwow()
    lib.orc_xorwow_init(seed, subsequence, 0, C.byref(st))
    return [int(lib.orc_xorwow_next(C.byref(st))) for _ in range(n)]


def first_sample(seed, subsequence):
    st = Xorwow()
    lib.orc_xorwow_init(seed, subsequence, 0, C.byref(st))
    re = lib.orc_uniform_double(C.byref(st)) * 4.0 - 2.0
    im = lib.orc_uniform_double(C.byref(st)) * 4.0 - 2.0
    return re, im


def render(w, h, max_iter, min_iter, n_threads, passes, box=(-2.0, 2.0, -2.0, 2.0), first_subsequence=0,
           seed=1337, samples_per_thread=50, omp_threads=None, hist=None, states=None, burning_ship=False):
    """`passes` launches of DrawBuddhabrot over n_threads threads -> (u64 hist [h,w], counters dict).

    burning_ship: the reference's RENDER_BURNING_SHIP build (cudabrot.cu:15-17).
    box = (min_real, max_real, min_imag, max_imag).  omp_threads=None: sequential (the reference's
    race-free semantics); otherwise the OpenMP variant with that many workers (0 = all).
    """
    d = make_dims(w, h, box[0], box[1], box[2], box[3])
    it = Iters(max_iter, min_iter)
    st = init_states(seed, first_subsequence, n_threads) if states is None else states
    if hist is None:
        hist = np.zeros((h, w), dtype=np.uint64)
    cnt = Counters()
    lib.orc_set_burning_ship(1 if burning_ship else 0)
    try:
        for _ in range(passes):
            if omp_threads is None:
                lib.orc_draw_buddhabrot(C.byref(d), hist.ctypes.data, C.byref(it), st.ctypes.data, n_threads,
                                        samples_per_thread, C.byref(cnt))
            else:
                lib.orc_draw_buddhabrot_omp(C.byref(d), hist.ctypes.data, C.byref(it), st.ctypes.data, n_threads,
                                            samples_per_thread, C.byref(cnt), omp_threads)
    finally:
        lib.orc_set_burning_ship(0)
    return hist, cnt.as_dict()


def record_points(w, h, box, re, im):
    """IterateAndRecord (cudabrot.cu:347-365) for each given starting point, one after another ->
    (u64 hist [h,w], replay steps, increments).  Every point must escape (the caller checked)."""
    d = make_dims(w, h, box[0], box[1], box[2], box[3])
    hist = np.zeros((h, w), dtype=np.uint64)
    incr = C.c_uint64(0)
    steps = 0
    for a, b in zip(re, im):
        steps += int(lib.orc_iterate_and_record(C.byref(d), hist.ctypes.data, float(a), float(b), C.byref(incr)))
    return hist, steps, int(incr.value)


def fnv1a_pixels(hist):
    a = np.ascontiguousarray(hist, dtype=np.uint64)
    return int(lib.orc_fnv1a_pixels(a.ctypes.data, a.size))


def set_grayscale_pixels(hist, gamma):
    a = np.ascontiguousarray(hist, dtype=np.uint64)
    h, w = a.shape
    gray = np.empty((h, w), dtype=np.uint16)
    scale = C.c_double()
    mx = lib.orc_set_grayscale_pixels(a.ctypes.data, w, h, float(gamma), gray.ctypes.data, C.byref(scale))
    return gray, int(mx), float(scale.value)


def encode_pgm(gray):
    g = np.ascontiguousarray(gray, dtype=np.uint16)
    h, w = g.shape
    buf = np.empty(64 + 2 * w * h, dtype=np.uint8)
    n = lib.orc_encode_pgm(g.ctypes.data, w, h, buf.ctypes.data)
    return buf[:n].tobytes()


def ref_library(kind="fma"):
    """The reference's own lines compiled for the host (oracle/_ref/libref_<kind>.so), or None."""
    path = os.path.join(REF_DIR, "libref_%s.so" % kind)
    if not os.path.exists(path):
        return None
    r = C.CDLL(path)
    r.ref_draw.restype = C.c_int
    r.ref_draw.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                           C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_void_p]
    r.ref_rng_u32.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
    r.ref_first_sample.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    r.ref_iterate_mandelbrot.restype = C.c_int
    r.ref_iterate_mandelbrot.argtypes = [C.c_double, C.c_double, C.c_int]
    r.ref_in_set_shortcut.restype = C.c_int
    r.ref_in_set_shortcut.argtypes = [C.c_double, C.c_double]
    r.ref_set_grayscale_pixels.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
    r.ref_save_image.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
    return r
