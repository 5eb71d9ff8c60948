"""The C-ABI library without a GPU: it loads, exports every declared symbol, and its host-side entry
points (canvas validation, tone map, PGM writer) match the oracle.  No device compute here."""

import ctypes as C
import os
import re

import numpy as np
import pytest


def declared_symbols(repo_root):
    text = open(os.path.join(repo_root, "include", "cudabrot_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cb_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(cb, repo_root):
    names = declared_symbols(repo_root)
    assert len(names) >= 15
    raw = C.CDLL(cb.library_path())
    for n in names:
        assert hasattr(raw, n), "libcudabrot_amd.so does not export %s" % n
    assert cb.lib.cb_abi_version() == 1


def test_knobs_are_read_only_behind_the_debug_gate(cb, monkeypatch):
    """A stray CUDABROT_AMD_* variable must not change the product's path: every knob goes through cb_debug_knob,
    which answers only when CUDABROT_AMD_DEBUG=1 is set as well."""
    monkeypatch.setenv("CUDABROT_AMD_NO_WORKSPACE", "1")
    monkeypatch.delenv("CUDABROT_AMD_DEBUG", raising=False)
    assert cb.lib.cb_debug_knob(b"CUDABROT_AMD_NO_WORKSPACE") is None
    monkeypatch.setenv("CUDABROT_AMD_DEBUG", "0")
    assert cb.lib.cb_debug_knob(b"CUDABROT_AMD_NO_WORKSPACE") is None
    monkeypatch.setenv("CUDABROT_AMD_DEBUG", "1")
    assert cb.lib.cb_debug_knob(b"CUDABROT_AMD_NO_WORKSPACE") == b"1"
    assert cb.lib.cb_debug_knob(b"CUDABROT_AMD_TWO_LEVEL") is None
    # and no product source reads the environment any other way
    import glob
    root = os.path.dirname(os.path.dirname(cb.library_path()))
    for f in glob.glob(os.path.join(root, "cudabrot_amd", "csrc", "*.*")):
        text = open(f).read()
        if f.endswith("host_abi.cpp"):
            continue
        assert "getenv(" not in text, f


def test_struct_layouts_match_the_reference_kernel_arguments(cb):
    # FractalDimensions is 56 bytes, IterationControl 8 (cudabrot.cu:46-67)
    assert C.sizeof(cb.FractalDimensions) == 56
    assert C.sizeof(cb.IterationControl) == 8
    assert C.sizeof(cb.Counters) == 136
    assert cb.FractalDimensions.delta_real.offset == 40


def test_rng_state_bytes(cb):
    assert cb.rng_state_bytes(0) == 0
    assert cb.rng_state_bytes(512 * 512) == 512 * 512 * 24


def test_recompute_pixel_deltas_messages(cb):
    """cudabrot.cu:505-527, messages verbatim (including the reference's swapped wording for imag)."""
    d = cb.FractalDimensions(1000, 1000, -2.0, -2.0, 2.0, 2.0, 0, 0)
    assert cb.recompute_pixel_deltas(d) == (True, None)
    assert d.delta_real == 4.0 / 1000 and d.delta_imag == 4.0 / 1000
    d = cb.FractalDimensions(0, 10, -2.0, -2.0, 2.0, 2.0, 0, 0)
    assert cb.recompute_pixel_deltas(d) == (False, "Output width must be positive.")
    d = cb.FractalDimensions(10, -1, -2.0, -2.0, 2.0, 2.0, 0, 0)
    assert cb.recompute_pixel_deltas(d) == (False, "Output height must be positive.")
    d = cb.FractalDimensions(10, 10, 1.0, -2.0, 1.0, 2.0, 0, 0)
    assert cb.recompute_pixel_deltas(d) == (False, "Maximum real value must be greater than minimum real value.")
    d = cb.FractalDimensions(10, 10, -2.0, 3.0, 2.0, 2.0, 0, 0)
    assert cb.recompute_pixel_deltas(d) == (
        False, "Minimum imaginary value must be greater than maximum imaginary value.")
    d = cb.FractalDimensions.make(20000, 15000, -2.0, 2.0, -1.5, 1.5)  # generate_hires_color_image.sh
    assert d.delta_real == 4.0 / 20000 and d.delta_imag == 3.0 / 15000


def test_invalid_arguments_are_rejected_before_any_device_work(cb):
    d = cb.FractalDimensions.make(16, 16)
    it = cb.IterationControl(100, 20)
    with pytest.raises(cb.CudabrotError):
        cb.draw_buddhabrot(d, 0, it, 0, 64, 50)  # null device pointers
    with pytest.raises(cb.CudabrotError):
        cb.Renderer(d, it, n_threads=0)


@pytest.mark.parametrize("gamma", [1.0, 2.2, 0.45, 0.0, -3.0])
def test_set_grayscale_pixels_matches_oracle(cb, oracle, gamma):
    hist, _ = oracle.render(150, 100, 300, 20, 2000, 2, (-2.0, 1.0, -1.0, 1.0))
    hist[3, 4] += 1 << 33  # a count beyond 32 bits
    g1, m1, s1 = cb.set_grayscale_pixels(hist, gamma)
    g2, m2, s2 = oracle.set_grayscale_pixels(hist, gamma)
    assert (m1, s1) == (m2, s2)
    assert np.array_equal(g1, g2)
    assert g1.max() == 65535 or gamma <= 0


def test_tone_value_is_the_per_pixel_map_of_set_grayscale_pixels(cb, oracle):
    """cb_tone_value feeds the device tone map's tables: it must be SetGrayscalePixels for one pixel."""
    counts = np.array([[0, 1, 2, 3, 10, 999, 1000, 65535, 65536, 10**6 - 1, 10**6]], dtype=np.uint64)
    for gamma in (1.0, 2.2, 0.45, 0.0, -3.0):
        ref = oracle.set_grayscale_pixels(counts, gamma)[0].reshape(-1)
        got = [cb.tone_value(int(c), 10**6, gamma) for c in counts.reshape(-1)]
        assert got == ref.tolist()
    assert cb.tone_value(0, 0, 1.0) == 0  # scale = inf, 0 * inf = NaN -> 0


def test_save_image_be_writes_swapped_pixels_unchanged(cb, oracle, tmp_path):
    gray = (np.arange(23 * 7, dtype=np.uint32) * 2777 % 65536).astype(np.uint16).reshape(7, 23)
    path = str(tmp_path / "be.pgm")
    be = np.ascontiguousarray(gray.astype(">u2"))
    assert cb.lib.cb_save_image_be(path.encode(), be.ctypes.data, 23, 7) == 0
    assert open(path, "rb").read() == oracle.encode_pgm(gray)


def test_set_grayscale_pixels_empty_histogram(cb, oracle):
    """max == 0 -> scale = inf (cudabrot.cu:436); every pixel comes out 0."""
    z = np.zeros((8, 8), dtype=np.uint64)
    for gamma in (1.0, -1.0):
        g, m, s = cb.set_grayscale_pixels(z, gamma)
        assert m == 0 and s == float("inf") and not g.any()
        assert np.array_equal(g, oracle.set_grayscale_pixels(z, gamma)[0])


def test_save_image_bytes(cb, oracle, tmp_path):
    gray = (np.arange(37 * 11, dtype=np.uint32) * 1777 % 65536).astype(np.uint16).reshape(11, 37)
    path = str(tmp_path / "x.pgm")
    assert cb.save_image(path, gray) == 0
    data = open(path, "rb").read()
    assert data.startswith(b"P5\n37 11\n65535\n")
    assert data == oracle.encode_pgm(gray)
    body = np.frombuffer(data[len(b"P5\n37 11\n65535\n"):], dtype=">u2").reshape(11, 37)
    assert np.array_equal(body, gray)
    assert cb.save_image(str(tmp_path / "no_such_dir" / "x.pgm"), gray) == 1  # open failure is reported


def test_header_is_plain_c_and_the_library_links_from_c(cb, repo_root, tmp_path):
    """include/cudabrot_amd.h is the drop-in boundary for a C host program (the reference is C): it must compile
    as C99 with warnings on, and a C program must link against the shared library."""
    import shutil
    import subprocess

    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text(
        '#include "cudabrot_amd.h"\n#include <stdio.h>\n'
        "int main(void) {\n"
        "  cb_fractal_dimensions d = {0};\n  const char *msg = 0;\n"
        "  d.w = 10; d.h = 10; d.min_real = -2; d.max_real = 2; d.min_imag = -2; d.max_imag = 2;\n"
        '  printf("%d %d %d %d\\n", cb_recompute_pixel_deltas(&d, &msg), (int) sizeof(cb_counters),\n'
        "         (int) cb_rng_state_bytes(64), (int) cb_tone_value(5, 10, 1.0));\n  return 0;\n}\n")
    libdir = os.path.dirname(cb.library_path())
    exe = str(tmp_path / "t")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(repo_root, "include"),
                        str(src), "-L", libdir, "-lcudabrot_amd", "-Wl,-rpath," + libdir, "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.split() == ["1", "136", "1536", "32767"]


def test_every_entry_point_rejects_null_or_nonsense_arguments(cb):
    """The C ABI returns an error code (never crashes, never touches the device) for null handles and
    pointers, empty canvases, zero thread counts and channel counts out of range (cudabrot_amd.h: every
    function returns 0 or a hipError_t value)."""
    import ctypes as C

    lib = cb.capi.lib
    d = cb.FractalDimensions.make(16, 16)
    empty = cb.FractalDimensions(0, 16, -2.0, -2.0, 2.0, 2.0, 0.25, 0.25)
    it = cb.IterationControl(100, 20)
    null = C.c_void_p(0)
    windows = (cb.IterationControl * 5)(*[cb.IterationControl(100, 20)] * 5)
    bad = [
        lib.cb_initialize_rng(1337, 0, 64, null, null),
        lib.cb_draw_buddhabrot(C.byref(d), null, C.byref(it), null, 64, 50, null, 0, null, 0, null, null),
        lib.cb_draw_buddhabrot(C.byref(empty), 8, C.byref(it), 8, 64, 50, null, 0, null, 0, null, null),
        lib.cb_draw_buddhabrot(C.byref(d), 8, C.byref(it), 8, 64, 50, null, 77, null, 0, null, null),  # no such kernel variant
        lib.cb_flush_scatter(C.byref(d), null, 64, null, 0, null),
        lib.cb_flush_scatter(C.byref(empty), 8, 64, null, 0, null),
        lib.cb_draw_buddhabrot_channels(C.byref(d), 8, windows, 0, 8, 64, 50, null, 0, null, 0, null, null),
        lib.cb_draw_buddhabrot_channels(C.byref(d), 8, windows, 5, 8, 64, 50, null, 0, null, 0, null, null),
        lib.cb_flush_scatter_channels(C.byref(d), 8, 0, 64, null, 0, null),
        lib.cb_flush_scatter_channels(C.byref(d), 8, 5, 64, null, 0, null),
        lib.cb_renderer_render_passes(null, 1, 0),
        lib.cb_renderer_prepare(null, 0),
        lib.cb_renderer_finish(null),
        lib.cb_renderer_read_histogram(null, null),
        lib.cb_renderer_write_histogram(null, null),
        lib.cb_renderer_read_counters(null, None),
        lib.cb_renderer_read_rng_states(null, null),
        lib.cb_renderer_write_rng_states(null, null),
        lib.cb_renderer_grayscale_image(null, 1.0, 0, null, None, None),
        lib.cb_renderer_grayscale_plane(null, 0, 1.0, 0, null, None, None),
        lib.cb_renderers_reduce(None, 2),
        lib.cb_tone_map_device(null, 16, 16, 1.0, 0, null, None, None, null),
    ]
    assert all(rc != 0 for rc in bad), bad
    out = C.c_void_p()
    assert lib.cb_renderer_create(C.byref(out), 0, C.byref(empty), C.byref(it), 1337, 0, 64) != 0 and not out.value
    assert lib.cb_renderer_create_channels(C.byref(out), 0, C.byref(d), windows, 5, 1337, 0, 64) != 0 and not out.value
    assert lib.cb_scatter_workspace_bytes(C.byref(empty), 64, 50) == 0
    assert lib.cb_scatter_workspace_bytes(C.byref(d), 0, 50) == 0
    assert lib.cb_renderer_device_histogram(null) in (None, 0)
    lib.cb_renderer_destroy(null)     # a no-op
    assert cb.capi._error_string(bad[0])   # every code has a name
