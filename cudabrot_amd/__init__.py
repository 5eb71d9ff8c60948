"""cudabrot_amd -- Python host-side mirror of the MI355X-native Buddhabrot hot path.

The product is native: ``libcudabrot_amd.so`` (hand-written gfx950 HIP kernels behind the C ABI of
``include/cudabrot_amd.h``) and the ``cudabrot`` command-line binary.  This package only binds that
C ABI with ctypes so that tests and ``bench.py`` can drive it; it contains no arithmetic of its own and
no CPU fallback -- if the shared library is missing, importing :mod:`cudabrot_amd.capi` raises.
"""

from .capi import (  # noqa: F401
    CB_DEFAULT_RNG_SEED,
    CB_DEFAULT_THREADS,
    CB_KERNEL_DEFAULT,
    CB_KERNEL_FLAG_BURNING_SHIP,
    CB_KERNEL_FULL_ITERATE,
    CB_KERNEL_SIMPLE,
    CB_KERNEL_TIMED,
    CB_SAMPLES_PER_THREAD,
    CB_TONE_AUTO,
    CB_TONE_LUT,
    CB_TONE_THRESHOLDS,
    Counters,
    CudabrotError,
    FractalDimensions,
    IterationControl,
    Renderer,
    carry_bytes,
    draw_buddhabrot,
    flush_scatter,
    initialize_rng,
    lib,
    library_path,
    recompute_pixel_deltas,
    rng_state_bytes,
    save_image,
    scatter_workspace_bytes,
    set_grayscale_pixels,
    tone_map_device,
    tone_value,
)
from .sharding import shard_subsequences  # noqa: F401

__all__ = [
    "CB_DEFAULT_RNG_SEED",
    "CB_DEFAULT_THREADS",
    "CB_KERNEL_DEFAULT",
    "CB_KERNEL_FLAG_BURNING_SHIP",
    "CB_KERNEL_FULL_ITERATE",
    "CB_KERNEL_SIMPLE",
    "CB_KERNEL_TIMED",
    "CB_SAMPLES_PER_THREAD",
    "CB_TONE_AUTO",
    "CB_TONE_LUT",
    "CB_TONE_THRESHOLDS",
    "Counters",
    "CudabrotError",
    "FractalDimensions",
    "IterationControl",
    "Renderer",
    "carry_bytes",
    "draw_buddhabrot",
    "flush_scatter",
    "initialize_rng",
    "lib",
    "library_path",
    "recompute_pixel_deltas",
    "rng_state_bytes",
    "save_image",
    "scatter_workspace_bytes",
    "set_grayscale_pixels",
    "shard_subsequences",
    "tone_map_device",
    "tone_value",
]
