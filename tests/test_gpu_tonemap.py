"""Device tone map: SetGrayscalePixels + the byte swap of SaveImage on the device
(cudabrot.cu:425-468, 566-570) must give the bytes of the host path, for both table forms."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _device_tone_map(cb, hist, gamma, mode):
    import torch

    h, w = hist.shape
    dev = torch.device("cuda:0")
    d_hist = torch.from_numpy(hist.astype(np.uint64).view(np.int64).reshape(-1)).to(dev)
    d_gray = torch.zeros(h * w, dtype=torch.int16, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    mx, scale = cb.tone_map_device(d_hist.data_ptr(), w, h, gamma, d_gray.data_ptr(), mode=mode, stream=stream)
    torch.cuda.synchronize()
    body = d_gray.cpu().numpy().view(">u2").reshape(h, w)   # big-endian samples = the PGM body
    return body, mx, scale


def _synthetic_histogram(seed, h, w, top):
    rng = np.random.default_rng(seed)
    # heavy-tailed like a Buddhabrot: mostly small counts, a few near the maximum
    a = (rng.pareto(1.2, size=(h, w)) * 3).astype(np.uint64)
    a = np.minimum(a, np.uint64(top))
    a[rng.integers(0, h), rng.integers(0, w)] = top
    a[0, :7] = np.arange(7, dtype=np.uint64)
    return a


@pytest.mark.parametrize("gamma", [1.0, 2.2, 0.45, 0.0, -3.0])
@pytest.mark.parametrize("mode_name", ["CB_TONE_LUT", "CB_TONE_THRESHOLDS", "CB_TONE_AUTO"])
def test_device_tone_map_equals_host_path(cb, oracle, gamma, mode_name):
    hist = _synthetic_histogram(7, 201, 333, 1_000_003)      # odd pixel count: the last pixel is a lone store
    body, mx, scale = _device_tone_map(cb, hist, gamma, getattr(cb, mode_name))
    gray, mx_ref, scale_ref = oracle.set_grayscale_pixels(hist, gamma)
    assert (mx, scale) == (mx_ref, scale_ref)
    assert np.array_equal(body.astype(np.uint16), gray)


def test_counts_beyond_the_table_limit_use_thresholds(cb, oracle):
    """max >= 2^24: CB_TONE_AUTO switches to the threshold table; counts beyond 32 bits are fine."""
    hist = _synthetic_histogram(11, 64, 96, 1_000_000)
    hist = hist * np.uint64(40_000)                            # max = 4e10
    hist[5, 5] = (1 << 37) + 12345
    for gamma in (1.0, 2.2):
        body, mx, _ = _device_tone_map(cb, hist, gamma, cb.CB_TONE_AUTO)
        gray, mx_ref, _ = oracle.set_grayscale_pixels(hist, gamma)
        assert mx == mx_ref == (1 << 37) + 12345
        assert np.array_equal(body.astype(np.uint16), gray)
    with pytest.raises(cb.CudabrotError):                      # a 2^37-entry table is refused, not attempted
        _device_tone_map(cb, hist, 1.0, cb.CB_TONE_LUT)


def test_empty_histogram_maps_to_zero(cb):
    z = np.zeros((9, 13), dtype=np.uint64)
    for mode in (cb.CB_TONE_LUT, cb.CB_TONE_THRESHOLDS):
        body, mx, scale = _device_tone_map(cb, z, 1.0, mode)
        assert mx == 0 and scale == float("inf") and not body.any()


def test_renderer_grayscale_image_equals_host_tone_map_of_its_histogram(cb, oracle):
    dims = cb.FractalDimensions.make(320, 240)
    with cb.Renderer(dims, cb.IterationControl(300, 20), n_threads=8192) as r:
        r.render_passes(3)
        body, mx, scale = r.grayscale_image(2.2)                # finishes carried work itself
        hist = r.read_histogram()
    gray, mx_ref, scale_ref = oracle.set_grayscale_pixels(hist, 2.2)
    assert (mx, scale) == (mx_ref, scale_ref)
    assert np.array_equal(body.astype(np.uint16), gray)
    ref_hist, _ = oracle.render(320, 240, 300, 20, 8192, 3, omp_threads=0)
    assert np.array_equal(hist, ref_hist)
