"""The `cudabrot` binary end to end on a GPU: stdout sequence, PGM bytes, -s buffer semantics, SIGINT
(cudabrot.cu:215-280, 471-501, 548-577, 756-791)."""

import os
import re
import signal
import subprocess
import time

import numpy as np
import pytest

from conftest import STATE_HEADER_BYTES, read_state_file

pytestmark = pytest.mark.gpu

T = 512 * 512  # the CLI always runs the reference's 512 x 512 threads (cudabrot.cu:20,23)


@pytest.fixture(scope="module")
def exe(repo_root):
    path = os.path.join(repo_root, "cudabrot")
    assert os.access(path, os.X_OK), "./cudabrot is not built"
    return path


def run(exe, *args, **kw):
    return subprocess.run([exe, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, **kw)


@pytest.fixture(scope="module")
def small_render(oracle):
    """Oracle result of `--passes 2 -w 300 -h 200 -m 200` (all host cores; same multiset of samples)."""
    hist, cnt = oracle.render(300, 200, 200, 20, T, 2, omp_threads=0)
    return hist, cnt


def test_fixed_pass_run_matches_oracle_byte_for_byte(exe, oracle, small_render, tmp_path):
    out, buf = str(tmp_path / "o.pgm"), str(tmp_path / "state.bin")
    r = run(exe, "--passes", "2", "-w", "300", "-h", "200", "-m", "200", "-g", "2.2", "-o", out, "-s", buf)
    assert r.returncode == 0 and r.stderr == ""
    hist, _ = small_render
    gray, mx, scale = oracle.set_grayscale_pixels(hist, 2.2)
    lines = r.stdout.split("\n")
    # the reference's stdout sequence
    assert lines[0] == "Creating 300x200 image, 200 max iterations."
    assert lines[1] == "Calculating image..."
    assert lines[2].startswith("Approximate memory needed: ")
    assert lines[3] == "Loading previous image state from %s." % buf
    assert lines[4] == "File %s doesn't exist yet. Not loading." % buf
    assert lines[5] == "Calculating Buddhabrot."
    assert re.fullmatch(r"2 Buddhabrot passes took [0-9.]+ seconds\.", lines[6])
    assert lines[7] == "Max value: %d, scale: %f" % (mx, scale)
    assert lines[8] == "Saving in-progress buffer to %s." % buf
    assert lines[9] == "Saving image."
    assert lines[10] == "Done! Output image saved: %s" % out
    with open(out, "rb") as f:
        assert f.read() == oracle.encode_pgm(gray)
    state = read_state_file(buf, 200, 300)   # header + native-endian u64[h][w]
    assert np.array_equal(state, hist)


def test_resume_adds_to_the_saved_buffer(exe, small_render, tmp_path):
    """-s: load, render ON TOP, save (cudabrot.cu:783-785).  The generator restarts from seed 1337
    (SURVEY.md F5), so a second identical run doubles every count."""
    buf = str(tmp_path / "state.bin")
    args = ["--passes", "2", "-w", "300", "-h", "200", "-m", "200", "-o", os.devnull, "-s", buf]
    assert run(exe, *args).returncode == 0
    r = run(exe, *args)
    assert r.returncode == 0
    assert "doesn't exist yet" not in r.stdout
    state = read_state_file(buf, 200, 300)
    assert np.array_equal(state, 2 * small_render[0])


def test_rng_state_sidecar_continues_the_sample_stream(exe, oracle, tmp_path):
    """--rng-state (extension): buffer + sidecar resume with NEW samples, so 2 passes,
    save, 1 more pass equals one run of 3 passes; without the flag the behaviour stays the reference's."""
    buf, side = str(tmp_path / "state.bin"), str(tmp_path / "state.rng")
    common = ["-w", "300", "-h", "200", "-m", "200", "-o", os.devnull, "-s", buf, "--rng-state", side]
    r1 = run(exe, "--passes", "2", *common)
    assert r1.returncode == 0
    assert "File %s doesn't exist yet. Not loading." % side in r1.stdout
    assert "Saving generator state to %s." % side in r1.stdout
    r2 = run(exe, "--passes", "1", *common)
    assert r2.returncode == 0
    assert "Continuing the sample stream after 2 passes." in r2.stdout
    three, _ = oracle.render(300, 200, 200, 20, T, 3, omp_threads=0)
    state = read_state_file(buf, 200, 300)
    assert np.array_equal(state, three)
    # a sidecar for another seed is refused
    r3 = run(exe, "--passes", "1", "--seed", "99", *common)
    assert r3.returncode == 1 and "is not a generator state for seed 99" in r3.stdout


def test_burning_ship_flag(exe, oracle, tmp_path):
    """--burning-ship (extension): the reference's RENDER_BURNING_SHIP build (cudabrot.cu:15-17)."""
    buf = str(tmp_path / "ship.bin")
    r = run(exe, "--passes", "1", "--burning-ship", "-w", "300", "-h", "200", "-m", "200", "-o", os.devnull, "-s", buf)
    assert r.returncode == 0
    hist, _ = oracle.render(300, 200, 200, 20, T, 1, burning_ship=True, omp_threads=0)
    assert np.array_equal(read_state_file(buf, 200, 300), hist)


def test_channel_flags_render_the_colour_recipe_in_one_run(exe, oracle, tmp_path):
    """--channel MAX:MIN:FILE (extension): what generate_hires_color_image.sh does with
    three runs -- every image equals the separate run with that -m / -c and the same passes."""
    outs = [str(tmp_path / ("c%d.pgm" % j)) for j in range(3)]
    windows = [(100, 20), (400, 100), (1500, 400)]
    buf = str(tmp_path / "planes.bin")
    args = []
    for (m, c), o in zip(windows, outs):
        args += ["--channel", "%d:%d:%s" % (m, c, o)]
    r = run(exe, "--passes", "2", "-w", "300", "-h", "200", "-g", "2.2", "-s", buf, *args)
    assert r.returncode == 0, r.stdout
    assert "Creating 300x200 image, 1500 max iterations." in r.stdout
    planes = read_state_file(buf, 200, 300, planes=3)
    for j, ((m, c), o) in enumerate(zip(windows, outs)):
        hist, _ = oracle.render(300, 200, m, c, T, 2, omp_threads=0)
        assert np.array_equal(planes[j], hist)
        gray, mx, scale = oracle.set_grayscale_pixels(hist, 2.2)
        assert "Max value: %d, scale: %f" % (mx, scale) in r.stdout
        with open(o, "rb") as f:
            assert f.read() == oracle.encode_pgm(gray)
    bad = run(exe, "--channel", "100:20")          # no file name
    assert bad.returncode == 0 and "Invalid channel" in bad.stdout and "Usage:" in bad.stdout


def test_gpus_flag_shards_the_subsequences_and_sums_once(exe, oracle, tmp_path):
    """--gpus N (extension): rank r renders subsequences [r T, (r+1) T) on its own device and
    the histograms are summed onto rank 0 once.  CUDABROT_AMD_FAKE_GPUS=1 puts every rank on device 0 (this
    box has one GPU): 3 "GPUs" x 2 passes == one run of 3 T threads x 2 passes."""
    buf = str(tmp_path / "multi.bin")
    env = dict(os.environ, CUDABROT_AMD_FAKE_GPUS="1")
    r = run(exe, "--gpus", "3", "--passes", "2", "--stats", "-w", "300", "-h", "200", "-m", "200", "-o", os.devnull,
            "-s", buf, env=env)
    assert r.returncode == 0, r.stdout
    assert re.search(r"^6 Buddhabrot passes took", r.stdout, re.M)          # 3 ranks x 2 passes
    hist, cnt = oracle.render(300, 200, 200, 20, 3 * T, 2, omp_threads=0)
    assert np.array_equal(read_state_file(buf, 200, 300), hist)
    import json
    stats = json.loads(r.stderr.strip().splitlines()[-1])
    assert stats["samples"] == cnt["samples"] and stats["increments"] == cnt["increments"] and stats["status"] == 0


@pytest.mark.parametrize("seed", [4242, 123456789012345])   # the generator's seed is 64 bits wide
def test_seed_flag_selects_another_sample_stream(exe, oracle, tmp_path, seed):
    buf = str(tmp_path / "seed.bin")
    r = run(exe, "--passes", "1", "--seed", str(seed), "-w", "300", "-h", "200", "-m", "200", "-o", os.devnull, "-s", buf)
    assert r.returncode == 0
    hist, _ = oracle.render(300, 200, 200, 20, T, 1, seed=seed, omp_threads=0)
    assert np.array_equal(read_state_file(buf, 200, 300), hist)


def test_reference_format_u32_buffer_is_accepted_and_widened(exe, small_render, tmp_path):
    buf = str(tmp_path / "ref_state.bin")
    base = (np.arange(300 * 200, dtype=np.uint32) % 1000).reshape(200, 300)
    base.tofile(buf)                                   # what the reference writes: uint32[h][w]
    r = run(exe, "--passes", "2", "-w", "300", "-h", "200", "-m", "200", "-o", os.devnull, "-s", buf)
    assert r.returncode == 0
    assert "%s has no header and the size of 32-bit counters: read as the reference's format." % buf in r.stdout
    state = read_state_file(buf, 200, 300)   # rewritten in the native format: header + u64
    assert np.array_equal(state, base.astype(np.uint64) + small_render[0])


def test_state_format_raw_writes_the_buffer_the_reference_reads(exe, small_render, tmp_path):
    """--state-format raw: the -s file is the reference's own format again -- bare uint32[h][w], no header
    (cudabrot.cu:262-280) -- so a buffer can go BACK to the reference binary (or to np.fromfile); resuming from it
    adds to it like the reference does (cudabrot.cu:215-258)."""
    buf = str(tmp_path / "raw_state.bin")
    args = ("-w", "300", "-h", "200", "-m", "200", "-o", os.devnull, "-s", buf, "--state-format", "raw")
    r = run(exe, "--passes", "2", *args)
    assert r.returncode == 0
    assert os.path.getsize(buf) == 300 * 200 * 4                      # what LoadInProgressBuffer checks, :236-245
    first = np.fromfile(buf, dtype=np.uint32).reshape(200, 300)
    assert np.array_equal(first.astype(np.uint64), small_render[0])
    r = run(exe, "--passes", "2", *args)                                # resume: seed 1337 again, like the reference
    assert r.returncode == 0 and "read as the reference's format" in r.stdout
    second = np.fromfile(buf, dtype=np.uint32).reshape(200, 300)
    assert np.array_equal(second.astype(np.uint64), 2 * small_render[0])
    # a count beyond 32 bits cannot be written narrow: raw 64-bit counters, announced, read back under the flag
    big = small_render[0].copy()
    big[3, 4] = 1 << 40
    big.astype(np.uint64).tofile(buf)
    r = run(exe, "--passes", "0", *args)
    assert r.returncode == 0 and "read as raw 64-bit counters" in r.stdout and "not a file the reference can load" in r.stdout
    assert np.array_equal(np.fromfile(buf, dtype=np.uint64).reshape(200, 300), big)


def test_native_buffer_of_another_canvas_is_refused_even_at_the_reference_size(exe, tmp_path):
    """A native (u64) buffer of HALF the pixel count has exactly w*h*4 bytes of payload for this canvas: the
    format is decided by the header, never by the size, so it is refused with the reference's size-mismatch
    line (cudabrot.cu:239-245) instead of being read as 32-bit counters."""
    buf = str(tmp_path / "half.bin")
    assert run(exe, "--passes", "1", "-w", "300", "-h", "100", "-m", "100", "-o", os.devnull, "-s", buf).returncode == 0
    size = os.path.getsize(buf)
    assert size == STATE_HEADER_BYTES + 300 * 100 * 8
    r = run(exe, "--passes", "1", "-w", "300", "-h", "200", "-m", "100", "-o", os.devnull, "-s", buf)
    assert r.returncode == 1
    assert "%s holds a 300x100 buffer of 1 plane(s), 8-byte counters" % buf in r.stdout
    assert "The size of %s doesn't match the expected size of %d bytes." % (buf, STATE_HEADER_BYTES + 300 * 200 * 8) in r.stdout
    assert os.path.getsize(buf) == size                                # left untouched
    # the same payload WITHOUT the header is, by its size, a reference file of the 300 x 200 canvas: accepted
    with open(buf, "rb") as f:
        payload = f.read()[STATE_HEADER_BYTES:]
    with open(buf, "wb") as f:
        f.write(payload)
    r = run(exe, "--passes", "0", "-w", "300", "-h", "200", "-m", "100", "-o", os.devnull, "-s", buf)
    assert r.returncode == 0 and "read as the reference's format" in r.stdout
    assert np.array_equal(read_state_file(buf, 200, 300), np.frombuffer(payload, dtype=np.uint32).reshape(200, 300))


def test_rng_state_sidecar_with_several_gpus(exe, oracle, tmp_path):
    """True resume with --gpus N (SURVEY.md 8f N3 + 8e): the sidecar keeps every rank's generator, so 2 ranks x
    (2 + 1) passes in two runs == one run of 2 T threads x 3 passes; a sidecar of another rank count is refused.
    Every rank on device 0 here (CUDABROT_AMD_FAKE_GPUS); tests/test_gpu_multi.py runs it on real devices."""
    env = dict(os.environ, CUDABROT_AMD_FAKE_GPUS="1")
    buf, side = str(tmp_path / "m.bin"), str(tmp_path / "m.rng")
    common = ["--gpus", "2", "-w", "300", "-h", "200", "-m", "200", "-o", os.devnull, "-s", buf, "--rng-state", side]
    r1 = run(exe, "--passes", "2", *common, env=env)
    assert r1.returncode == 0, r1.stdout
    assert re.search(r"^4 Buddhabrot passes took", r1.stdout, re.M)
    r2 = run(exe, "--passes", "1", *common, env=env)
    assert r2.returncode == 0, r2.stdout
    assert "Continuing the sample stream after 2 passes." in r2.stdout
    three, _ = oracle.render(300, 200, 200, 20, 2 * T, 3, omp_threads=0)
    assert np.array_equal(read_state_file(buf, 200, 300), three)
    r3 = run(exe, "--passes", "1", "--gpus", "3", *common[2:], env=env)
    assert r3.returncode == 1 and "is not a generator state for seed 1337, 262144 threads and 3 GPU(s)." in r3.stdout


def test_eight_ranks_with_fused_channels_and_true_resume(exe, oracle, tmp_path):
    """The shape of the first 8-GPU run, rehearsed with every rank on device 0 (CUDABROT_AMD_FAKE_GPUS): --gpus 8
    with three fused --channel windows and the --rng-state sidecar, 1 + 1 passes in two runs == one render of
    8 T threads x 2 passes per window (persistent rank threads, the shared pass budget, eight generators in the
    sidecar, one reduce of three planes)."""
    env = dict(os.environ, CUDABROT_AMD_FAKE_GPUS="1")
    windows = [(60, 20), (200, 60), (600, 200)]
    outs = [str(tmp_path / ("g%d.pgm" % j)) for j in range(3)]
    buf, side = str(tmp_path / "g8.bin"), str(tmp_path / "g8.rng")
    args = ["--gpus", "8", "-w", "160", "-h", "120", "-s", buf, "--rng-state", side]
    for (m, c), o in zip(windows, outs):
        args += ["--channel", "%d:%d:%s" % (m, c, o)]
    r1 = run(exe, "--passes", "1", *args, env=env)
    assert r1.returncode == 0, r1.stdout
    assert re.search(r"^8 Buddhabrot passes took", r1.stdout, re.M)          # 8 ranks x 1 pass
    r2 = run(exe, "--passes", "1", *args, env=env)
    assert r2.returncode == 0, r2.stdout
    assert "Continuing the sample stream after 1 passes." in r2.stdout
    planes = read_state_file(buf, 120, 160, planes=3)
    for j, (m, c) in enumerate(windows):
        hist, _ = oracle.render(160, 120, m, c, 8 * T, 2, omp_threads=0)
        assert np.array_equal(planes[j], hist), j


def test_timed_run_with_several_ranks_renders_the_same_passes_on_each(exe, oracle, tmp_path):
    """-t with --gpus N: the ranks share a pass budget that the main thread alone raises, so however the clock ends
    the run every rank has rendered the same P passes and the result is `N T threads for P passes` exactly."""
    env = dict(os.environ, CUDABROT_AMD_FAKE_GPUS="1")
    buf = str(tmp_path / "t.bin")
    r = run(exe, "--gpus", "2", "-t", "0.5", "-w", "200", "-h", "150", "-m", "100", "-o", os.devnull, "-s", buf, env=env)
    assert r.returncode == 0, r.stdout
    m = re.search(r"^(\d+) Buddhabrot passes took", r.stdout, re.M)
    total = int(m.group(1))
    assert total >= 2 and total % 2 == 0
    got = read_state_file(buf, 150, 200)
    # the oracle at this size is too slow for thousands of passes: the sum of the histogram pins the pass count
    # through the exact per-pass increment count of 2 T threads, which is not constant -- so compare short runs
    if total // 2 <= 4:
        hist, _ = oracle.render(200, 150, 100, 20, 2 * T, total // 2, omp_threads=0)
        assert np.array_equal(got, hist)
    else:
        again = str(tmp_path / "t2.bin")
        r2 = run(exe, "--gpus", "2", "--passes", str(total // 2), "-w", "200", "-h", "150", "-m", "100", "-o", os.devnull,
                 "-s", again, env=env)
        assert r2.returncode == 0
        assert np.array_equal(got, read_state_file(again, 150, 200))


def test_buffer_size_mismatch_is_an_error(exe, tmp_path):
    buf = str(tmp_path / "bad.bin")
    with open(buf, "wb") as f:
        f.write(b"\0" * 1000)
    r = run(exe, "--passes", "1", "-w", "300", "-h", "200", "-o", os.devnull, "-s", buf)
    assert r.returncode == 1                                           # cudabrot.cu:239-245
    assert "The size of %s doesn't match the expected size of %d bytes." % (buf, STATE_HEADER_BYTES + 300 * 200 * 8) in r.stdout
    assert os.path.getsize(buf) == 1000                                # left untouched


@pytest.mark.parametrize("gamma", ["1.0", "0", "-1", "0.5"])
def test_gamma_variants(exe, oracle, small_render, tmp_path, gamma):
    out = str(tmp_path / "g.pgm")
    r = run(exe, "--passes", "2", "-w", "300", "-h", "200", "-m", "200", "-g", gamma, "-o", out)
    assert r.returncode == 0
    gray, _, _ = oracle.set_grayscale_pixels(small_render[0], float(gamma))
    with open(out, "rb") as f:
        assert f.read() == oracle.encode_pgm(gray)


@pytest.mark.parametrize("tonemap", ["host", "lut", "thresholds"])
def test_tonemap_paths_write_the_same_image(exe, oracle, small_render, tmp_path, tonemap):
    """--tonemap (extension): the device table forms and the reference's host loop give the same bytes."""
    out = str(tmp_path / "t.pgm")
    r = run(exe, "--passes", "2", "-w", "300", "-h", "200", "-m", "200", "-g", "2.2", "--tonemap", tonemap, "-o", out)
    assert r.returncode == 0
    gray, mx, scale = oracle.set_grayscale_pixels(small_render[0], 2.2)
    assert "Max value: %d, scale: %f" % (mx, scale) in r.stdout
    with open(out, "rb") as f:
        assert f.read() == oracle.encode_pgm(gray)


def test_timed_run_renders_at_least_one_pass_and_stops(exe, tmp_path):
    out = str(tmp_path / "t.pgm")
    t0 = time.time()
    r = run(exe, "-t", "0.5", "-w", "256", "-h", "256", "-o", out)
    assert r.returncode == 0
    assert "Running for 0.500 seconds." in r.stdout
    m = re.search(r"(\d+) Buddhabrot passes took ([0-9.]+) seconds", r.stdout)
    assert m and int(m.group(1)) >= 1 and 0.5 <= float(m.group(2)) < 5.0
    assert time.time() - t0 < 60
    with open(out, "rb") as f:
        assert f.read(15) == b"P5\n256 256\n6553"


def test_sigint_finishes_the_pass_and_still_saves(exe, tmp_path):
    """-t < 0 runs until SIGINT; the handler lets the current launch finish, the image and the -s buffer
    are written and the exit code is 0 (cudabrot.cu:475-476, 756-760)."""
    out, buf = str(tmp_path / "i.pgm"), str(tmp_path / "i.bin")
    p = subprocess.Popen([exe, "-t", "-1", "-w", "256", "-h", "256", "-o", out, "-s", buf],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(3.0)
    p.send_signal(signal.SIGINT)
    stdout, _ = p.communicate(timeout=120)
    assert p.returncode == 0
    assert "Press ctrl+C to finish." in stdout
    assert "Signal 2 received, waiting for current pass to finish..." in stdout
    assert "Done! Output image saved: %s" % out in stdout
    assert os.path.getsize(out) == len(b"P5\n256 256\n65535\n") + 256 * 256 * 2
    assert os.path.getsize(buf) == STATE_HEADER_BYTES + 256 * 256 * 8


def test_invalid_device_reports_like_the_reference_and_exits_one(exe):
    r = run(exe, "-d", "99", "--passes", "1", "-w", "16", "-h", "16", "-o", os.devnull)
    assert r.returncode == 1
    assert r.stdout.strip().split("\n")[-1].startswith("CUDA error ")   # cudabrot.cu:137, wording kept


def test_eight_ranks_all_use_the_interior_map(exe, tmp_path):
    """keep the first 8-GPU run boring -- every rank's launches must use the (embedded) interior map; the
    record is the renderer's own (cb_renderer_interior_map_level), not the process-wide figure that rank threads
    overwrite.  --stats prints it rank by rank."""
    import json

    env = dict(os.environ, CUDABROT_AMD_FAKE_GPUS="1")
    r = run(exe, "--gpus", "8", "--passes", "2", "-w", "256", "-h", "256", "-m", "2000", "-o", os.devnull, "--stats", env=env)
    assert r.returncode == 0, r.stdout
    stats = json.loads([l for l in r.stderr.splitlines() if l.startswith("{")][-1])
    assert stats["interior_map_levels"] == [12] * 8, stats
    assert stats["status"] == 0 and stats["skipped_steps"] > 0 and stats["samples"] == 8 * T * 50 * 2
