"""Randomized end-to-end check of the `cudabrot` binary against the oracle: random canvas, window, iteration
limits, gamma, pass count, tone-map method, fractal, seed and -s resume; the PGM must be byte-identical and the
-s buffer equal to the oracle's histogram.  CLI_FUZZ_SECONDS (default 25) bounds the loop."""

import os
import random
import subprocess
import time

import numpy as np
import pytest

from conftest import read_state_file

pytestmark = pytest.mark.gpu

T = 512 * 512


def _trial(rng):
    t = {}
    t["w"] = rng.choice([1, 17, 100, 128, 255, 300, 512, 640])
    t["h"] = rng.choice([1, 8, 100, 127, 200, 256, 333, 480])
    kind = rng.random()
    if kind < 0.5:
        t["box"] = None
    elif kind < 0.75:
        t["box"] = (-2.0, 1.0, -1.5, 1.5)
    else:
        cx, cy = rng.uniform(-1.5, 0.5), rng.uniform(-1.0, 1.0)
        rx, ry = rng.uniform(0.05, 2.0), rng.uniform(0.05, 2.0)
        t["box"] = (round(cx - rx, 6), round(cx + rx, 6), round(cy - ry, 6), round(cy + ry, 6))
    t["m"] = rng.choice([1, 5, 20, 21, 50, 100, 250, 400])
    t["c"] = rng.choice([None, 0, 1, 19, 20, 33, 500])
    t["g"] = rng.choice([None, 1.0, 2.2, 0.5, 0.0, -1.0, round(rng.uniform(0.1, 5.0), 3)])
    t["passes"] = rng.choice([1, 1, 2, 3])
    t["tonemap"] = rng.choice([None, "host", "lut", "thresholds"])
    t["ship"] = rng.random() < 0.2
    t["seed"] = rng.choice([None, None, 7, 123456789012345])
    t["resume"] = rng.random() < 0.25
    return t


def _args(t, out, buf):
    a = ["--passes", str(t["passes"]), "-w", str(t["w"]), "-h", str(t["h"]), "-m", str(t["m"]), "-o", out]
    if t["box"]:   # max before min: the canvas is validated after every flag (cudabrot.cu:704-749)
        a += ["--max-real", repr(t["box"][1]), "--min-real", repr(t["box"][0]),
              "--max-imag", repr(t["box"][3]), "--min-imag", repr(t["box"][2])]
    if t["c"] is not None:
        a += ["-c", str(t["c"])]
    if t["g"] is not None:
        a += ["-g", repr(t["g"])]
    if t["tonemap"]:
        a += ["--tonemap", t["tonemap"]]
    if t["ship"]:
        a += ["--burning-ship"]
    if t["seed"] is not None:
        a += ["--seed", str(t["seed"])]
    if buf:
        a += ["-s", buf]
    return a


def test_random_command_lines_against_the_oracle(repo_root, oracle, tmp_path):
    exe = os.path.join(repo_root, "cudabrot")
    rng = random.Random(int(os.environ.get("CLI_FUZZ_SEED", "3")))
    t_end = time.time() + float(os.environ.get("CLI_FUZZ_SECONDS", "25"))
    n = 0
    while time.time() < t_end or n < 3:
        t = _trial(rng)
        out, buf = str(tmp_path / "o.pgm"), str(tmp_path / "s.bin")
        for f in (out, buf):
            if os.path.exists(f):
                os.remove(f)
        if not t["resume"]:
            pass  # the -s buffer is written either way: it is how the histogram is compared
        runs = 2 if t["resume"] else 1
        for _ in range(runs):
            r = subprocess.run([exe] + _args(t, out, buf), stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, text=True, timeout=300)
            assert r.returncode == 0 and "Done!" in r.stdout, (t, r.stdout[-500:], r.stderr[-500:])
        box = t["box"] or (-2.0, 2.0, -2.0, 2.0)
        hist, _ = oracle.render(t["w"], t["h"], t["m"], 20 if t["c"] is None else t["c"], T, t["passes"], box,
                                seed=1337 if t["seed"] is None else t["seed"], omp_threads=0,
                                burning_ship=t["ship"])
        hist = hist * np.uint64(runs)     # a resumed run replays the same stream on top
        gray, mx, _ = oracle.set_grayscale_pixels(hist, 1.0 if t["g"] is None else t["g"])
        state = read_state_file(buf, t["h"], t["w"])
        assert np.array_equal(state, hist), "histogram differs: %r" % (t,)
        with open(out, "rb") as f:
            assert f.read() == oracle.encode_pgm(gray), "image differs (same histogram, max %d): %r" % (mx, t)
        n += 1
    print("cli fuzz: %d command lines identical" % n)
