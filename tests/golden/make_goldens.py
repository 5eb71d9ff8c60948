#!/usr/bin/env python3
"""Regenerates the golden vectors from the reference's OWN lines (oracle/_ref/, built by
`make -C oracle ref` where /root/reference is present): the histogram rows of appendix_b.json are
re-derived and compared with the committed file, and burning_ship.json (the reference compiled with
-DRENDER_BURNING_SHIP, cudabrot.cu:15-17) is written.

usage: python tests/golden/make_goldens.py            # verify appendix_b.json, (re)write burning_ship.json
       python tests/golden/make_goldens.py --full     # also (re)write full_size.json: BASELINE.json's C4 and C5
                                                      # at full size (minutes; ~20 GB of host memory)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import binding as oracle  # noqa: E402  (test infrastructure)


def ref_render(ref, g):
    hist = np.zeros((g["h"], g["w"]), dtype=np.uint32)
    box = g["box"]
    rc = ref.ref_draw(g["w"], g["h"], box[0], box[1], box[2], box[3], g["max_iter"], g["min_iter"], 0,
                      g["threads"], g["passes"], 50, hist.ctypes.data)
    assert rc == 0
    return hist.astype(np.uint64)


def row(g, hist):
    out = dict(g)
    out.update(samples=g["threads"] * 50 * g["passes"], increments=int(hist.sum()), max=int(hist.max()),
               nonzero=int((hist > 0).sum()), fnv1a64="%016x" % oracle.fnv1a_pixels(hist))
    return out


# ---- full-size configs (BASELINE.json C4, C5) ------------------------------------------------------
#
# The reference's lines run one "thread" after another, and threads share nothing but the additive
# histogram, so a run over subsequences [0, T) is the sum of runs over disjoint sub-ranges: each worker
# process runs the reference's lines (libref_fma.so) over its own sub-range into its own u32 histogram.
FULL_SIZE = [
    # C4: README.md:69-76 (20000 x 20000), BASELINE.json configs[3]
    dict(name="c4_20000_m20000", w=20000, h=20000, max_iter=20000, min_iter=20, threads=262144, passes=3,
         box=[-2.0, 2.0, -2.0, 2.0]),
    # C5: generate_hires_color_image.sh:27-59 -- 20000 x 15000 on [-2,2] x [-1.5,1.5], -m/-c per channel
    dict(name="c5_recipe_m60000_c45000", w=20000, h=15000, max_iter=60000, min_iter=45000, threads=262144, passes=2,
         box=[-2.0, 2.0, -1.5, 1.5]),
    dict(name="c5_recipe_m8000_c1000", w=20000, h=15000, max_iter=8000, min_iter=1000, threads=262144, passes=2,
         box=[-2.0, 2.0, -1.5, 1.5]),
    dict(name="c5_recipe_m500_c20", w=20000, h=15000, max_iter=500, min_iter=20, threads=262144, passes=2,
         box=[-2.0, 2.0, -1.5, 1.5]),
    # C5 as BASELINE.json words it: max_iter 200 / 2000 / 20000 (min_iter: the default 20, cudabrot.cu:766)
    dict(name="c5_baseline_m200", w=20000, h=15000, max_iter=200, min_iter=20, threads=262144, passes=2,
         box=[-2.0, 2.0, -1.5, 1.5]),
    dict(name="c5_baseline_m2000", w=20000, h=15000, max_iter=2000, min_iter=20, threads=262144, passes=2,
         box=[-2.0, 2.0, -1.5, 1.5]),
    dict(name="c5_baseline_m20000", w=20000, h=15000, max_iter=20000, min_iter=20, threads=262144, passes=2,
         box=[-2.0, 2.0, -1.5, 1.5]),
    # the deepest window of the recipe on a small canvas: a row of test_iteration_window_edges' kind
    dict(name="small_m60000_c45000", w=300, h=300, max_iter=60000, min_iter=45000, threads=65536, passes=2,
         box=[-2.0, 2.0, -2.0, 2.0]),
]


def _ref_part(args):
    g, first, n, path = args
    ref = oracle.ref_library("fma")
    hist = np.zeros((g["h"], g["w"]), dtype=np.uint32)
    box = g["box"]
    rc = ref.ref_draw(g["w"], g["h"], box[0], box[1], box[2], box[3], g["max_iter"], g["min_iter"], first, n,
                      g["passes"], 50, hist.ctypes.data)
    assert rc == 0
    np.save(path, hist)
    return path


def ref_render_parallel(g, workers=6):
    import multiprocessing as mp
    import tempfile

    tmp = tempfile.mkdtemp(prefix="cb_gold_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    per = (g["threads"] + workers - 1) // workers
    jobs = [(g, k * per, min(per, g["threads"] - k * per), os.path.join(tmp, "part%d.npy" % k))
            for k in range(workers) if k * per < g["threads"]]
    total = np.zeros((g["h"], g["w"]), dtype=np.uint64)
    try:
        with mp.get_context("spawn").Pool(len(jobs)) as pool:
            for path in pool.imap_unordered(_ref_part, jobs):
                total += np.load(path)
                os.remove(path)
    finally:
        for _, _, _, path in jobs:
            if os.path.exists(path):
                os.remove(path)
        os.rmdir(tmp)
    return total


def write_full_size():
    import time

    rows = []
    for g in FULL_SIZE:
        t0 = time.time()
        hist = ref_render_parallel(g)
        rows.append(row(g, hist))
        print("%s: %d increments, max %d, %.0f s" % (g["name"], rows[-1]["increments"], rows[-1]["max"], time.time() - t0),
              flush=True)
        del hist
    out = {"_comment": "BASELINE.json's C4 and C5 at full size: cudabrot.cu:43-67,284-414 compiled for the host "
                       "(oracle/Makefile ref, libref_fma.so = the contraction hipcc applies on gfx950), one process per "
                       "sub-range of the 262144 subsequences, u32 histograms summed; rocRAND XORWOW seed 1337; written by "
                       "tests/golden/make_goldens.py --full",
           "histograms": rows}
    with open(os.path.join(HERE, "full_size.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("full_size.json: %d rows written" % len(rows))


def main():
    fma, ship = oracle.ref_library("fma"), oracle.ref_library("ship_fma")
    if fma is None or ship is None:
        raise SystemExit("oracle/_ref is not built (needs /root/reference): make -C oracle ref")
    committed = json.load(open(os.path.join(HERE, "appendix_b.json")))
    for g in committed["histograms"]:
        keys = ("name", "w", "h", "max_iter", "min_iter", "threads", "passes", "box")
        again = row({k: g[k] for k in keys}, ref_render(fma, g))
        assert again == g, "appendix_b.json row %s does not match the reference's lines: %r" % (g["name"], again)
    print("appendix_b.json: %d histogram rows re-derived from the reference's lines, identical" % len(committed["histograms"]))
    ship_rows = []
    for g in (
        dict(name="ship_256", w=256, h=256, max_iter=100, min_iter=20, threads=20000, passes=1, box=[-2.0, 2.0, -2.0, 2.0]),
        dict(name="ship_crop", w=400, h=300, max_iter=2000, min_iter=20, threads=8000, passes=2, box=[-2.0, 2.0, -2.0, 1.0]),
        dict(name="ship_c0", w=333, h=77, max_iter=500, min_iter=0, threads=3000, passes=2, box=[-1.9, 0.9, -1.3, 0.4]),
    ):
        ship_rows.append(row(g, ref_render(ship, g)))
    out = {"_comment": "cudabrot.cu:43-67,284-414 compiled for the host with -DRENDER_BURNING_SHIP=1 "
                       "(oracle/Makefile ref, libref_ship_fma.so), sequential, rocRAND XORWOW seed 1337; "
                       "written by tests/golden/make_goldens.py",
           "histograms": ship_rows}
    with open(os.path.join(HERE, "burning_ship.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("burning_ship.json: %d rows written" % len(ship_rows))
    if "--full" in sys.argv[1:]:
        write_full_size()


if __name__ == "__main__":
    main()
