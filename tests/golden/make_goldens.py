#!/usr/bin/env python3
"""Regenerates the golden vectors from the reference's OWN lines (oracle/_ref/, built by
`make -C oracle ref` where /root/reference is present): the histogram rows of appendix_b.json are
re-derived and compared with the committed file, and burning_ship.json (the reference compiled with
-DRENDER_BURNING_SHIP, cudabrot.cu:15-17) is written.

usage: python tests/golden/make_goldens.py            # verify appendix_b.json, (re)write burning_ship.json
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


if __name__ == "__main__":
    main()
