"""tools/interior_map.c on the CPU: the generator / prover of the interior map, its model and its check (no GPU).

The map's claim is proved cell by cell by the generator itself (see the file's header); these tests hold the program to
what it says -- a map it makes survives its own check (samples inside marked cells, iterated to max_iter with the
reference's arithmetic: none escapes), the model finds no disagreement between the map and the iteration -- and hold
the map KEPT in the tree (cudabrot_amd/interior_map.bin.gz: an hour of proving, `make interior-map`) to the program:
every cell a shallower, quick run of the prover marks is marked in it, nothing outside the prover's reach is, and
samples inside its cells do not escape."""

import gzip
import os
import re
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "interior_map.c")
KEPT = os.path.join(ROOT, "cudabrot_amd", "interior_map.bin.gz")


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("interior") / "interior_map")
    # (-march=native: the prover lives on fma(), which without the instruction is a library call, five times slower)
    subprocess.run(["gcc", "-O3", "-march=native", "-fopenmp", "-ffp-contract=off", "-o", exe, SRC, "-lm"], check=True)
    return exe


def run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def bits_of(path_or_bytes):
    raw = path_or_bytes if isinstance(path_or_bytes, bytes) else open(path_or_bytes, "rb").read()
    head = struct.unpack("<4I", raw[:16])
    return head, np.frombuffer(raw, dtype=np.uint8, offset=16)


def test_a_small_map_passes_its_own_check_and_model(tool, tmp_path):
    flat, deep = str(tmp_path / "map10_0.bin"), str(tmp_path / "map10_2.bin")
    out = run(tool, "make", "10", flat, "0")
    m = re.search(r"(\d+) tried, (\d+) marked", out)
    assert m and int(m.group(2)) == 23769, out            # level 10, whole cells only (a change of the prover shows here)
    out = run(tool, "make", "10", deep, "2")
    n_deep = int(re.search(r"(\d+) tried, (\d+) marked", out).group(2))
    assert n_deep > 28000, out                            # ... and with cells proven by their quarters, two levels down
    head, b0 = bits_of(flat)
    _, b2 = bits_of(deep)
    assert head == (0x4D494243, 10, 2560, 1280) and b0.size == 2560 * 1280 // 8
    assert b0[0] & 1 == 0 and b2[0] & 1 == 0              # cell 0 is the kernel's "outside": never marked
    assert np.all((b0 & ~b2) == 0)                        # what a whole-cell proof reaches, the deeper one reaches too
    out = run(tool, "check", "10", deep, "200000")
    assert re.search(r"200000 samples inside %d marked cells iterated to 20000: 0 escaped" % n_deep, out), out
    out = run(tool, "model", "10", deep, "300000")
    assert "DISAGREEMENT" not in out
    m = re.search(r"periodicity check alone ([\d.]+), with the map ([\d.]+)", out)
    assert m and float(m.group(2)) < float(m.group(1))


def test_the_kept_map_against_the_prover(tool, tmp_path):
    """cudabrot_amd/interior_map.bin.gz: level 12, cells proven up to nine levels of quarters down (ten: 2185 rim cells).  A run of the prover
    one level deep (seconds) must mark nothing the kept map lacks; the kept map must mark nothing outside the cells a
    CENTRE-only iteration finds bounded (the prover's necessary condition); 400000 samples inside its cells are
    iterated to max_iter."""
    kept = str(tmp_path / "kept.bin")
    raw = gzip.open(KEPT, "rb").read()
    open(kept, "wb").write(raw)
    head, kept_bits = bits_of(raw)
    magic, level, cols, rows = head
    assert magic == 0x4D494243 and level == 12 and cols == 10240 and rows == 5120
    assert kept_bits.size == cols * rows // 8 and kept_bits[0] & 1 == 0
    shallow = str(tmp_path / "shallow.bin")
    run(tool, "make", "12", shallow, "1")
    _, shallow_bits = bits_of(shallow)
    assert np.all((shallow_bits & ~kept_bits) == 0), "a cell the prover marks is missing from the kept map"
    n_kept = int(np.unpackbits(kept_bits).sum())
    assert n_kept > int(np.unpackbits(shallow_bits).sum()) > 700000
    # a necessary condition, checked for EVERY marked cell: its centre's orbit stays below 1.99 in modulus (numpy's
    # complex arithmetic rounds differently from the reference's sequence; the proof's bound has the room)
    marked = np.flatnonzero(np.unpackbits(kept_bits, bitorder="little"))
    assert marked.size == n_kept
    s = 2.0 ** -level
    c = (-2.0 + (marked % cols + 0.5) * s) + 1j * ((marked // cols + 0.5) * s)
    z = c.copy()
    worst = 0.0
    for _ in range(400):
        z = z * z + c
        worst = max(worst, float(np.abs(z).max()))
    assert worst < 1.99, worst
    out = run(tool, "check", "12", kept, "400000")
    assert re.search(r"400000 samples inside %d marked cells iterated to 20000: 0 escaped" % n_kept, out), out
    built = os.path.join(ROOT, "cudabrot_amd", "csrc", "build", "interior_map.bin")
    if os.path.exists(built):                             # what `make` unpacked (and embeds in the library)
        assert open(built, "rb").read() == raw


def test_the_kept_map_is_the_pinned_one_and_the_embedded_one():
    """The map that decides results is pinned three ways: the digest kept in the tree (csrc/map_digests.sha256, which
    `make` checks before it assembles the map into the library), the kept file, and the bytes the built library carries."""
    import ctypes
    import hashlib

    raw = gzip.open(KEPT, "rb").read()
    pinned = open(os.path.join(ROOT, "cudabrot_amd", "csrc", "map_digests.sha256")).read().split()
    assert pinned[1].endswith("interior_map.bin") and hashlib.sha256(raw).hexdigest() == pinned[0]
    lib = os.path.join(ROOT, "cudabrot_amd", "libcudabrot_amd.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    so = ctypes.CDLL(lib)
    begin = ctypes.addressof(ctypes.c_ubyte.in_dll(so, "cb_embedded_interior_map"))
    end = ctypes.addressof(ctypes.c_ubyte.in_dll(so, "cb_embedded_interior_map_end"))
    assert end - begin == len(raw)
    assert ctypes.string_at(begin, end - begin) == raw


DEPTHS = os.path.join(ROOT, "cudabrot_amd", "interior_map.depths.gz")


def test_a_stratum_of_every_depth_is_proven_again(tool, tmp_path):
    """VERDICT r03 #2b/c: `make verify-interior-map` has proven EVERY marked cell of the kept map again and noted the
    depth each proof needed (cudabrot_amd/interior_map.depths.gz: one byte per marked cell, none unknown, none beyond
    the kept budget of ten levels of quarters -- nine for the map as round 3 made it, ten for the 2185 cells round 4's
    anytime run over its rim added).  Here, on every CPU run: the notes belong to this map, and a random
    stratum of EVERY depth -- 40 cells each of depth 1..10, the deep ones being where a bit set by accident would have to
    hide -- plus 2000 random cells are proven again by the prover AT THE DEPTH NOTED (a cell that is not interior cannot
    be proven at any depth; one that needs another depth than noted says the notes are not this tool's)."""
    raw = gzip.open(KEPT, "rb").read()
    kept = str(tmp_path / "kept.bin")
    open(kept, "wb").write(raw)
    _, kept_bits = bits_of(raw)
    n_marked = int(np.unpackbits(kept_bits).sum())
    notes = gzip.open(DEPTHS, "rb").read()
    magic, level, n, budget = struct.unpack("<4I", notes[:16])
    depths = np.frombuffer(notes, dtype=np.uint8, offset=16)
    assert magic == 0x44494243 and level == 12 and n == n_marked == depths.size and budget == 10
    assert int(depths.max()) <= 10, "a marked cell without a proof within the kept budget"
    side = str(tmp_path / "depths.bin")
    open(side, "wb").write(notes)
    # 40 cells of EVERY depth 1..10 (a stratum each, drawn at random) and 2000 random others (two thirds of them depth 0)
    out = run(tool, "verify", "12", kept, "10", side, "2000", "0", "2026", "40")
    m = re.search(r"proved (\d+) cells, (\d+) NOT proven, (\d+) with another depth than noted", out)
    assert m and int(m.group(2)) == 0 and int(m.group(3)) == 0, out
    assert int(m.group(1)) >= 10 * 40 + 1500, out
