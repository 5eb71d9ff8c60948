"""tools/interior_map.c on the CPU: the generator / prover of the interior map, its model and its check (no GPU).

The map's claim is proved cell by cell by the generator itself (see the file's header); these tests hold the program to
what it says: that a map it makes survives its own check (samples inside marked cells, iterated to max_iter with the
reference's arithmetic: none escapes), that the model finds no disagreement between the map and the iteration, that the
marked area is what it was when the kernel's share of work was measured, and that the file `make` puts beside the
library is the one the library accepts."""

import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "interior_map.c")


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("interior") / "interior_map")
    subprocess.run(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-o", exe, SRC, "-lm"], check=True)
    return exe


def run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def test_a_small_map_passes_its_own_check_and_model(tool, tmp_path):
    path = str(tmp_path / "map10.bin")
    out = run(tool, "make", "10", path)
    m = re.search(r"(\d+) tried, (\d+) marked", out)
    assert m and int(m.group(2)) == 23769, out            # level 10: what the prover marks (a change of the prover shows here)
    head = struct.unpack("<4I", open(path, "rb").read(16))
    assert head == (0x4D494243, 10, 2560, 1280)
    assert os.path.getsize(path) == 16 + 2560 * 1280 // 8
    assert open(path, "rb").read(17)[16] & 1 == 0         # cell 0 is the kernel's "outside": never marked
    out = run(tool, "check", "10", path, "200000")
    assert re.search(r"200000 samples inside 23769 marked cells iterated to 20000: 0 escaped", out), out
    out = run(tool, "model", "10", path, "300000")
    assert "DISAGREEMENT" not in out
    m = re.search(r"periodicity check alone ([\d.]+), with the map ([\d.]+)", out)
    assert m and float(m.group(2)) < float(m.group(1))


def test_the_built_map_is_what_the_library_accepts(tool):
    """cudabrot_amd/interior_map.bin (made by `make`, level 13): header, size, and a check of 400000 samples inside
    its marked cells."""
    path = os.path.join(ROOT, "cudabrot_amd", "interior_map.bin")
    if not os.path.exists(path):
        pytest.skip("interior_map.bin has not been built (make -C cudabrot_amd/csrc)")
    magic, level, cols, rows = struct.unpack("<4I", open(path, "rb").read(16))
    assert magic == 0x4D494243 and 8 <= level <= 15
    assert cols == (5 << level) // 2 and rows == (5 << level) // 4
    assert os.path.getsize(path) == 16 + (cols * rows + 7) // 8
    out = run(tool, "check", str(level), path, "400000")
    assert re.search(r"400000 samples inside \d+ marked cells iterated to 20000: 0 escaped", out), out
