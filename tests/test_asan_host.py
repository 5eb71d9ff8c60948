"""Host code of the product under AddressSanitizer + UndefinedBehaviorSanitizer -- CPU only.

tests/asan/Makefile builds (ROCm clang, -fsanitize=address,undefined, never the GPU code):
  host_selftest   state_files.cpp / host_output.cpp / xorwow_host.cpp / host_abi.cpp driven through the -s and
                  sidecar file paths that mirror cudabrot.cu:192-280 (size mismatch, short read, wrong canvas,
                  reference-format widening), the PGM writer and tone map (vs the oracle's), the XORWOW tables
  cudabrot_asan   the binary with its own host code sanitized: the argument parser (cudabrot.cu:625-754) and
                  the device-error exit run under the sanitizers
A sanitizer finding aborts the process with a report on stderr and a nonzero status, which fails the test.
"""

import os
import subprocess

import pytest

ASAN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "asan")
# the HIP runtime the unsanitized library pulls in is not ours to check for leaks
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


@pytest.fixture(scope="module")
def built():
    r = subprocess.run(["make", "-C", ASAN_DIR, "all"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    return os.path.join(ASAN_DIR, "build")


def test_host_selftest_under_sanitizers(built, tmp_path):
    r = subprocess.run([os.path.join(built, "host_selftest"), str(tmp_path)], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300, env=ENV)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
    for name in ("state_files", "rng_sidecar", "output_stage", "xorwow_tables", "canvas_validation"):
        assert "ok %s\n" % name in r.stdout
    # the messages the file paths print are the binary's (and, where it has them, the reference's)
    assert "doesn't match the expected size of" in r.stdout          # cudabrot.cu:240-242
    assert "doesn't exist yet. Not loading." in r.stdout             # cudabrot.cu:228
    assert "read as the reference's format" in r.stdout


@pytest.mark.parametrize("args,first_line", [
    (["--help"], None),
    (["--bogus"], "Invalid argument: --bogus"),
    (["-m"], "Argument -m needs a value."),
    (["-m", "12x"], "Invalid number given to argument -m: 12x"),
    (["-w", ""], "Invalid number given to argument -w: "),
    (["-w", "99999999999999999999"], None),           # strtol saturates, the truncation to int is the reference's
    (["-g", "1e999"], None),
    (["-h", "-4"], "Output height must be positive."),
    (["--min-real", "3", "--max-real", "4"], "Maximum real value must be greater than minimum real value."),
    (["--channel", "a:b:c"], "Invalid channel (want MAX:MIN:FILE, at most 4 of them): a:b:c"),
    (["--channel", "9:1:a", "--channel", "9:1:b", "--channel", "9:1:c", "--channel", "9:1:d", "--channel", "9:1:e"],
     "Invalid channel (want MAX:MIN:FILE, at most 4 of them): 9:1:e"),
    (["--seed", "18446744073709551615", "--bogus"], "Invalid argument: --bogus"),
    (["-o"], "Missing output file name."),
])
def test_argument_parser_under_sanitizers(built, args, first_line):
    r = subprocess.run([os.path.join(built, "cudabrot_asan"), *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=120, env=ENV)
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    if first_line is not None:
        assert r.returncode == 0                                  # usage always exits 0 (cudabrot.cu:619)
        assert r.stdout.split("\n")[0] == first_line
        assert "Usage: " in r.stdout


def test_run_without_device_exits_one_cleanly_under_sanitizers(built, tmp_path):
    """Past the parser the binary reaches SetupCUDA; without a GPU that is the reference's device-error line and
    exit 1 (cudabrot.cu:134-141) -- with every host buffer released on the way out (ASan would report otherwise)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the no-device exit cannot be observed")
    r = subprocess.run([os.path.join(built, "cudabrot_asan"), "-w", "64", "-h", "48", "--passes", "1", "-s",
                        str(tmp_path / "s.bin"), "-o", str(tmp_path / "o.pgm")], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=120, env=ENV)
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 1
    assert r.stdout.strip().split("\n")[-1].startswith("CUDA error ")
