"""The `cudabrot` binary's process contract that needs no GPU: flags, messages, exit codes
(cudabrot.cu:579-754)."""

import os
import subprocess

import pytest


@pytest.fixture(scope="module")
def exe(repo_root):
    path = os.path.join(repo_root, "cudabrot")
    if not os.access(path, os.X_OK):
        pytest.fail("./cudabrot is not built (run `make` or __graft_entry__.build())")
    return path


def run(exe, *args, **kw):
    return subprocess.run([exe, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, **kw)


def test_help_prints_usage_and_exits_zero(exe):
    r = run(exe, "--help")
    assert r.returncode == 0  # cudabrot.cu:619
    assert r.stdout.startswith("Usage: %s [options]\n\nOptions may be one or more of the following:\n" % exe)
    for flag in ("--help", "-d <device number>", "-o <output file name>", "-m <max escape iterations>",
                 "-c <min escape iterations>", "-g <gamma correction>", "-t <seconds to run>", "-w <width>",
                 "-h <height>", "-s <save/load file>", "--min-real <min real>", "--max-real <max real>",
                 "--min-imag <min imag>", "--max-imag <max imag>"):
        assert flag in r.stdout
    assert r.stdout.rstrip().endswith("include in the output image. Defaults to 2.0.")
    assert r.stderr == ""


@pytest.mark.parametrize(
    "args,first_line",
    [
        (["--bogus"], "Invalid argument: --bogus"),                                   # cudabrot.cu:751
        (["-m"], "Argument -m needs a value."),                                       # :629
        (["-t"], "Argument -t needs a value."),                                       # :648
        (["-o"], "Missing output file name."),                                        # :674
        (["-s"], "Missing in-progress buffer file name."),                            # :683
        (["-m", "12x"], "Invalid number given to argument -m: 12x"),                  # :636
        (["-w", ""], "Invalid number given to argument -w: "),                        # empty string is not a number
        (["-g", "1.5q"], "Invalid number given to argument -g: 1.5q"),                # :653
        (["-w", "0"], "Output width must be positive."),                              # :508
        (["-h", "-4"], "Output height must be positive."),                            # -h is height, not help
        (["--min-real", "2.0"], "Maximum real value must be greater than minimum real value."),
        (["--max-imag", "-2"], "Minimum imaginary value must be greater than maximum imaginary value."),
        # flag ORDER matters: the canvas is re-validated after each flag (cudabrot.cu:704-749)
        (["--min-real", "3", "--max-real", "4"], "Maximum real value must be greater than minimum real value."),
        # extension flags follow the same conventions
        (["--gpus"], "Argument --gpus needs a value."),
        (["--passes", "many"], "Invalid number given to argument --passes: many"),
        (["--state-format", "json"], "Invalid state format (want native or raw): json"),
        (["--state-format"], "Argument --state-format needs a value."),
        (["--channel", "100:20"], "Invalid channel (want MAX:MIN:FILE, at most 4 of them): 100:20"),
        (["--channel", "a:b:c"], "Invalid channel (want MAX:MIN:FILE, at most 4 of them): a:b:c"),
        (["--channel", "9:1:a", "--channel", "9:1:b", "--channel", "9:1:c", "--channel", "9:1:d", "--channel", "9:1:e"],
         "Invalid channel (want MAX:MIN:FILE, at most 4 of them): 9:1:e"),
    ],
)
def test_bad_arguments_print_message_then_usage_and_exit_zero(exe, args, first_line):
    r = run(exe, *args)
    assert r.returncode == 0
    lines = r.stdout.split("\n")
    assert lines[0] == first_line
    assert lines[1] == "Usage: %s [options]" % exe


def test_canvas_flags_in_a_valid_order_pass_validation(exe):
    # same values as the failing case above, other order: parsing succeeds and the run gets as far as
    # the device (exit 1 without a GPU, 0 with one) -- either way no usage text
    r = run(exe, "--max-real", "4", "--min-real", "3", "-w", "8", "-h", "8", "--passes", "0", "-o", os.devnull)
    assert "Usage:" not in r.stdout
    assert r.stdout.startswith("Creating 8x8 image, 100 max iterations.\nCalculating image...\n")


def test_high_iteration_warning(exe):
    r = run(exe, "-m", "60001", "--bogus")
    assert r.stdout.startswith("Warning: Using a high number of iterations may cause the program respond slowly "
                               "to Ctrl+C or time running out.\nInvalid argument: --bogus\n")


def test_banner_and_memory_line(exe):
    r = run(exe, "-w", "4096", "-h", "4096", "-m", "20000", "--passes", "0", "-o", os.devnull)
    lines = r.stdout.split("\n")
    assert lines[0] == "Creating 4096x4096 image, 20000 max iterations."
    assert lines[1] == "Calculating image..."
    # 4096*4096*8 B histogram + 262144*24 B generator states; + 2 B/pixel image on the host
    assert lines[2] == "Approximate memory needed: 134.000 MiB GPU, 160.000 MiB CPU"


def test_without_a_gpu_the_device_error_line_goes_to_stdout_and_exit_is_one(exe):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the no-device path cannot be observed")
    r = run(exe, "-w", "16", "-h", "16", "-t", "0")
    assert r.returncode == 1                                   # cudabrot.cu:140
    assert r.stderr == ""                                      # errors go to stdout (cudabrot.cu:137)
    last = r.stdout.strip().split("\n")[-1]
    assert last.startswith("CUDA error ") and " in " in last and ", line " in last
    assert "Done!" not in r.stdout                             # and certainly no CPU fallback render
