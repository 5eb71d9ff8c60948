"""What the compiler makes of the draw kernel that shares its CUs with the scatter (draw_wide.hip), checked where it is
built: hipcc cross-compiles for gfx950 without a GPU and reports every kernel's resources (-Rpass-analysis=kernel-
resource-usage).  DESIGN.md section 4.1b rests on these: two such waves per SIMD beside the scatter's four of 64 registers
(2 x 128 + 4 x 64 = 512), 40 KiB of LDS per workgroup (two workgroups + the region sort's 72.5 KiB in a CU's 160), and --
since round 4, for EVERY instance, the chunked (C4) and the timed ones included -- no vector register spilled, no
scratch: round 3's text said so while four instances spilled seven registers."""

import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cudabrot_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("define", [[], ["-DCB_BURNING_SHIP"]], ids=["mandelbrot", "burning_ship"])
def test_every_instance_of_the_wide_kernel_fits_two_waves_per_simd_without_scratch(tmp_path, define):
    flags = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950", "-S",
             "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run([HIPCC, *flags, *define, "-o", str(tmp_path / "wide.s"), os.path.join(CSRC, "draw_wide.hip")],
                         capture_output=True, text=True, cwd=CSRC)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = [], None
    for line in out.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = {"name": body.split(":", 1)[1].strip()}
            kernels.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            cur[k.strip()] = v.strip()
    wide = [k for k in kernels if "draw_wide_kernel" in k["name"]]
    assert len(wide) == 8, [k["name"] for k in kernels]          # <pow2, timed, chunked>
    for k in wide:
        assert int(k["VGPRs"]) <= 128 and int(k["AGPRs"]) == 0, k
        assert int(k["VGPRs Spill"]) == 0 and int(k["ScratchSize [bytes/lane]"]) == 0, k
        assert int(k["LDS Size [bytes/block]"]) == 40960, k
    shutil.rmtree(tmp_path, ignore_errors=True)
