"""The limb-per-lane field arithmetic of the lone-wavefront kernels (csrc/fe_wide.hpp) against the lane's own on the device.

The kernels that use it (k_commit_wave, k_finish_wave, k_commit_T_wave, k_verify_finish_wave) are behind every small-job parity
test; this one feeds the power chain values of every limb class directly (tools/fe_wide_check.hip: 12 288 comparisons of canonical
encodings, element powers and compressed points)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_wavefront_power_chain_and_compress_equal_the_lane_forms(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "fe_wide_check")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "bulletproofs-r1cs-gadgets_amd", "csrc"),
                    os.path.join(ROOT, "tools", "fe_wide_check.hip"), "-o", exe], check=True, capture_output=True, timeout=600)
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "mismatches 0" in run.stdout, run.stdout
