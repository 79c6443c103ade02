"""host/scalar_host.hpp (64-bit limbs: Montgomery product, add, sub, safegcd inversion on 62-bit limbs - what the C++ front-end's
Scalar uses) against csrc/sc.hpp (the device's arithmetic, itself pinned to the oracle by tests/test_hostsim_prims.py): 200 000
random pairs, the edge values 0, 1, l-1, 2, l-2, R^2 and sparse single-word values, side by side in one C++ program."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_scalar_arithmetic_equals_the_device_code():
    src = os.path.join(ROOT, "tests", "hostsim", "host_scalar_check.cpp")
    out = os.path.join(ROOT, "tests", "hostsim", "_build", "host_scalar_check")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DBPR1CS_HOST_ONLY", src, "-o", out])
    r = subprocess.run([out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr
