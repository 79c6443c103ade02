import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def _build_hostsim(src, out):
    """g++ build of the device headers for the CPU (test-only simulator)."""
    csrc = os.path.join(ROOT, "bulletproofs-r1cs-gadgets_amd", "csrc")
    srcp = os.path.join(ROOT, "tests", "hostsim", src)
    outp = os.path.join(ROOT, "tests", "hostsim", "_build", out)
    os.makedirs(os.path.dirname(outp), exist_ok=True)
    deps = [srcp] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")]
    if not os.path.exists(outp) or any(os.path.getmtime(d) > os.path.getmtime(outp) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-DBPR1CS_HOSTSIM", "-I" + csrc, "-shared", "-fPIC", srcp, "-o", outp])
    return outp


@pytest.fixture(scope="session")
def prim_lib():
    import ctypes
    return ctypes.CDLL(_build_hostsim("prim_check.cpp", "prim_check.so"))


@pytest.fixture(scope="session")
def sim_lib():
    import importlib
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    return bp.load_library(_build_hostsim("pipeline_sim.cpp", "libbpr1cs_sim.so"))


@pytest.fixture(scope="session")
def sim_glib(sim_lib):
    """front-end (host/frontend.cpp) linked against the simulator backend"""
    import importlib
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    bdir = os.path.join(ROOT, "tests", "hostsim", "_build")
    src = os.path.join(ROOT, "bulletproofs-r1cs-gadgets_amd", "host", "frontend.cpp")
    out = os.path.join(bdir, "libbpr1cs_gadgets_sim.so")
    hdir = os.path.dirname(src)
    deps = [src] + [os.path.join(hdir, f) for f in os.listdir(hdir)] + [os.path.join(bdir, "libbpr1cs_sim.so")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-O3", "-std=c++17", "-DBPR1CS_HOST_ONLY", "-shared", "-fPIC", src, "-o", out,
                               "-L" + bdir, "-lbpr1cs_sim", "-Wl,-rpath," + bdir])
    return bp.load_gadgets_library(out)


@pytest.fixture(scope="session")
def hip_glib(hip_lib):
    import importlib
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    return bp.load_gadgets_library()


@pytest.fixture(scope="session")
def hip_lib():
    import importlib
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    lib = bp.load_library()
    if lib.bpr1cs_device_count() < 1:
        pytest.fail("no gfx950 device visible: GPU tests must run on the MI355X box")
    return lib
