"""The shipped libraries load without a GPU and export every symbol the headers declare;
compute entry points fail loudly (no CPU fallback)."""
import ctypes
import importlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bpr1cs_[a-z0-9_]+)\s*\(", src)))


def test_libraries_export_every_declared_symbol():
    import __graft_entry__ as ge
    bp = ge.build()
    lib = ctypes.CDLL(bp.LIB_PATH)
    for sym in _declared("bpr1cs.h"):
        assert hasattr(lib, sym), sym
    glib = ctypes.CDLL(bp.GADGETS_LIB_PATH)
    for sym in _declared("bpr1cs_gadgets.h"):
        if sym in _declared("bpr1cs.h"):
            continue
        assert hasattr(glib, sym), sym


def test_no_cpu_fallback_without_device():
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    lib = bp.load_library()
    if lib.bpr1cs_device_count() > 0:
        return  # on a GPU box this check is vacuous
    h = ctypes.c_void_p()
    assert lib.bpr1cs_gens_create(16, ctypes.byref(h)) == -16  # BPR1CS_ERR_NO_DEVICE
    try:
        bp.Gens(16)
        assert False, "expected R1CSError"
    except bp.R1CSError as e:
        assert e.code == -16
