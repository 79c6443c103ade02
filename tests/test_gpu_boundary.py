"""GPU twins of tests/test_boundary.py: the low-level C-ABI entry points on the device."""
import pytest

import boundary_cases as bc

pytestmark = pytest.mark.gpu


def test_ipa_create_matches_oracle(hip_lib):
    bc.check_ipa_create(hip_lib, n=64, unfold=2)   # two rounds from the tables, then variable-base
    bc.check_ipa_create(hip_lib, n=16, unfold=0)
    bc.check_ipa_create(hip_lib, n=128, unfold=0)  # seven variable-base rounds (pairs on one set of multiples, two-level folds)
    bc.check_ipa_create(hip_lib, n=64, unfold=1)
    bc.check_ipa_create(hip_lib, n=8, unfold=9)
    bc.check_ipa_create(hip_lib, n=1, unfold=2)


def test_proof_wire_format(hip_lib):
    bc.check_proof_format(hip_lib)


def test_malformed_inputs_are_refused(hip_lib):
    bc.check_validation(hip_lib)


def test_split_shared_base_verifier(hip_lib, hip_glib):
    bc.check_split_verifier(hip_lib, hip_glib, batch=6)


def test_two_threads_two_handles(hip_lib, hip_glib):
    bc.check_two_threads_two_handles(hip_lib, hip_glib, rounds=4)


def test_out_of_memory_is_an_error_code_not_an_abort(hip_lib):
    """Generator tables that cannot fit (capacity 2^22 at W = 11: ~25 TB) make bpr1cs_gens_create return
    BPR1CS_ERR_OUT_OF_MEMORY; the process and the device stay usable."""
    import ctypes
    bp = bc.bp
    h = ctypes.c_void_p()
    opts = (ctypes.c_int32 * 2)(bp.OPT_WINDOW_BITS, 11)
    assert hip_lib.bpr1cs_gens_create_opts(1 << 22, opts, 1, ctypes.byref(h)) == -19
    g = bp.Gens(16, lib=hip_lib)   # still works
    assert len(g.point(2, 3)) == 32
