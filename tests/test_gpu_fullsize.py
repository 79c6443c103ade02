"""BASELINE-size checks (north-star circuit: VSMT-4 depth 32, n = 18 656, N = 32 768, 1377-byte proofs) with a
ragged batch: GPU proofs equal the C oracle's byte for byte; every GPU proof passes the device verifier
(size-independent property); tampering and a wrong root are rejected."""
import importlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_vsmt4_depth32_ragged_batch(hip_lib, hip_glib):
    sys.path.insert(0, ROOT)
    import bench
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "c")])
    from cref import COracle
    B, levels = 70, 32           # not a multiple of the wavefront size
    root, values, blindings, seeds, m = bench.build_workload(bp, levels, B, 12, 0)
    circ = bp.CompiledGadget("vsmt_4", [levels, 140], [root], lib=hip_lib, glib=hip_glib)
    assert (circ.n, circ.q, circ.m, circ.proof_len) == (18656, 43330, 100, 1377)
    hip_lib.bpr1cs_set_window_bits(8)
    hip_lib.bpr1cs_set_unfold_rounds(5)
    try:
        gens = bp.Gens(32768, lib=hip_lib)
    finally:
        hip_lib.bpr1cs_set_window_bits(8)
    P, C = bp.prove_batch(gens, circ, b"VSMT", values, blindings, seeds, B)
    o = COracle()
    oc = o.compile_vsmt4(levels, 140, root)
    for j in (0, B - 1):
        ref = o.prove_vsmt4(oc, values[j * m * 32:(j + 1) * m * 32], blindings[j * m * 32:(j + 1) * m * 32], seeds[32 * j:32 * j + 32])
        assert P[j] == ref, "proof %d differs from the C oracle" % j
    assert bp.verify_batch(gens, circ, b"VSMT", P, C, B) == [True] * B
    # negative: tampered IPA element, swapped commitments, wrong public root
    bad = [bytearray(p) for p in P]
    bad[3][1 + 32 * 20 + 5] ^= 1
    res = bp.verify_batch(gens, circ, b"VSMT", [bytes(p) for p in bad], C, B)
    assert res[3] is False and sum(res) == B - 1
    C2 = [list(c) for c in C]
    C2[5][0], C2[5][1] = C2[5][1], C2[5][0]
    res = bp.verify_batch(gens, circ, b"VSMT", P, C2, B)
    assert res[5] is False and sum(res) == B - 1
    wrong_root = bytes([root[0] ^ 1]) + root[1:]
    circ2 = bp.CompiledGadget("vsmt_4", [levels, 140], [wrong_root], lib=hip_lib, glib=hip_glib)
    assert bp.verify_batch(gens, circ2, b"VSMT", P[:4], C[:4], 4) == [False] * 4
    # a different proving configuration (all IPA rounds from un-folded tables vs early fold) gives the same bytes
    hip_lib.bpr1cs_set_unfold_rounds(2)
    P2, _ = bp.prove_batch(gens, circ, b"VSMT", values[:3 * m * 32], blindings[:3 * m * 32], seeds[:96], 3)
    hip_lib.bpr1cs_set_unfold_rounds(5)
    assert P2 == P[:3]
