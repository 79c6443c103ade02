"""BASELINE-size checks (north-star circuit: VSMT-4 depth 32, n = 18 656, N = 32 768, 1377-byte proofs) with a
ragged batch: GPU proofs equal the C oracle's byte for byte; every GPU proof passes the device verifier
(size-independent property); tampering and a wrong root are rejected."""
import importlib
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_vsmt4_depth32_ragged_batch(hip_lib, hip_glib):
    import fullsize_cases as fc
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    B, levels = 70, 32           # not a multiple of the wavefront size: the first 70 proofs of the C4 digest fixture
    case = fc.vsmt4(bp, hip_glib, levels, B, 64, 0)
    root, values, blindings, seeds, m = case["sp"][0], case["values"], case["blindings"], case["seeds"], case["m"]
    circ = bp.CompiledGadget("vsmt_4", [levels, 140], [root], lib=hip_lib, glib=hip_glib)
    assert (circ.n, circ.q, circ.m, circ.proof_len) == (18656, 43330, 100, 1377)
    gens = bp.Gens(32768, lib=hip_lib, window_bits=8, unfold=5)
    P, C = bp.prove_batch(gens, circ, b"VSMT", values, blindings, seeds, B)
    fc.check_digests("c4_vsmt4_d32_x2024", case, P, first=B)   # W = 8 tables, IPA switch round 5: the same bytes as the W = 11 / round 4 run
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
    gens.set_option("unfold", 2)
    P2, _ = bp.prove_batch(gens, circ, b"VSMT", values[:3 * m * 32], blindings[:3 * m * 32], seeds[:96], 3)
    assert P2 == P[:3]


def test_vsmt2_depth32_config_c3(hip_lib, hip_glib):
    """binary sparse Merkle tree, depth 32 (n = 18 176, N = 32 768, m = 69).  No oracle runs at
    this size in the test budget, so size-independent properties: every proof verifies (per proof and batched), tampering
    is rejected, and the bytes do not depend on HOW they were computed (annotated vs plain witness program, IPA switch
    round) - the small-depth twin of this circuit is compared with the oracle in test_gpu_frontend.py."""
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    sys.path.insert(0, ROOT)
    import bench
    depth, B = 32, 6
    tree = bp.SparseMerkleTree(2, depth, 140, glib=hip_glib)
    leaves = [(i, i) for i in range(1, 11)] + [(bench.synth_scalar(b"l2-idx", k) & 0xffffffff, bench.synth_scalar(b"l2-val", k)) for k in range(B)]
    tree.update_many(leaves)
    sel = leaves[10:10 + B]
    lv, paths = tree.get_many([i for i, _ in sel])
    sc = bench.sc
    values, bl = b"", b""
    m = 2 * depth + 5
    for k, (idx, val) in enumerate(sel):
        assert lv[32 * k:32 * k + 32] == sc(val)
        nodes = [paths[32 * (depth * k + t):32 * (depth * k + t) + 32] for t in range(depth)]   # root level first
        values += sc(val) + b"".join(sc((idx >> t) & 1) for t in range(depth)) + b"".join(reversed(nodes)) + sc(0) + sc(101) + sc(0) + sc(0)
        bl += b"".join(sc(bench.synth_scalar(b"bl2", k * 1024 + t)) for t in range(m - 4)) + bytes(128)  # statics: blinding 0
    seeds = b"".join(bytes([k + 1]) * 32 for k in range(B))
    root = tree.root()
    gens = bp.Gens(32768, lib=hip_lib, window_bits=8)
    out = {}
    try:
        for macro, unfold in ((1, 4), (0, 2)):
            os.environ["BPR1CS_WITNESS_MACRO"] = str(macro)   # test knob of bpr1cs_circuit_create
            gens.set_option("unfold", unfold)
            circ = bp.CompiledGadget("vsmt_2", [depth, 140], [root], lib=hip_lib, glib=hip_glib)
            assert (circ.n, circ.q, circ.m) == (18176, 42369, 69)
            assert (hip_lib.bpr1cs_circuit_macro_perms(circ.h) > 0) == bool(macro)
            out[macro] = bp.prove_batch(gens, circ, b"VSMT", values, bl, seeds, B)
    finally:
        os.environ.pop("BPR1CS_WITNESS_MACRO", None)
    P, C = out[1]
    assert out[0][0] == P and out[0][1] == C
    assert bp.verify_batch(gens, circ, b"VSMT", P, C, B) == [True] * B
    pt, wf = bp.verify_batch_combined(gens, circ, b"VSMT", P, C, B, bytes(range(32)))
    assert wf and pt == bytes(32)
    bad = bytearray(P[2]); bad[1 + 32 * 9 + 1] ^= 2
    res = bp.verify_batch(gens, circ, b"VSMT", P[:2] + [bytes(bad)] + P[3:], C, B)
    assert res[2] is False and sum(res) == B - 1


def test_poseidon_2to1_cube_batch_4096_config_c2(hip_lib, hip_glib):
    """Poseidon 2:1 Cube preimage (n = 376, N = 512, m = 6), 4096 proofs in one batch: ALL proofs equal
    the C oracle's (digest fixture), all of them pass the per-proof and the batched device verifier."""
    import fullsize_cases as fc
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    case = fc.CASES["c2_poseidon2_cube_x4096"](bp, hip_glib)
    B, label = case["B"], case["label"]
    circ = bp.CompiledGadget("poseidon_hash_2", case["ip"], case["sp"], lib=hip_lib, glib=hip_glib)
    assert (circ.n, circ.m) == (376, 6)
    gens = bp.Gens(512, lib=hip_lib, window_bits=8)
    P, C = bp.prove_batch(gens, circ, label, case["values"], case["blindings"], case["seeds"], B)
    fc.check_digests("c2_poseidon2_cube_x4096", case, P, C)
    assert bp.verify_batch(gens, circ, label, P, C, B) == [True] * B
    pt, wf = bp.verify_batch_combined(gens, circ, label, P, C, B, bytes(range(32)))
    assert wf and pt == bytes(32)
