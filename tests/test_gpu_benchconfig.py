"""The BENCHMARKED configuration under pytest: W = 11 tables in the storage format bench.py gets,
merged S-box tables, IPA switch round 4, jobs in flight - proof bytes against the C oracle; plus SURVEY §8d configs C3 and
C5 at their full batch sizes against the extended C oracle."""
import importlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "c")])
    from cref import COracle
    return COracle()


@pytest.fixture(scope="module")
def w11_gens(hip_lib):
    """32768-capacity generators with W = 11 tables exactly as bench.py creates them (format chosen automatically)."""
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    bp.release_cached_memory(hip_lib)
    hip_lib.bpr1cs_set_window_bits(11)
    hip_lib.bpr1cs_set_table_format(-1)
    try:
        gens = bp.Gens(32768, lib=hip_lib)
    finally:
        hip_lib.bpr1cs_set_window_bits(8)
    info = gens.table_info()
    assert info["window_bits"] == 11 and info["windows"] == 23
    yield gens
    gens.close()
    bp.release_cached_memory(hip_lib)


def test_vsmt4_depth32_bench_configuration_two_jobs_in_flight(hip_lib, hip_glib, w11_gens):
    """bench.py's workload and settings: VSMT-4 depth 32 (reference src/gadget_vsmt_4.rs:363-482), W = 11, merged S-box
    tables, unfold 4, TWO jobs in flight (1024 proofs and a ragged 1000): ten proofs spread over both jobs - first, last,
    middle, lanes of the ragged last wavefront - equal the C oracle's byte for byte; all pass the device verifier."""
    sys.path.insert(0, ROOT)
    import bench
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    o = _oracle()
    levels, BA, BB = 32, 1024, 1000
    root, values, blindings, seeds, m = bench.build_workload(bp, levels, BA + BB, 64, 0)
    circ = bp.CompiledGadget("vsmt_4", [levels, 140], [root], lib=hip_lib, glib=hip_glib)
    assert (circ.n, circ.q, circ.m, circ.proof_len) == (18656, 43330, 100, 1377) and circ.has_witness_program
    assert hip_lib.bpr1cs_circuit_macro_perms(circ.h) == 32
    hip_lib.bpr1cs_set_unfold_rounds(4)
    cut = BA * m * 32
    jobA = bp.ProveJob(w11_gens, circ, b"VSMT", values[:cut], blindings[:cut], seeds[:32 * BA], BA)
    jobB = bp.ProveJob(w11_gens, circ, b"VSMT", values[cut:], blindings[cut:], seeds[32 * BA:], BB)   # begins while A is in flight
    PA, CA = jobA.finish()
    PB, CB = jobB.finish()
    oc = o.compile_vsmt4(levels, 140, root)
    P = PA + PB
    for j in (0, 511, 1023, BA + 0, BA + 500, BA + 960, BA + 999, 77, BA + 63, BA + 64):
        ref = o.prove_vsmt4(oc, values[j * m * 32:(j + 1) * m * 32], blindings[j * m * 32:(j + 1) * m * 32], seeds[32 * j:32 * j + 32])
        assert P[j] == ref, "proof %d differs from the C oracle" % j
    assert bp.verify_batch(w11_gens, circ, b"VSMT", PB, CB, BB) == [True] * BB
    pt, wf = bp.verify_batch_combined(w11_gens, circ, b"VSMT", PA, CA, BA)
    assert wf and pt == bytes(32)


def test_vsmt4_8_levels_w11_every_proof_of_a_ragged_batch(hip_lib, hip_glib):
    """W = 11 tables, merged S-box tables, mid-size circuit (8 levels: n = 4664, N = 8192): EVERY proof of a 70-proof batch
    (not a multiple of the wavefront size) equals the C oracle's."""
    sys.path.insert(0, ROOT)
    import bench
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    o = _oracle()
    levels, B = 8, 70
    root, values, blindings, seeds, m = bench.build_workload(bp, levels, B, 70, 7)
    circ = bp.CompiledGadget("vsmt_4", [levels, 140], [root], lib=hip_lib, glib=hip_glib)
    assert circ.n == 583 * levels
    bp.release_cached_memory(hip_lib)
    hip_lib.bpr1cs_set_window_bits(11)
    try:
        gens = bp.Gens(8192, lib=hip_lib)
    finally:
        hip_lib.bpr1cs_set_window_bits(8)
    hip_lib.bpr1cs_set_unfold_rounds(4)
    P, C = bp.prove_batch(gens, circ, b"VSMT", values, blindings, seeds, B)
    oc = o.compile_vsmt4(levels, 140, root)
    for j in range(B):
        assert P[j] == o.prove_vsmt4(oc, values[j * m * 32:(j + 1) * m * 32], blindings[j * m * 32:(j + 1) * m * 32], seeds[32 * j:32 * j + 32]), j
    assert bp.verify_batch(gens, circ, b"VSMT", P, C, B) == [True] * B
    gens.close()


def test_vsmt2_depth32_batch_1024_config_c3(hip_lib, hip_glib, w11_gens):
    """SURVEY §8d config C3 at its full batch: binary sparse Merkle tree, depth 32 (reference src/gadget_vsmt_2.rs:171-209;
    n = 18 176, N = 32 768, m = 69), 1024 proofs: sampled proofs equal the C oracle's (gadget restated in oracle/c), all
    verify per proof and batched, a tampered one is named."""
    sys.path.insert(0, ROOT)
    import bench
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    from cref import VSMT_2
    o = _oracle()
    depth, B = 32, 1024
    tree = bp.SparseMerkleTree(2, depth, 140, glib=hip_glib)
    leaves = [(i, i) for i in range(1, 11)]
    seen = {i for i, _ in leaves}
    k = 0
    while len(leaves) < 10 + B:
        idx = bench.synth_scalar(b"l2-idx", k) & 0xffffffff
        k += 1
        if idx not in seen:
            seen.add(idx)
            leaves.append((idx, bench.synth_scalar(b"l2-val", k)))
    tree.update_many(leaves)
    sel = leaves[10:10 + B]
    lv, paths = tree.get_many([i for i, _ in sel])
    sc = bench.sc
    m = 2 * depth + 5
    vals, bls = [], []
    for k, (idx, val) in enumerate(sel):
        nodes = [paths[32 * (depth * k + t):32 * (depth * k + t) + 32] for t in range(depth)]   # root level first
        vals.append(sc(val) + b"".join(sc((idx >> t) & 1) for t in range(depth)) + b"".join(reversed(nodes)) + sc(0) + sc(101) + sc(0) + sc(0))
        bls.append(b"".join(sc(bench.synth_scalar(b"bl2", k * 1024 + t)) for t in range(m - 4)) + bytes(128))   # statics: blinding 0
    values, bl = b"".join(vals), b"".join(bls)
    seeds = b"".join(bench.synth_rng_seed(10**6 + k) for k in range(B))
    root = tree.root()
    circ = bp.CompiledGadget("vsmt_2", [depth, 140], [root], lib=hip_lib, glib=hip_glib)
    assert (circ.n, circ.q, circ.m) == (18176, 42369, 69)
    hip_lib.bpr1cs_set_unfold_rounds(4)
    P, C = bp.prove_batch(w11_gens, circ, b"VSMT", values, bl, seeds, B)
    o.lib.oracle_warm_gens(32768)
    for j in (0, 517, B - 1):
        r = o.prove(VSMT_2, [depth, 140], root, b"VSMT", vals[j], bls[j], seeds[32 * j:32 * j + 32])
        assert (r["n"], r["q"], r["m"]) == (18176, 42369, 69)
        assert P[j] == r["proof"], "proof %d differs from the C oracle" % j
        assert C[j] == r["comms"]
    assert bp.verify_batch(w11_gens, circ, b"VSMT", P, C, B) == [True] * B
    bad = bytearray(P[300]); bad[1 + 32 * 9 + 1] ^= 2
    Pb = P[:300] + [bytes(bad)] + P[301:]
    pt, wf = bp.verify_batch_combined(w11_gens, circ, b"VSMT", Pb, C, B)
    assert wf and pt != bytes(32)
    res = bp.verify_batch(w11_gens, circ, b"VSMT", Pb, C, B)
    assert res[300] is False and sum(res) == B - 1


def test_mimc_set_membership_batch_8192_config_c5(hip_lib, hip_glib):
    """SURVEY §8d config C5 at its per-GPU batch: MiMC-322 preimage + set membership on one prover (n = 665, N = 1024,
    m = 10), 8192 proofs: sampled proofs equal the C oracle's, the cross-proof batched verifier accepts the batch and
    rejects it with two tampered proofs, which the per-proof verifier names; the split (multi-GPU) form agrees."""
    import frontend_cases as fc
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    from pyref import scenarios as S, gadgets as g
    from pyref.ed import sc_to_bytes
    o = _oracle()
    B, rounds = 8192, g.MIMC_ROUNDS
    consts = [S.synth_scalar(b"mimc-const", i) for i in range(rounds)]
    ip = [rounds, len(fc.SET)]
    for x in fc.SET:
        ip += fc._u64(x)
    # one image for the whole batch (the circuit is compiled once): same preimage, per-proof set element / blindings / seeds
    xl, xr = S.synth_scalar(b"ml", 0), S.synth_scalar(b"mr", 0)
    image = g.mimc(xl, xr, consts)
    circ = bp.CompiledGadget("mimc_set_membership", ip, consts + [image], lib=hip_lib, glib=hip_glib)
    assert (circ.n, circ.q, circ.m) == (2 * rounds + 3 * len(fc.SET), 2 * 2 * rounds + 1 + 7 * len(fc.SET) + 2, 2 + len(fc.SET) + 1)
    m = circ.m
    vals, bls = [], []
    for j in range(B):
        v = fc.SET[j % len(fc.SET)]
        vals.append(b"".join(sc_to_bytes(x) for x in [xl, xr] + [1 if e == v else 0 for e in fc.SET] + [v]))
        bls.append(b"".join(sc_to_bytes(S.synth_scalar(b"c5bl", j * 16 + t)) for t in range(m)))
    seeds = b"".join(S.synth_seed(5 * 10**6 + j) for j in range(B))
    hip_lib.bpr1cs_set_window_bits(8)
    hip_lib.bpr1cs_set_unfold_rounds(4)
    gens = bp.Gens(1024, lib=hip_lib)
    P, C = bp.prove_batch(gens, circ, b"MiMC+SetMembership", b"".join(vals), b"".join(bls), seeds, B)
    for j in (0, 4099, B - 1):
        r = o.prove_case("mimc_set_membership", ip, consts + [image], b"MiMC+SetMembership", vals[j], bls[j], seeds[32 * j:32 * j + 32])
        assert P[j] == r["proof"] and C[j] == r["comms"], "proof %d differs from the C oracle" % j
    pt, wf = bp.verify_batch_combined(gens, circ, b"MiMC+SetMembership", P, C, B)
    assert wf and pt == bytes(32)
    bad = list(P)
    for j, off in ((1234, 1 + 8 * 32 + 3), (8000, 1 + 32 * 11 + 7)):
        x = bytearray(bad[j]); x[off] ^= 1; bad[j] = bytes(x)
    pt, wf = bp.verify_batch_combined(gens, circ, b"MiMC+SetMembership", bad, C, B)
    assert pt != bytes(32)
    res = bp.verify_batch(gens, circ, b"MiMC+SetMembership", bad, C, B)
    assert [j for j, ok in enumerate(res) if not ok] == [1234, 8000]
    sh = importlib.import_module("bulletproofs-r1cs-gadgets_amd.sharding")
    assert sh.verify_sharded(bp, gens, circ, b"MiMC+SetMembership", P, C, B, 0, 1, 0) is True
    assert sh.verify_sharded(bp, gens, circ, b"MiMC+SetMembership", bad, C, B, 0, 1, 0) is False


def test_vsmt4_as_shipped_depth_128(hip_lib, hip_glib):
    """The depth the reference ships (TreeDepth = 128, src/gadget_vsmt_4.rs:25,28): n = 74 624, N = 131 072, 388 committed
    values.  Window width chosen by the library from the free memory (W = 11 cannot hold 262 146 bases), a ragged batch of 70:
    first and last proof equal the C oracle's byte for byte, all pass both device verifiers."""
    sys.path.insert(0, ROOT)
    import bench
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    o = _oracle()
    levels, B = 128, 70
    root, values, blindings, seeds, m = bench.build_workload(bp, levels, B, 70, 11)
    assert m == 388
    circ = bp.CompiledGadget("vsmt_4", [levels, 140], [root], lib=hip_lib, glib=hip_glib)
    assert circ.n == 583 * levels and circ.has_witness_program
    bp.release_cached_memory(hip_lib)
    hip_lib.bpr1cs_set_window_bits(0)
    hip_lib.bpr1cs_set_table_format(-1)
    try:
        gens = bp.Gens(131072, lib=hip_lib)
    finally:
        hip_lib.bpr1cs_set_window_bits(8)
    try:
        info = gens.table_info()
        assert info["window_bits"] < 11 and info["bytes"] < 250e9, info
        hip_lib.bpr1cs_set_unfold_rounds(4)
        P, C = bp.prove_batch(gens, circ, b"VSMT", values, blindings, seeds, B)
        assert len(P[0]) == 1 + 32 * (13 + 2 * 17)
        oc = o.compile_vsmt4(levels, 140, root)
        for j in (0, B - 1):
            ref = o.prove_vsmt4(oc, values[j * m * 32:(j + 1) * m * 32], blindings[j * m * 32:(j + 1) * m * 32], seeds[32 * j:32 * j + 32])
            assert P[j] == ref, "proof %d differs from the C oracle" % j
        assert bp.verify_batch(gens, circ, b"VSMT", P, C, B) == [True] * B
        pt, wf = bp.verify_batch_combined(gens, circ, b"VSMT", P, C, B)
        assert wf and pt == bytes(32)
        bad = bytearray(P[3]); bad[40] ^= 1
        assert bp.verify_batch(gens, circ, b"VSMT", [P[0], bytes(bad)], [C[0], C[3]], 2) == [True, False]
    finally:
        gens.close()
        bp.release_cached_memory(hip_lib)



def test_vsmt2_as_shipped_depth_253(hip_lib, hip_glib):
    """The binary tree at the depth the reference ships (TreeDepth = 253, src/gadget_vsmt_2.rs:23): n = 143 704, N = 262 144,
    511 committed values, 524 290 generator bases (window width from the free memory), a ragged batch of 66: the first proof
    equals the C oracle's byte for byte, all pass the device verifier, a tampered one is named."""
    sys.path.insert(0, ROOT)
    import bench
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    from cref import VSMT_2
    o = _oracle()
    depth, B = 253, 66
    tree = bp.SparseMerkleTree(2, depth, 140, glib=hip_glib)
    leaves, seen, k = [(i, i) for i in range(1, 11)], set(range(1, 11)), 0
    while len(leaves) < 10 + B:
        idx = bench.synth_scalar(b"l253-idx", k) & ((1 << 250) - 1)
        k += 1
        if idx not in seen:
            seen.add(idx)
            leaves.append((idx, bench.synth_scalar(b"l253-val", k)))
    tree.update_many(leaves)
    sel = leaves[10:10 + B]
    lv, paths = tree.get_many([i for i, _ in sel])
    sc = bench.sc
    m = 2 * depth + 5
    vals, bls = [], []
    for k, (idx, val) in enumerate(sel):
        nodes = [paths[32 * (depth * k + t):32 * (depth * k + t) + 32] for t in range(depth)]   # root level first
        vals.append(sc(val) + b"".join(sc((idx >> t) & 1) for t in range(depth)) + b"".join(reversed(nodes)) + sc(0) + sc(101) + sc(0) + sc(0))
        bls.append(b"".join(sc(bench.synth_scalar(b"bl253", k * 1024 + t)) for t in range(m - 4)) + bytes(128))   # statics: blinding 0
    seeds = b"".join(bench.synth_rng_seed(2 * 10**6 + k) for k in range(B))
    root = tree.root()
    circ = bp.CompiledGadget("vsmt_2", [depth, 140], [root], lib=hip_lib, glib=hip_glib)
    assert (circ.n, circ.m) == (568 * depth, m)
    bp.release_cached_memory(hip_lib)
    hip_lib.bpr1cs_set_window_bits(0)
    hip_lib.bpr1cs_set_table_format(-1)
    try:
        gens = bp.Gens(262144, lib=hip_lib)
    finally:
        hip_lib.bpr1cs_set_window_bits(8)
    try:
        info = gens.table_info()
        assert info["window_bits"] < 11 and info["bytes"] < 260e9, info
        hip_lib.bpr1cs_set_unfold_rounds(4)
        P, C = bp.prove_batch(gens, circ, b"VSMT", b"".join(vals), b"".join(bls), seeds, B)
        assert len(P[0]) == 1 + 32 * (13 + 2 * 18)
        o.lib.oracle_warm_gens(262144)
        r = o.prove(VSMT_2, [depth, 140], root, b"VSMT", vals[0], bls[0], seeds[:32])
        assert (r["n"], r["m"]) == (568 * depth, m)
        assert P[0] == r["proof"] and C[0] == r["comms"], "proof 0 differs from the C oracle"
        assert bp.verify_batch(gens, circ, b"VSMT", P, C, B) == [True] * B
        bad = bytearray(P[65]); bad[1 + 32 * 9 + 1] ^= 2
        assert bp.verify_batch(gens, circ, b"VSMT", P[:65] + [bytes(bad)], C, B) == [True] * 65 + [False]
    finally:
        gens.close()
        bp.release_cached_memory(hip_lib)
