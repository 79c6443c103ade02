"""The BENCHMARKED configuration under pytest: W = 11 tables in the storage format bench.py gets, merged S-box tables, IPA
switch round 4, jobs in flight (shared back-phase scratch, IPA tail on the job's own stream) - EVERY proof of every batch
against the committed SHA-256 digests of the C oracle's proofs (tests/golden/fullsize_digests.json: generated in the build
container, no oracle runs on the GPU box); SURVEY §8d configs C3 and C5 at their full batch sizes and the depths the
reference ships likewise."""
import importlib
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def w11_gens(hip_lib):
    """32768-capacity generators exactly as bench.py and any plain caller get them: bpr1cs_gens_create with no options picks
    W = 11 from the free memory of the device."""
    import gc
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    gc.collect()                       # handles of earlier tests that are no longer referenced give their tables back
    bp.release_cached_memory(hip_lib)
    gens = bp.Gens(32768, lib=hip_lib)
    info = gens.table_info()
    assert info["window_bits"] == 11 and info["windows"] == 23
    yield gens
    gens.close()
    bp.release_cached_memory(hip_lib)


def test_vsmt4_depth32_bench_configuration_two_jobs_in_flight(hip_lib, hip_glib, w11_gens):
    """bench.py's workload and settings: VSMT-4 depth 32 (reference src/gadget_vsmt_4.rs:363-482), W = 11, merged S-box
    tables, unfold 4, TWO jobs in flight (1024 proofs and a ragged 1000) sharing the back-phase arena: ALL 2024 proofs equal
    the C oracle's (digest fixture), as do the commitments; all pass the device verifier."""
    import fullsize_cases as fc
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    BA, BB = 1024, 1000
    case = fc.CASES["c4_vsmt4_d32_x2024"](bp, hip_glib)
    m, values, blindings, seeds = case["m"], case["values"], case["blindings"], case["seeds"]
    circ = bp.CompiledGadget("vsmt_4", case["ip"], case["sp"], lib=hip_lib, glib=hip_glib)
    assert (circ.n, circ.q, circ.m, circ.proof_len) == (18656, 43330, 100, 1377) and circ.has_witness_program
    assert hip_lib.bpr1cs_circuit_macro_perms(circ.h) == 32
    cut = BA * m * 32
    jobA = bp.ProveJob(w11_gens, circ, b"VSMT", values[:cut], blindings[:cut], seeds[:32 * BA], BA)
    jobB = bp.ProveJob(w11_gens, circ, b"VSMT", values[cut:], blindings[cut:], seeds[32 * BA:], BB)   # begins while A is in flight
    PA, CA = jobA.finish()
    PB, CB = jobB.finish()
    fc.check_digests("c4_vsmt4_d32_x2024", case, PA + PB, CA + CB)
    assert bp.verify_batch(w11_gens, circ, b"VSMT", PB, CB, BB) == [True] * BB
    pt, wf = bp.verify_batch_combined(w11_gens, circ, b"VSMT", PA, CA, BA)
    assert wf and pt == bytes(32)
    # ONE call over all 2024 proofs, as bench.py makes it: the library cuts the batch into device jobs by itself (2048 proofs fit
    # next to the tables: one job here), the last IPA rounds on the heavy stream instead of the job's tail stream: the same bytes
    w11_gens.release_scratch()   # the arena is sized for 1024-proof jobs: growing it next to 234 GB of tables would hold both sizes for a moment
    w11_gens.set_option("tail_rounds", 0)
    try:
        P2, C2 = bp.prove_batch(w11_gens, circ, b"VSMT", values, blindings, seeds, BA + BB)
    finally:
        w11_gens.set_option("tail_rounds", -1)
    assert P2 == PA + PB and C2 == CA + CB
    st = bp.last_prove_stats(hip_lib)
    assert st["jobs"] == 1 and st["job_proofs"] == 2024 and st["msm_launches"] == 7
    # ... and cut into FOUR jobs (512, 512, 512, 488 proofs), two in flight (what a larger batch or a smaller device gets): the same bytes
    w11_gens.release_scratch()
    w11_gens.set_option("job_proofs", 512)
    try:
        P3, C3 = bp.prove_batch(w11_gens, circ, b"VSMT", values, blindings, seeds, BA + BB)
    finally:
        w11_gens.set_option("job_proofs", -1)
    assert P3 == PA + PB and C3 == CA + CB
    st = bp.last_prove_stats(hip_lib)
    assert st["jobs"] == 4 and st["job_proofs"] == 512 and st["msm_launches"] == 28
    w11_gens.release_scratch()


def test_vsmt4_8_levels_w11_every_proof_of_a_ragged_batch(hip_lib, hip_glib):
    """W = 11 tables, merged S-box tables, mid-size circuit (8 levels: n = 4664, N = 8192): EVERY proof of a 70-proof batch
    (not a multiple of the wavefront size) equals the C oracle's (digest fixture)."""
    import fullsize_cases as fc
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    case = fc.CASES["vsmt4_l8_x70"](bp, hip_glib)
    B = case["B"]
    circ = bp.CompiledGadget("vsmt_4", case["ip"], case["sp"], lib=hip_lib, glib=hip_glib)
    assert circ.n == 583 * 8
    bp.release_cached_memory(hip_lib)
    gens = bp.Gens(8192, lib=hip_lib, window_bits=11)
    P, C = bp.prove_batch(gens, circ, b"VSMT", case["values"], case["blindings"], case["seeds"], B)
    fc.check_digests("vsmt4_l8_x70", case, P, C)
    assert bp.verify_batch(gens, circ, b"VSMT", P, C, B) == [True] * B
    # (default at N = 8192: the product scalars of the un-folded rounds are produced inside the MSM kernel's term fetch, MsmGeo) the same bytes
    # with the scalars written out by their own kernel (2) and with the factor vectors as arrays (1)
    for fv in (2, 1):
        gens.set_option("factor_vectors", fv)
        Pv, _ = bp.prove_batch(gens, circ, b"VSMT", case["values"], case["blindings"], case["seeds"], B)
        assert Pv == P, fv
    gens.set_option("factor_vectors", -1)
    # every proof from its caller's own transcript (bpr1cs_prove_batch_transcripts, Prover::new(&pc_gens, &mut transcript)): the same
    # bytes when the transcripts are fresh Transcript::new(b"VSMT"), and every transcript advances
    ts = [bp.Transcript(b"VSMT", lib=hip_lib) for _ in range(B)]
    P1, C1 = bp.prove_batch_transcripts(gens, circ, ts, case["values"], case["blindings"], case["seeds"], B)
    assert P1 == P and C1 == C
    fresh = bp.Transcript(b"VSMT", lib=hip_lib).challenge_bytes(b"probe", 32)
    after = [t.challenge_bytes(b"probe", 32) for t in ts]
    assert len(set(after)) == B and fresh not in after
    gens.close()


def test_vsmt2_depth32_batch_1024_config_c3(hip_lib, hip_glib, w11_gens):
    """SURVEY §8d config C3 at its full batch: binary sparse Merkle tree, depth 32 (reference src/gadget_vsmt_2.rs:171-209;
    n = 18 176, N = 32 768, m = 69), 1024 proofs: ALL equal the C oracle's (digest fixture; gadget restated in oracle/c), all
    verify per proof and batched, a tampered one is named."""
    import fullsize_cases as fc
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    case = fc.CASES["c3_vsmt2_d32_x1024"](bp, hip_glib)
    B = case["B"]
    circ = bp.CompiledGadget("vsmt_2", case["ip"], case["sp"], lib=hip_lib, glib=hip_glib)
    assert (circ.n, circ.q, circ.m) == (18176, 42369, 69)
    P, C = bp.prove_batch(w11_gens, circ, b"VSMT", case["values"], case["blindings"], case["seeds"], B)
    fc.check_digests("c3_vsmt2_d32_x1024", case, P, C)
    assert bp.verify_batch(w11_gens, circ, b"VSMT", P, C, B) == [True] * B
    bad = bytearray(P[300]); bad[1 + 32 * 9 + 1] ^= 2
    Pb = P[:300] + [bytes(bad)] + P[301:]
    pt, wf = bp.verify_batch_combined(w11_gens, circ, b"VSMT", Pb, C, B)
    assert wf and pt != bytes(32)
    res = bp.verify_batch(w11_gens, circ, b"VSMT", Pb, C, B)
    assert res[300] is False and sum(res) == B - 1


def test_mimc_set_membership_batch_8192_config_c5(hip_lib, hip_glib):
    """SURVEY §8d config C5 at its per-GPU batch: MiMC-322 preimage + set membership on one prover (n = 665, N = 1024,
    m = 10), 8192 proofs: ALL equal the C oracle's (digest fixture), the cross-proof batched verifier accepts the batch and
    rejects it with two tampered proofs, which the per-proof verifier names; the split (multi-GPU) form agrees."""
    import fullsize_cases as fc
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    case = fc.CASES["c5_mimc_set_x8192"](bp, hip_glib)
    B, rounds = case["B"], fc.MIMC_ROUNDS
    circ = bp.CompiledGadget("mimc_set_membership", case["ip"], case["sp"], lib=hip_lib, glib=hip_glib)
    assert (circ.n, circ.q, circ.m) == (2 * rounds + 3 * len(fc.SET), 2 * 2 * rounds + 1 + 7 * len(fc.SET) + 2, 2 + len(fc.SET) + 1)
    gens = bp.Gens(1024, lib=hip_lib, window_bits=8)
    P, C = bp.prove_batch(gens, circ, case["label"], case["values"], case["blindings"], case["seeds"], B)
    fc.check_digests("c5_mimc_set_x8192", case, P, C)
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
    EVERY proof equals the C oracle's (digest fixture), all pass both device verifiers."""
    import fullsize_cases as fc
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    case = fc.CASES["vsmt4_d128_x70"](bp, hip_glib)
    B = case["B"]
    assert case["m"] == 388
    circ = bp.CompiledGadget("vsmt_4", case["ip"], case["sp"], lib=hip_lib, glib=hip_glib)
    assert circ.n == 583 * 128 and circ.has_witness_program
    bp.release_cached_memory(hip_lib)
    gens = bp.Gens(131072, lib=hip_lib)
    try:
        info = gens.table_info()
        assert info["window_bits"] < 11 and info["bytes"] < 250e9, info
        P, C = bp.prove_batch(gens, circ, b"VSMT", case["values"], case["blindings"], case["seeds"], B)
        assert len(P[0]) == 1 + 32 * (13 + 2 * 17)
        fc.check_digests("vsmt4_d128_x70", case, P, C)
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
    511 committed values, 524 290 generator bases (window width from the free memory), a ragged batch of 66: EVERY proof
    equals the C oracle's (digest fixture), all pass the device verifier, a tampered one is named."""
    import fullsize_cases as fc
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    case = fc.CASES["vsmt2_d253_x66"](bp, hip_glib)
    B, depth = case["B"], 253
    circ = bp.CompiledGadget("vsmt_2", case["ip"], case["sp"], lib=hip_lib, glib=hip_glib)
    assert (circ.n, circ.m) == (568 * depth, 2 * depth + 5)
    bp.release_cached_memory(hip_lib)
    gens = bp.Gens(262144, lib=hip_lib)
    try:
        info = gens.table_info()
        assert info["window_bits"] < 11 and info["bytes"] < 260e9, info
        P, C = bp.prove_batch(gens, circ, b"VSMT", case["values"], case["blindings"], case["seeds"], B)
        assert len(P[0]) == 1 + 32 * (13 + 2 * 18)
        fc.check_digests("vsmt2_d253_x66", case, P, C)
        assert bp.verify_batch(gens, circ, b"VSMT", P, C, B) == [True] * B
        bad = bytearray(P[65]); bad[1 + 32 * 9 + 1] ^= 2
        assert bp.verify_batch(gens, circ, b"VSMT", P[:65] + [bytes(bad)], C, B) == [True] * 65 + [False]
    finally:
        gens.close()
        bp.release_cached_memory(hip_lib)


def test_two_circuits_alternating_in_small_jobs_on_one_handle(hip_lib, hip_glib):
    """One generator handle, two circuits of very different shape (8-level VSMT-4: n = 4664, N = 8192; Poseidon 2:1 Cube: n = 376),
    each batch cut into 64-proof jobs with two in flight: the handle's arenas are grown, shrunk and shared across circuits, the merged
    tables of both circuits live side by side - EVERY proof against the digest fixture, three rounds."""
    import fullsize_cases as fc
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    ca = fc.CASES["vsmt4_l8_x70"](bp, hip_glib)
    cb = fc.CASES["c2_poseidon2_cube_x4096"](bp, hip_glib)
    circa = bp.CompiledGadget("vsmt_4", ca["ip"], ca["sp"], lib=hip_lib, glib=hip_glib)
    circb = bp.CompiledGadget("poseidon_hash_2", cb["ip"], cb["sp"], lib=hip_lib, glib=hip_glib)
    gens = bp.Gens(8192, lib=hip_lib, window_bits=8, job_proofs=64)
    nb, J = 300, 64
    for rnd in range(3):
        P, C = bp.prove_batch(gens, circa, b"VSMT", ca["values"], ca["blindings"], ca["seeds"], ca["B"])
        fc.check_digests("vsmt4_l8_x70", ca, P, C)
        assert bp.last_prove_stats(hip_lib)["jobs"] == -(-ca["B"] // J)
        P, _ = bp.prove_batch(gens, circb, cb["label"], cb["values"][:nb * 6 * 32], cb["blindings"][:nb * 6 * 32], cb["seeds"][:nb * 32], nb)
        fc.check_digests("c2_poseidon2_cube_x4096", cb, P, first=nb)
        assert bp.last_prove_stats(hip_lib)["jobs"] == -(-nb // J)
        if rnd == 1:
            J = 128
            gens.set_option("job_proofs", J)
    gens.close()
