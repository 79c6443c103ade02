"""GPU: the C++ gadget front-end compiled circuits + DEVICE witness programs (Poseidon
S-box / MDS synthesis, sparse-Merkle selection logic) vs the oracle, bit-exact."""
import pytest

import frontend_cases as fc

pytestmark = pytest.mark.gpu


def test_native_hashes_and_trees(hip_glib):
    fc.check_native_hashes(hip_glib)
    fc.check_trees(hip_glib, levels4=4, depth2=3, partial_rounds=2)


@pytest.mark.parametrize("case", ["bound_check", "bound_check_64", "set_membership", "factors", "range_proof", "is_zero", "not_equals",
                                  "set_membership_1", "set_non_membership"])
def test_compiled_small(hip_lib, hip_glib, case):
    fc.check_compiled(hip_lib, hip_glib, case, batch=3, unfold=2)


@pytest.mark.parametrize("case", ["poseidon_hash_2_cube", "poseidon_hash_2_inverse", "poseidon_hash_4_inverse"])
def test_compiled_poseidon(hip_lib, hip_glib, case):
    fc.check_compiled(hip_lib, hip_glib, case, batch=2, unfold=4)


def test_compiled_vsmt_2(hip_lib, hip_glib):
    fc.check_compiled(hip_lib, hip_glib, "vsmt_2_d3", batch=2, unfold=4)


def test_compiled_vsmt_4_four_levels(hip_lib, hip_glib):
    # 4 levels of the north-star circuit with the full 148-round Poseidon: n=2332, N=4096
    fc.check_compiled(hip_lib, hip_glib, "vsmt_4_l4", batch=2, unfold=4)


@pytest.mark.parametrize("case", ["vsmt_4_cube", "vsmt_2_cube"])
def test_compiled_tree_gadgets_with_the_cube_sbox(hip_lib, hip_glib, case):
    """Cube-S-box variants of both tree gadgets (4 levels x 148 rounds: n = 1580, N = 2048; depth 3)"""
    fc.check_compiled(hip_lib, hip_glib, case, batch=2, unfold=4)
    fc.check_prove_verify_roundtrip(hip_lib, hip_glib, case)


def test_compiled_mimc_322_rounds(hip_lib, hip_glib):
    # config 5's preimage half: MiMC-322, n = 644, N = 1024 (per-proof image => one circuit, batch of 1)
    fc.check_compiled(hip_lib, hip_glib, "mimc", batch=1, unfold=4)
    fc.check_prove_verify_roundtrip(hip_lib, hip_glib, "mimc", batch=1)


def test_prover_single_host_synthesis(hip_glib):
    fc.check_prove_single(hip_glib, "bound_check")
    fc.check_prove_single(hip_glib, "poseidon_hash_2_cube")


@pytest.mark.parametrize("case", ["bound_check_64", "set_membership", "poseidon_hash_2_inverse", "vsmt_2_d3", "vsmt_4_l4"])
def test_prove_verify_roundtrip_on_device(hip_lib, hip_glib, case):
    fc.check_prove_verify_roundtrip(hip_lib, hip_glib, case)


@pytest.mark.parametrize("case", ["poseidon_hash_2_inverse", "poseidon_hash_2_inverse_pr1_zero", "vsmt_4_l4"])
def test_poseidon_joint_evaluation_equals_plain_program(hip_lib, hip_glib, case):
    """team kernel: the annotated program (poseidon_team: fractions over a common denominator, one inversion per
    permutation) and the plain op-by-op program (one inversion per S-box) give the oracle's proof bytes, including
    an S-box input of 0."""
    fc.check_macro_vs_plain(hip_lib, hip_glib, case, batch=2 if not case.endswith("pr1_zero") else 1)


def test_bulk_poseidon_and_tree_construction_on_device(hip_lib, hip_glib):
    """N2: k_poseidon_team (8 lanes per permutation, one inversion each) and the level-by-level tree builder"""
    fc.check_bulk_tree(hip_lib, hip_glib, levels=8, partial_rounds=140, count=40)


def test_config_c5_mimc_plus_set_membership(hip_lib, hip_glib):
    """MiMC-322 preimage + set membership (k = 7) on one prover: n = 665, N = 1024, m = 10"""
    ob, P, C = fc.check_compiled(hip_lib, hip_glib, "mimc_set_membership", batch=2, unfold=4)
    assert (ob["n"], ob["m"]) == (665, 10)
    fc.check_prove_verify_roundtrip(hip_lib, hip_glib, "mimc_set_membership", batch=1)


def test_zero_sbox_input_inside_a_full_wavefront_batch(hip_lib, hip_glib):
    """A batch of 96 proofs (k_msm_fixed2's wavefront-per-64-proofs path, not the small-batch kernel) in which ONE proof has an
    Inverse-S-box input of 0 (upstream's invert(0) = 0: its a_O wire is 0 where every other proof has 1): the `a_O - 1` form
    of the A_O sum (MSM_MINUS_ONE) must take its non-skipped branch for that wavefront.  All 96 proofs equal the C oracle's
    (95 of them prove a false statement - the public output belongs to proof 40 - which does not matter for the bytes)."""
    import subprocess, os, importlib
    from pyref import scenarios as S, gadgets as g
    from pyref.ed import sc_to_bytes, L
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "c")])
    from cref import COracle, POSEIDON_HASH_2
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    o = COracle()
    pr, B, special = 1, 96, 40
    params = S.poseidon_params(pr)
    xz = (-params.round_keys[1]) % L            # first-round S-box input of element 1 = xl + round_key[1] = 0
    xs = [(xz if j == special else S.synth_scalar(b"zx", j), S.synth_scalar(b"zy", j)) for j in range(B)]
    out = g.Poseidon_hash_2(xs[special][0], xs[special][1], params, g.INVERSE)
    vals = [b"".join(sc_to_bytes(x) for x in (a, b, 0, 101, 0, 0)) for a, b in xs]
    bls = [sc_to_bytes(S.synth_scalar(b"zb", 2 * j)) + sc_to_bytes(S.synth_scalar(b"zb", 2 * j + 1)) + bytes(128) for j in range(B)]
    seeds = [S.synth_seed(7 * 10**6 + j) for j in range(B)]
    circ = bp.CompiledGadget("poseidon_hash_2", [1, pr], [out], lib=hip_lib, glib=hip_glib)
    assert circ.n == 147 and hip_lib.bpr1cs_circuit_macro_perms(circ.h) == 1
    gens = bp.Gens(256, lib=hip_lib)
    P, C = bp.prove_batch(gens, circ, b"Poseidon_hash_2", b"".join(vals), b"".join(bls), b"".join(seeds), B)
    for j in range(B):
        r = o.prove(POSEIDON_HASH_2, [1, pr], sc_to_bytes(out), b"Poseidon_hash_2", vals[j], bls[j], seeds[j], want_wires=(j == special))
        assert P[j] == r["proof"], "proof %d differs from the C oracle" % j
        if j == special:   # the exceptional wires are really there: a_L = 0 and a_O = 0 at the first S-box triple of element 1
            n = r["n"]
            aO = [r["wires"][32 * (2 * n + i):32 * (2 * n + i) + 32] for i in range(n)]
            assert sum(1 for w in aO if w == sc_to_bytes(1)) == 2 * 49 - 2   # two of the 98 "always 1" output wires are 0 in this proof
    # (no proof of this batch verifies: is_nonzero_gadget, src/gadget_zero_nonzero.rs:46-66, demands x * x^-1 = 1, which an S-box
    # input of 0 cannot satisfy, and the other 95 claim proof 40's output - byte parity is the point here)
    assert bp.verify_batch(gens, circ, b"Poseidon_hash_2", [P[special], P[0]], [C[special], C[0]], 2) == [False, False]


@pytest.mark.parametrize("case,batch", [("bound_check_64", 9), ("poseidon_hash_2_inverse", 3), ("vsmt_4_l4", 3)])
def test_reference_call_shape_on_shared_generators_on_device(hip_lib, hip_glib, case, batch):
    """ONE proof per prove() (bpr1cs_gadget_prove_on, batch 1 = what tools/rust_shim/prover.rs does: per-commit device calls, circuit
    create / cache hit, bpr1cs_prove_batch_transcripts with host wires; every IPA round from the tables for such small jobs), then
    `batch` witnesses synthesised on host threads and proved by one device call: the oracle's bytes"""
    fc.check_prove_on(hip_lib, hip_glib, case, batch)
