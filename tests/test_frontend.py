"""C++ gadget front-end (host/) vs the oracle's restatement of the reference gadgets.
Backend = the CPU simulator of the device code, so this runs without a GPU; the same
checks run against the HIP backend in test_gpu_frontend.py."""
import os

import pytest

from pyref import scenarios as S, gadgets as g
from pyref.ed import sc_to_bytes
import common
import frontend_cases as fc

bp = common.bp


def test_native_poseidon_matches_oracle(sim_glib):
    fc.check_native_hashes(sim_glib)


def test_trees_match_oracle(sim_glib):
    fc.check_trees(sim_glib, levels4=4, depth2=3, partial_rounds=2)


@pytest.mark.parametrize("case", ["bound_check", "set_membership", "factors", "range_proof", "is_zero", "is_zero_violated", "not_equals",
                                  "set_membership_1", "set_non_membership"])
def test_compiled_gadget_batch(sim_lib, sim_glib, case):
    fc.check_compiled(sim_lib, sim_glib, case, batch=2)


def test_compiled_poseidon_cube_small(sim_lib, sim_glib):
    fc.check_compiled(sim_lib, sim_glib, "poseidon_hash_2_cube_pr1", batch=2)


def test_prover_single(sim_lib, sim_glib):
    fc.check_prove_single(sim_glib, "bound_check")
    fc.check_prove_single(sim_glib, "set_membership")


@pytest.mark.parametrize("case,batch", [("bound_check", 5), ("set_membership", 3), ("poseidon_hash_2_inverse_pr1", 3), ("vsmt_4_pr2_cube", 2)])
def test_reference_call_shape_on_shared_generators(sim_lib, sim_glib, case, batch):
    """one proof per prove() on generators created once (bpr1cs_gadget_prove_on), and several witnesses per call"""
    fc.check_prove_on(sim_lib, sim_glib, case, batch)


@pytest.mark.parametrize("case", ["bound_check", "set_membership", "not_equals", "set_non_membership"])
def test_prove_verify_roundtrip(sim_lib, sim_glib, case):
    fc.check_prove_verify_roundtrip(sim_lib, sim_glib, case)


def test_compiled_poseidon_inverse_joint_evaluation(sim_lib, sim_glib):
    """Inverse-S-box permutation: the annotated witness program (all S-boxes from one inversion, poseidon_team in
    csrc/kernels.hpp) and the plain op-by-op program must both reproduce the oracle's proof bytes."""
    fc.check_macro_vs_plain(sim_lib, sim_glib, "poseidon_hash_2_inverse_pr1", batch=2)


def test_compiled_poseidon_inverse_zero_sbox_input(sim_lib, sim_glib):
    """x = 0 at an S-box (Scalar::invert(0) = 0 upstream): both wires must be 0 and the other S-boxes unaffected."""
    fc.check_macro_vs_plain(sim_lib, sim_glib, "poseidon_hash_2_inverse_pr1_zero", batch=1)


def test_bulk_poseidon_and_tree_construction(sim_lib, sim_glib):
    fc.check_bulk_tree(sim_lib, sim_glib)


def test_compiled_mimc_plus_set_membership_small(sim_lib, sim_glib):
    """SURVEY §8d config C5 at 8 MiMC rounds (the 322-round circuit runs in the GPU suite)"""
    fc.check_compiled(sim_lib, sim_glib, "mimc_set_membership_r8", batch=2)


@pytest.mark.parametrize("case", ["vsmt_4_pr2_cube", "vsmt_2_cube"])
def test_compiled_tree_gadgets_with_the_cube_sbox(sim_lib, sim_glib, case):
    """the sparse-Merkle gadgets over Poseidon with the Cube S-box (`sbox` iparam; the reference hard-wires
    Inverse at src/gadget_vsmt_4.rs:301, src/gadget_vsmt_2.rs:203): proof bytes equal the oracle's"""
    fc.check_compiled(sim_lib, sim_glib, case, batch=2)


def test_job_memory_knobs_do_not_change_a_byte(sim_lib, sim_glib):
    """N = 512 (9 IPA rounds, folded generators at round 4, tail hand-off at round 6): private scratch instead of the shared
    back-phase arena, no tail hand-off, a different hand-off round - the oracle's proof bytes every time"""
    for shared, tail in ((0, 7), (1, 0), (1, 3), (0, 0)):
        fc.check_compiled(sim_lib, sim_glib, "vsmt_2_cube", batch=2, shared_back=shared, tail_rounds=tail)
    for unfold in (4, 0, 9):   # the argument's factor vectors written out instead of their closed form
        fc.check_compiled(sim_lib, sim_glib, "vsmt_2_cube", batch=2, unfold=unfold, factor_vectors=1)
    for unfold in (0, 1, 9):   # closed form: no un-folded round / one / every round from the tables
        fc.check_compiled(sim_lib, sim_glib, "vsmt_2_cube", batch=2, unfold=unfold)


def test_host_synthesis_wires_equal_the_c_oracle_at_full_size(sim_glib):
    """bpr1cs_gadget_synthesize (host front-end alone, no device) against oracle/c's own synthesis, wire for wire, on the circuits whose
    host path the GPU suite proves: 140-round Inverse-S-box Poseidon 2:1 (partial rounds taken from the native state's VALUES,
    gadgets.hpp), the depth-32 4-ary tree circuit (n = 18 656), the 64-bit bound check, MiMC + set membership"""
    import importlib
    import subprocess
    import os
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "c")])
    from cref import COracle
    import fullsize_cases as fcs
    wl = importlib.import_module("bulletproofs-r1cs-gadgets_amd.workloads")
    o = COracle()
    cases = [wl.bound_check64(3), wl.mimc_set_membership(2), wl.poseidon_2to1_cube(bp, sim_glib, 2), wl.vsmt4(bp, sim_glib, 32, 2, 2, 0),
             wl.vsmt2(bp, sim_glib, 32, 2, b"l2", 0xffffffff, 10**6)]
    # + the Inverse 2:1 preimage circuit (a frontend_cases scenario: the oracle's values)
    gname, ip, sp, sc, cap = fc.case("poseidon_hash_2_inverse", 1)
    ob = common.oracle_batch(lambda j: fc.case("poseidon_hash_2_inverse", 1)[3], cap, 1, key="p2inv_j1")
    cases.append(dict(gadget=gname, ip=ip, sp=sp, label=ob["label"], B=1, m=ob["m"], values=ob["values"], blindings=ob["blindings"], seeds=ob["seeds"]))
    for w in cases:
        for j in range(min(2, w["B"])):
            v, bl, seed = fcs.slice_proof(w, j)
            r = o.prove_case(w["gadget"], w["ip"], w["sp"], w["label"], v, bl, seed, want_wires=True, prove=False)
            wires, n, q = bp.gadget_synthesize(w["gadget"], w["ip"], w["sp"], v, w["m"], glib=sim_glib)
            assert (n, q) == (r["n"], r["q"]), (w["gadget"], n, q, r["n"], r["q"])
            assert wires == r["wires"][:96 * n], "%s proof %d: host wires differ from the C oracle's" % (w["gadget"], j)


def test_host_synthesis_with_a_zero_sbox_input_takes_the_one_by_one_path(sim_glib):
    """The host front-end takes the S-box values of a full round, and of all partial rounds of a permutation, from ONE inversion each
    (gadgets.hpp: Montgomery's trick / the state as fractions over a common denominator).  An S-box input of 0 - an unsatisfiable
    witness, 1/0 = 0 by the reference's convention (gadget_poseidon.rs:120-125) - has no such form: those rounds fall back to one
    inversion per S-box.  Wires against the Python oracle's for a zero in a FULL round, and (140 partial rounds) for inputs without one."""
    for name in ("poseidon_hash_2_inverse_pr1_zero", "poseidon_hash_2_inverse_pr1", "poseidon_hash_2_inverse"):
        gname, ip, sp, sc, cap = fc.case(name, 0)
        ob = common.oracle_batch(lambda j: fc.case(name, 0)[3], cap, 1, satisfiable=not name.endswith("_zero"), key=name + "_synth")
        wires, n, q = bp.gadget_synthesize(gname, ip, sp, ob["values"], ob["m"], glib=sim_glib)
        assert n == ob["n"] and wires == ob["wires"], name


def test_prover_chain_ahead_and_a_commitment_after_the_synthesis_began(sim_lib, sim_glib):
    """host/r1cs.hpp ChainAhead: the C++ Prover runs the proof's TranscriptRng chain on a thread of its own from the gadget's first
    constraint-system call on and proves through bpr1cs_prove_batch_draws; a commit() after that call drops the chain and prove() takes
    bpr1cs_prove_batch_transcripts.  tests/hostsim/late_commit_check.cpp drives a Prover by hand both ways (fresh and advanced
    transcript): same proof bytes, same commitments, same transcript state afterwards as the Prover without the chain."""
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bdir = os.path.join(ROOT, "tests", "hostsim", "_build")
    exe = os.path.join(bdir, "late_commit_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DBPR1CS_HOST_ONLY", os.path.join(ROOT, "tests", "hostsim", "late_commit_check.cpp"), "-o", exe,
                           "-L" + bdir, "-lbpr1cs_gadgets_sim", "-lbpr1cs_sim", "-Wl,-rpath," + bdir, "-pthread"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
