"""The kit that pins the oracle to the real Rust stack (tools/upstream_golden/): its committed inputs are current, and -
whenever somebody has run it with cargo and dropped the result at tests/golden/upstream_proofs.txt - every proof and
commitment the reference produced equals this repository's vectors byte for byte.  Without that file the comparison is
SKIPPED and byte-level parity with the Rust crate stays unpinned."""
import hashlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIT = os.path.join(ROOT, "tools", "upstream_golden")
UP = os.path.join(ROOT, "tests", "golden", "upstream_proofs.txt")


def test_kit_inputs_are_current(tmp_path):
    """upstream_inputs.txt is what export_inputs.py writes from the committed vectors and the full-size case builders"""
    sys.path.insert(0, KIT)
    import export_inputs
    out = tmp_path / "inputs.txt"
    export_inputs.main(str(out))
    assert out.read_text() == open(os.path.join(KIT, "upstream_inputs.txt")).read()


def test_kit_files_present_and_name_the_reference_functions():
    src = open(os.path.join(KIT, "upstream_golden.rs")).read()
    for fn in ("gen_proof_of_bounded_num", "vanilla_merkle_merkle_tree_4_verif_gadget", "vanilla_merkle_merkle_tree_verif_gadget",
               "Poseidon_hash_2_gadget", "Poseidon_hash_4_gadget", "mimc_gadget", "allocate_statics_for_prover", "BPR1CS_FIXED_RNG_HEX"):
        assert fn in src
    assert "BPR1CS_FIXED_RNG_HEX" in open(os.path.join(KIT, "bulletproofs_fixed_rng.patch")).read()


def test_upstream_vectors_equal_ours():
    if not os.path.exists(UP):
        pytest.skip("tests/golden/upstream_proofs.txt absent: nobody has run tools/upstream_golden with cargo yet - "
                    "byte-level parity with the Rust crate is UNPINNED")
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "proofs.json")))
    full = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")))
    seen = 0
    for line in open(UP):
        t = line.split()
        if not t:
            continue
        kind, case, j = t[0], t[1], int(t[2])
        if kind == "proof":
            proof = bytes.fromhex(t[3])
            if case in gold:
                assert proof.hex() == gold[case]["proofs"][j], "%s proof %d: the Rust crate and the oracle disagree" % (case, j)
            else:
                assert hashlib.sha256(proof).hexdigest()[:32] == full[case]["proofs"][j], "%s proof %d: the Rust crate and the oracle disagree" % (case, j)
            seen += 1
        elif kind == "comms" and case in gold:
            ours = gold[case]["commitments"][j]
            assert t[3:3 + len(ours)] == ours, "%s commitments %d differ" % (case, j)
    assert seen > 0


def test_first_divergence_probes_in_the_readme_are_current():
    """tools/upstream_golden/README.md carries the oracle's values of the first-divergence probes (generators, first TranscriptRng
    draw, A_I1, challenges ...): they must be what tools/upstream_golden/print_probes.py prints now"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "upstream_golden", "print_probes.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    readme = open(os.path.join(root, "tools", "upstream_golden", "README.md")).read()
    for line in out.stdout.strip().split("\n"):
        name, value = line.split()[0], line.split()[1]
        if name == "proof":
            continue
        assert value in readme, name
