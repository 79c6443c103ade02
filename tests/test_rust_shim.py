"""tools/rust_shim/: the Rust side of the boundary as (uncompiled) text.  What CAN be checked without cargo: ffi.rs is the
generator's output for the current header, and every `ffi::bpr1cs_*` call in the hand-written files names a function the header
declares and passes as many arguments as it takes."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tools", "rust_shim")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _call_args(text, start):
    """number of top-level arguments of the call whose '(' is at `start`"""
    depth, n, seen = 0, 0, False
    for ch in text[start:]:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                return n + (1 if seen else 0)
        elif ch == "," and depth == 1:
            n += 1
            seen = False
        elif depth >= 1 and not ch.isspace():
            seen = True
    raise AssertionError("unbalanced call")


def test_ffi_rs_is_generated_and_shim_calls_match_the_header():
    import gen_rust_bindings as g
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_bindings.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    _, structs, consts, funcs = g.parse(open(g.HEADER).read())
    arity = {name: len(params) for name, params, _ in funcs}
    const_names = {k for k, _ in consts}
    used = set()
    for fn in ("transcript.rs", "generators.rs", "prover.rs", "verifier.rs"):
        text = open(os.path.join(SHIM, fn)).read()
        for m in re.finditer(r"ffi::(bpr1cs_\w+)\s*\(", text):
            name = m.group(1)
            assert name in arity, "%s calls %s, which include/bpr1cs.h does not declare" % (fn, name)
            assert _call_args(text, m.end() - 1) == arity[name], "%s: %s takes %d arguments" % (fn, name, arity[name])
            used.add(name)
        for m in re.finditer(r"ffi::(BPR1CS_\w+)", text):
            assert m.group(1) in const_names, m.group(1)
        # the circuit description literal names every field of the C struct, in order
        for lit in re.findall(r"ffi::bpr1cs_circuit_desc \{(.*?)\};", text, flags=re.S):
            fields = re.findall(r"(?<![:\w])(\w+):(?!:)", lit)
            assert fields == [f for f, _ in dict(structs)["bpr1cs_circuit_desc"]], fn
    # the entry points INTEGRATION.md §2 routes the reference's calls to are the ones the shim uses
    assert {"bpr1cs_gens_create", "bpr1cs_gens_point", "bpr1cs_msm_fixed", "bpr1cs_circuit_create", "bpr1cs_prove_batch_transcripts",
            "bpr1cs_verify_batch", "bpr1cs_transcript_new", "bpr1cs_transcript_append_message", "bpr1cs_transcript_challenge_bytes"} <= used
