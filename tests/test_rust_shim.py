"""tools/rust_shim/: the Rust side of the boundary as (uncompiled) text.  What CAN be checked without cargo: ffi.rs is the
generator's output for the current header, and every `ffi::bpr1cs_*` call in the hand-written files names a function the header
declares and passes as many arguments as it takes - and (round 5) that the shim DEFINES every method the reference calls on a
constraint system / Prover / Verifier / generators / transcript, with the reference's argument count and, for the fork's own
additions, the reference's return shape (tests/golden/reference_api_surface.json, made by tools/reference_api_surface.py)."""
import json
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


# ---- the surface the reference's gadget files need (the shim had `allocate` where src/gadget_poseidon.rs:165 calls
# `allocate_single`, and no num_constraints / num_multipliers)
SURFACE = os.path.join(ROOT, "tests", "golden", "reference_api_surface.json")
SHIM_FILE = {"Prover": "prover.rs", "Verifier": "verifier.rs", "PedersenGens": "generators.rs", "BulletproofGens": "generators.rs",
             "Transcript": "transcript.rs"}
# return types the reference's own use fixes: `let (var_l, _) = cs.allocate_single(val_l)?; ... var_o.unwrap()` (gadget_poseidon.rs:165-184),
# `let (com, var) = prover.commit(..)`, `cs.evaluate_lc(..)` mapped as an Option (gadget_poseidon.rs:160-163),
# `let (_, _, o) = cs.multiply(..)`, `cs.allocate_multiplier(..)?` destructured into three (gadget_vsmt_4.rs:226)
RETURNS = {
    ("ConstraintSystem", "allocate_single"): "Result<(Variable, Option<Variable>), R1CSError>",
    ("ConstraintSystem", "allocate_multiplier"): "Result<(Variable, Variable, Variable), R1CSError>",
    ("ConstraintSystem", "multiply"): "(Variable, Variable, Variable)",
    ("ConstraintSystem", "evaluate_lc"): "Option<Scalar>",
    ("Prover", "commit"): "(CompressedRistretto, Variable)",
    ("Prover", "prove"): "Result<R1CSProof, R1CSError>",
    ("Prover", "num_constraints"): "usize",
    ("Prover", "num_multipliers"): "usize",
    ("Verifier", "commit"): "Variable",
    ("Verifier", "verify"): "Result<(), R1CSError>",
}


def _rust_fns(text):
    """{name: [(n_params_without_self, has_self, return_type)]} of every `fn` in a Rust source text"""
    out = {}
    for m in re.finditer(r"\bfn\s+(\w+)\s*(?:<[^>]*>)?\(", text):
        start = m.end() - 1
        depth, end = 0, None
        for k in range(start, len(text)):
            if text[k] in "([{":
                depth += 1
            elif text[k] in ")]}":
                depth -= 1
                if depth == 0:
                    end = k
                    break
        params = text[start + 1:end]
        parts, d, cur = [], 0, ""
        for ch in params:
            if ch in "([{<":
                d += 1
            elif ch in ")]}>":
                d -= 1
            if ch == "," and d == 0:
                parts.append(cur)
                cur = ""
            else:
                cur += ch
        if cur.strip():
            parts.append(cur)
        has_self = bool(parts) and re.fullmatch(r"\s*&?\s*(mut\s+)?self\s*", parts[0]) is not None
        rest = text[end + 1:text.index("{", end)]
        ret = rest.split("->", 1)[1].strip() if "->" in rest else "()"
        out.setdefault(m.group(1), []).append((len(parts) - (1 if has_self else 0), has_self, re.sub(r"\s+", " ", ret)))
    return out


def _trait_impl(text, trait):
    m = re.search(r"impl<[^>]*>\s+%s\s+for\s+\w+<[^>]*>\s*\{" % trait, text)
    assert m, "no `impl %s for ..` block" % trait
    depth = 0
    for k in range(m.end() - 1, len(text)):
        if text[k] == "{":
            depth += 1
        elif text[k] == "}":
            depth -= 1
            if depth == 0:
                return text[m.end():k]
    raise AssertionError("unbalanced impl block")


def test_shim_defines_every_method_the_reference_calls():
    surface = json.load(open(SURFACE))["surface"]
    assert len(surface) >= 17
    texts = {fn: open(os.path.join(SHIM, fn)).read() for fn in set(SHIM_FILE.values())}
    twin = open(os.path.join(ROOT, "bulletproofs-r1cs-gadgets_amd", "host", "r1cs.hpp")).read()
    checked = 0
    for row in surface:
        typ, meth, argc = row["on"], row["method"], row["args"]
        where = "%s.%s/%d (reference: %s)" % (typ, meth, argc, ", ".join(row["first_sites"]))
        if typ == "ConstraintSystem":
            # a trait method: BOTH implementations carry it, same arity, same return type
            for fn in ("prover.rs", "verifier.rs"):
                fns = _rust_fns(_trait_impl(texts[fn], "ConstraintSystem"))
                assert meth in fns, "%s: %s does not implement it" % (where, fn)
                n, has_self, ret = fns[meth][0]
                assert has_self and n == argc, "%s: %s takes %d" % (where, fn, n)
                if (typ, meth) in RETURNS:
                    assert ret == RETURNS[(typ, meth)], "%s: %s returns `%s`" % (where, fn, ret)
            assert re.search(r"\b%s\s*\(" % meth, twin), "%s: missing in host/r1cs.hpp" % where
        else:
            name = meth[2:] if meth.startswith("::") else meth
            fns = _rust_fns(texts[SHIM_FILE[typ]])
            assert name in fns, "%s: %s does not define it" % (where, SHIM_FILE[typ])
            ok = [f for f in fns[name] if f[0] == argc and f[1] == (not meth.startswith("::"))]
            assert ok, "%s: %s has %s" % (where, SHIM_FILE[typ], fns[name])
            if (typ, meth) in RETURNS:
                assert any(f[2] == RETURNS[(typ, meth)] for f in ok), "%s: returns %s" % (where, [f[2] for f in ok])
            if typ in ("Prover", "Verifier"):
                assert re.search(r"\b%s\s*\(" % name, twin) or name == "new", "%s: missing in host/r1cs.hpp" % where
        checked += 1
    assert checked == len(surface)
    # the pairing of the fork's allocate_single (trap T8): first call (left, None), second (right, Some(output of the SAME multiplier))
    for fn in ("prover.rs", "verifier.rs"):
        body = _trait_impl(texts[fn], "ConstraintSystem")
        body = body[body.index("fn allocate_single"):body.index("fn allocate_multiplier")]
        assert "Variable::MultiplierLeft(i), None" in body and "Variable::MultiplierRight(i), Some(Variable::MultiplierOutput(i))" in body, fn


def test_reference_api_surface_is_current():
    import pytest
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("the reference is not on this machine (the committed list is what the shim is checked against)")
    import reference_api_surface as ras
    assert ras.collect() == json.load(open(SURFACE))["surface"], "run tools/reference_api_surface.py --write"
