#!/usr/bin/env python3
"""Every method the reference calls on a constraint system, a Prover, a Verifier, the generators and a transcript - name and
argument count - collected from the live (non-comment) text of /root/reference/src/*.rs.  `--write` stores the list as
tests/golden/reference_api_surface.json (data: names and counts, no source text); tests/test_rust_shim.py checks that the Rust
shim (tools/rust_shim/) and the C++ twin (host/r1cs.hpp) define every entry with that arity, and - where the reference is on
disk - that the stored list is current."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
OUT = os.path.join(ROOT, "tests", "golden", "reference_api_surface.json")

# receiver names as the reference spells them -> the type the call lands on
RECEIVERS = {"cs": "ConstraintSystem", "prover": "Prover", "verifier": "Verifier", "pc_gens": "PedersenGens", "bp_gens": "BulletproofGens",
             "prover_transcript": "Transcript", "verifier_transcript": "Transcript", "transcript": "Transcript"}
STATICS = ("Prover", "Verifier", "Transcript", "PedersenGens", "BulletproofGens", "R1CSProof")


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def call_args(text, start):
    """number of top-level arguments of the call whose '(' is at `start` (closures' `|a, b|` lists are skipped)"""
    depth, n, seen, bar = 0, 0, False, False
    for ch in text[start:]:
        if ch == "|" and depth == 1:
            bar = not bar
            seen = True
        elif bar:
            continue
        elif ch in "([{":
            depth += 1
            if depth > 1:
                seen = True
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


def collect(src_dir=REF):
    found = {}
    for fn in sorted(os.listdir(src_dir)):
        if not fn.endswith(".rs"):
            continue
        text = strip_comments(open(os.path.join(src_dir, fn)).read())
        for m in re.finditer(r"\b(\w+)\.(\w+)\s*(?:::<[^>]*>)?\(", text):
            recv, meth = m.group(1), m.group(2)
            if recv not in RECEIVERS:
                continue
            key = (RECEIVERS[recv], meth, call_args(text, m.end() - 1))
            found.setdefault(key, set()).add("%s:%d" % (fn, text.count("\n", 0, m.start()) + 1))
        for m in re.finditer(r"\b(%s)::(\w+)\s*\(" % "|".join(STATICS), text):
            key = (m.group(1), "::" + m.group(2), call_args(text, m.end() - 1))
            found.setdefault(key, set()).add("%s:%d" % (fn, text.count("\n", 0, m.start()) + 1))
    rows = []
    for (typ, meth, argc), sites in sorted(found.items()):
        sites = sorted(sites, key=lambda s: (s.split(":")[0], int(s.split(":")[1])))
        rows.append({"on": typ, "method": meth, "args": argc, "calls": len(sites), "first_sites": sites[:3]})
    return rows


def main():
    rows = collect()
    if "--write" in sys.argv:
        with open(OUT, "w") as f:
            json.dump({"source": "live text of /root/reference/src/*.rs (tools/reference_api_surface.py)", "surface": rows}, f, indent=1)
            f.write("\n")
    for r in rows:
        print("%-18s %-22s %d args  x%-3d %s" % (r["on"], r["method"], r["args"], r["calls"], " ".join(r["first_sites"])))


if __name__ == "__main__":
    main()
