#!/usr/bin/env python3
"""Generate tests/golden/proofs.json: deterministic proofs of the restated reference scenarios
produced by the pure-Python oracle (oracle/pyref).  The reference itself cannot run here (no Rust
toolchain, un-vendored deps) and holds no byte-level vectors, so these are REGRESSION vectors of
the oracle (pinned at the primitive level to RFC 9496 / Merlin / bulletproofs constants), not
outputs of the Rust crate.  Inputs: witness as in scenarios.py, blindings synth_scalar(b"bl<j>", i),
rng_seed = SHA-256("seed"||LE64(j))."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
import frontend_cases as fc

CASES = ["factors", "bound_check", "bound_check_64", "set_membership", "poseidon_hash_2_cube", "poseidon_hash_2_inverse",
         "poseidon_hash_4_inverse", "vsmt_2_d3", "vsmt_4_l4", "vsmt_2_cube", "vsmt_4_cube"]   # the last two: SURVEY §8f N4 (Cube-S-box trees)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "proofs.json")
out = json.load(open(OUT)) if os.path.exists(OUT) and "--all" not in sys.argv else {}   # existing vectors are kept as they are
for name in CASES:
    if name in out:
        continue
    gname, ip, sp, _, cap = fc.case(name, 0)
    ob = common.oracle_batch(lambda j: fc.case(name, j)[3], cap, 2)
    out[name] = {"gadget": gname, "iparams": ip, "sparams": [int(s).to_bytes(32, "little").hex() if not isinstance(s, bytes) else s.hex() for s in sp],
                 "capacity": cap, "label": ob["label"].decode(), "n": ob["n"], "q": ob["q"], "m": ob["m"],
                 "values": ob["values"].hex(), "blindings": ob["blindings"].hex(), "seeds": ob["seeds"].hex(),
                 "proofs": [p.hex() for p in ob["proofs"]], "commitments": [[c.hex() for c in cs] for cs in ob["comms"]],
                 "wires_sha256": hashlib.sha256(ob["wires"]).hexdigest()}
    print(name, ob["n"], ob["q"], ob["m"], len(ob["proofs"][0]))
json.dump(out, open(OUT, "w"), indent=1)
