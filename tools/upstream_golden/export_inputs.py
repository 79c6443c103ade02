#!/usr/bin/env python3
"""Write tools/upstream_golden/upstream_inputs.txt: the inputs (committed values, V-blindings, rng seed) of the golden
vectors of tests/golden/proofs.json and of a few full-size cases (tests/fullsize_cases.py), in the line format
upstream_golden.rs reads.  Run in the build container (host trees through the CPU simulator build of the front-end)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

FULLSIZE = {"c1_bound_check64_x4096": [0, 4095], "vsmt4_d128_x70": [0, 69], "vsmt2_d253_x66": [0], "c5_mimc_set_x8192": [0, 8191], "c2_poseidon2_cube_x4096": [0]}


def hx(b):
    return b.hex() if isinstance(b, (bytes, bytearray)) else int(b).to_bytes(32, "little").hex()


def record(out, case, j, gadget, label, ip, sp, values, blindings, seed):
    m = len(values) // 32
    out.append("proof %s %d" % (case, j))
    out.append("gadget %s" % gadget)
    out.append("label %s" % label.hex())
    out.append("ip %d %s" % (len(ip), " ".join(str(x) for x in ip)))
    out.append("sp %d %s" % (len(sp), " ".join(hx(s) for s in sp)))
    out.append("values %d %s" % (m, " ".join(values[32 * i:32 * i + 32].hex() for i in range(m))))
    out.append("blindings %d %s" % (m, " ".join(blindings[32 * i:32 * i + 32].hex() for i in range(m))))
    out.append("seed %s" % seed.hex())
    out.append("end")


def main(path=None):
    out = []
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "proofs.json")))
    for name, gd in gold.items():
        m = gd["m"]
        vals, bls, seeds = bytes.fromhex(gd["values"]), bytes.fromhex(gd["blindings"]), bytes.fromhex(gd["seeds"])
        for j in range(len(gd["proofs"])):
            sp = [bytes.fromhex(s) for s in gd["sparams"]]
            if j > 0 and gd["gadget"].startswith("poseidon"):
                continue   # the public hash output differs per proof and only proof 0's is recorded as `sparams`
            record(out, name, j, gd["gadget"], gd["label"].encode(), gd["iparams"], sp, vals[j * m * 32:(j + 1) * m * 32],
                   bls[j * m * 32:(j + 1) * m * 32], seeds[32 * j:32 * j + 32])
    import make_fullsize_digests as mk
    import fullsize_cases as fc
    bp, glib = mk.host_frontend()
    for name, picks in FULLSIZE.items():
        case = fc.CASES[name](bp, glib)
        for j in picks:
            v, b, s = fc.slice_proof(case, j)
            record(out, name, j, case["gadget"], case["label"], case["ip"], case["sp"], v, b, s)
    with open(path or os.path.join(HERE, "upstream_inputs.txt"), "w") as f:
        f.write("\n".join(out) + "\n")
    print("%d records" % sum(1 for x in out if x == "end"))


if __name__ == "__main__":
    main()
