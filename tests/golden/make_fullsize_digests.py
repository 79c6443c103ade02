#!/usr/bin/env python3
"""Generate tests/golden/fullsize_digests.json: SHA-256 digests of EVERY proof of the bench-size parity batches
(tests/fullsize_cases.py), produced in the build container by the C oracle (oracle/c), one worker process per core.

    python tests/golden/make_fullsize_digests.py [--cases a,b,...] [--workers 8]

Inputs come from the HOST trees: the front-end is loaded in its CPU-simulator build (tests/hostsim), no GPU is involved.
The C oracle is pinned to the pure-Python oracle (oracle/pyref) on every gadget by tests/test_oracle_c.py, and
tests/test_fullsize_digests.py re-proves samples of this fixture with both on the CPU.
The oracle is not the Rust crate: see DESIGN.md §2 ("parity unpinned") and tools/upstream_golden/ for the kit that
produces the same fixture from the real reference on a machine with cargo."""
import argparse
import hashlib
import importlib
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(HERE, "fullsize_digests.json")


def host_frontend():
    """CPU build of the library + front-end (the tests' simulator): host trees, no device"""
    import conftest
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    sim = conftest._build_hostsim("pipeline_sim.cpp", "libbpr1cs_sim.so")
    bp.load_library(sim)
    bdir = os.path.dirname(sim)
    src = os.path.join(ROOT, "bulletproofs-r1cs-gadgets_amd", "host", "frontend.cpp")
    out = os.path.join(bdir, "libbpr1cs_gadgets_sim.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(sim):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-DBPR1CS_HOST_ONLY", "-shared", "-fPIC", src, "-o", out,
                               "-L" + bdir, "-lbpr1cs_sim", "-Wl,-rpath," + bdir])
    return bp, bp.load_gadgets_library(out)


def run_case(o, name, case, workers):
    import fullsize_cases as fc
    B = case["B"]
    # warm the generators before forking (copy-on-write pages shared by the workers)
    r0 = o.prove_case(case["gadget"], case["ip"], case["sp"], case["label"], *fc.slice_proof(case, 0), prove=False)
    N = 1 << max(0, (r0["n"] - 1)).bit_length()
    o.lib.oracle_warm_gens(N)
    workers = max(1, min(workers, B))
    t0 = time.time()
    kids = []
    for w in range(workers):
        r, wr = os.pipe()
        pid = os.fork()
        if pid == 0:
            code = 1
            try:
                os.close(r)
                buf = bytearray()
                for j in range(w, B, workers):
                    res = o.prove_case(case["gadget"], case["ip"], case["sp"], case["label"], *fc.slice_proof(case, j))
                    buf += j.to_bytes(4, "little") + hashlib.sha256(res["proof"]).digest() + hashlib.sha256(b"".join(res["comms"])).digest()
                with os.fdopen(wr, "wb") as f:
                    f.write(bytes(buf))
                code = 0
            finally:
                os._exit(code)
        os.close(wr)
        kids.append((pid, r))
    pd, cd = [None] * B, [None] * B
    for pid, r in kids:
        with os.fdopen(r, "rb") as f:
            data = f.read()
        _, status = os.waitpid(pid, 0)
        assert status == 0, "worker failed"
        for k in range(0, len(data), 68):
            j = int.from_bytes(data[k:k + 4], "little")
            pd[j], cd[j] = data[k + 4:k + 36], data[k + 36:k + 68]
    assert all(x is not None for x in pd)
    dt = time.time() - t0
    return dict(B=B, m=case["m"], n=r0["n"], q=r0["q"], gadget=case["gadget"], ip=case["ip"], label=case["label"].decode(),
                inputs_sha256=fc.input_digest(case), proofs=[d.hex()[:32] for d in pd],
                commitments_sha256=hashlib.sha256(b"".join(cd)).hexdigest(),   # over the per-proof SHA-256 of each proof's m commitments
                oracle_seconds=round(dt, 1), workers=workers), dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="")
    ap.add_argument("--workers", type=int, default=os.cpu_count() or 1)
    args = ap.parse_args()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "c")])
    from cref import COracle
    import fullsize_cases as fc
    bp, glib = host_frontend()
    o = COracle()
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    names = [c for c in args.cases.split(",") if c] or list(fc.CASES)
    for name in names:
        t0 = time.time()
        case = fc.CASES[name](bp, glib)
        print("%s: inputs built in %.1f s (B = %d, m = %d)" % (name, time.time() - t0, case["B"], case["m"]), flush=True)
        out[name], dt = run_case(o, name, case, args.workers)
        print("%s: %d proofs in %.1f s on %d workers" % (name, case["B"], dt, out[name]["workers"]), flush=True)
        json.dump(out, open(OUT, "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
