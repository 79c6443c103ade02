#!/usr/bin/env python3
"""The plain C caller of the boundary (tests/c_caller/prove_c4.c) with NOTHING else on the device: the inputs are written by a
short-lived Python process that has exited before the C program creates its handle, so the library sizes its jobs from the whole
device - what a service written in C gets (tests/test_gpu_c_caller.py runs the same program under a pytest process that holds
device memory of its own; bpr1cs_prove_stats.sizing_* reports what the handle saw either way).
    python tools/c_caller_standalone.py [warm-up proofs] [timed proofs]   -> one JSON line (also gpurun_out/c_caller_standalone.json)"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = r'''
import importlib, struct, sys
sys.path.insert(0, %r)
bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
wl = importlib.import_module("bulletproofs-r1cs-gadgets_amd.workloads")
bp.load_library(); glib = bp.load_gadgets_library()
have, levels = 2048, 32
w = wl.vsmt4(bp, glib, levels, have, have, 0)
open(sys.argv[1], "wb").write(struct.pack("<III", levels, w["m"], have) + w["sp"][0] + w["values"] + w["blindings"] + w["seeds"])
print(bp.POSEIDON_PARAMS_PATH)
'''


def main():
    warm = sys.argv[1] if len(sys.argv) > 1 else "8192"
    timed = sys.argv[2] if len(sys.argv) > 2 else "20480"
    with tempfile.TemporaryDirectory() as d:
        inp, exe, out = os.path.join(d, "inputs.bin"), os.path.join(d, "prove_c4"), os.path.join(d, "proofs.bin")
        params = subprocess.run([sys.executable, "-c", GEN % ROOT, inp], capture_output=True, text=True, check=True).stdout.strip().split("\n")[-1]
        csrc = os.path.join(ROOT, "bulletproofs-r1cs-gadgets_amd", "csrc")
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_caller", "prove_c4.c"), "-o", exe,
                               "-L" + csrc, "-lbpr1cs_gadgets", "-lbpr1cs_hip", "-Wl,-rpath," + csrc])
        r = subprocess.run([exe, inp, params, warm, timed, out], capture_output=True, text=True, timeout=1200)
        if r.returncode:
            sys.exit(r.stdout + r.stderr)
        line = [l for l in r.stdout.split("\n") if l.startswith("{")][-1]
    res = json.loads(line)
    res["note"] = "tests/c_caller/prove_c4.c alone on the device: ONE bpr1cs_prove_batch call over %s proofs on a handle created with no options" % timed
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "c_caller_standalone.json"), "w").write(json.dumps(res) + "\n")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
