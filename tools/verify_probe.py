#!/usr/bin/env python3
"""One proof per verify() - the verifier half of the reference's tests (src/gadget_vsmt_4.rs:442-479) - a few calls in a row, for a
kernel trace (tools/trace_lastcall.py DIR gap 3000 K_verify_finish).  python tools/verify_probe.py [c4|c1] [reps]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
wl = importlib.import_module("bulletproofs-r1cs-gadgets_amd.workloads")
case = sys.argv[1] if len(sys.argv) > 1 else "c4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
bp.load_library(); bp.load_gadgets_library()
w, cap = (wl.bound_check64(4), 128) if case == "c1" else (wl.vsmt4(bp, None, 32, 4, 4, 0), 32768)
gens = bp.Gens(cap)
m = w["m"]
P, C, _ = bp.gadget_prove_on(gens, w["gadget"], w["ip"], w["sp"], w["label"], w["values"][:m * 32], w["blindings"][:m * 32], m, 1, w["seeds"][:32])
for rep in range(reps):
    time.sleep(0.05)
    t0 = time.perf_counter()
    ok, sec = bp.gadget_verify_on(gens, w["gadget"], w["ip"], w["sp"], w["label"], P[0], C[0])
    print(case, "verify rep %d: %.2f ms ok=%s %s" % (rep, 1e3 * (time.perf_counter() - t0), ok, {k: round(1e3 * x, 2) for k, x in sec.items()}), flush=True)
