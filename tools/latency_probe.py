#!/usr/bin/env python3
"""Latency of the reference's own call shape - ONE proof per prove() (src/gadget_vsmt_4.rs:421-435, gadget_bound_check.rs:49-87) -
and of small host-wire batches, with the seconds per stage.  python tools/latency_probe.py [--cases c1,c4,d128,d253] [--batches 1,8,64]"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
wl = importlib.import_module("bulletproofs-r1cs-gadgets_amd.workloads")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="c1,c4")
    ap.add_argument("--batches", default="1,8,64")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--no-device-program", action="store_true")
    args = ap.parse_args()
    lib = bp.load_library()
    bp.load_gadgets_library()
    opts = {k: int(v) for k, v in (kv.split("=") for kv in args.opt)}
    out = {}
    for case in args.cases.split(","):
        nb = max(int(x) for x in args.batches.split(","))
        if case == "c1":
            w, cap = wl.bound_check64(64), 128
        elif case == "d128":   # the reference's literal tests (src/gadget_vsmt_4.rs:25, gadget_vsmt_2.rs:23)
            w, cap = wl.vsmt4(bp, None, 128, nb, nb, 11), 131072
        elif case == "d253":
            w, cap = wl.vsmt2(bp, None, 253, nb, b"l253", (1 << 250) - 1, 2 * 10**6), 262144
        else:
            w, cap = wl.vsmt4(bp, None, 32, 64, 64, 0), 32768
        t0 = time.time()
        gens = bp.Gens(cap, **opts)
        print("%s: generators in %.1f s" % (case, time.time() - t0), flush=True)
        m = w["m"]
        for B in [int(x) for x in args.batches.split(",")]:
            rows = []
            for rep in range(args.reps):
                v, b, s = w["values"][:B * m * 32], w["blindings"][:B * m * 32], w["seeds"][:B * 32]
                t0 = time.perf_counter()
                P, C, sec = bp.gadget_prove_on(gens, w["gadget"], w["ip"], w["sp"], w["label"], v, b, m, B, s)
                wall = time.perf_counter() - t0
                st = bp.last_prove_stats(lib)
                rows.append(dict(wall_ms=1e3 * wall, stage_ms={k: 1e3 * x for k, x in sec.items()}, phase_ms=st["phase_ms"], msm_ms=st["msm_ms"], msm_launches=st["msm_launches"]))
                print(case, "B=%d rep %d: %.1f ms  (%.2f ms/proof)  stages %s  phases %s msm %.1f ms / %d" % (
                    B, rep, 1e3 * wall, 1e3 * wall / B, {k: round(1e3 * x, 1) for k, x in sec.items()}, [round(x, 1) for x in st["phase_ms"]], st["msm_ms"], st["msm_launches"]), flush=True)
            out["%s_b%d" % (case, B)] = rows
            if B == 1:   # the verifier half of the reference's tests: one proof per verify()
                for rep in range(args.reps):
                    t0 = time.perf_counter()
                    ok, vsec = bp.gadget_verify_on(gens, w["gadget"], w["ip"], w["sp"], w["label"], P[0], C[0])
                    print(case, "verify rep %d: %.1f ms  ok=%s  stages %s" % (rep, 1e3 * (time.perf_counter() - t0), ok, {k: round(1e3 * x, 1) for k, x in vsec.items()}), flush=True)
        # the device-program path at the same batch sizes (no host synthesis), for comparison
        circ = bp.CompiledGadget(w["gadget"], w["ip"], w["sp"])
        for B in ([] if args.no_device_program else [int(x) for x in args.batches.split(",")]):
            for rep in range(2):
                v, b, s = w["values"][:B * m * 32], w["blindings"][:B * m * 32], w["seeds"][:B * 32]
                t0 = time.perf_counter()
                bp.prove_batch(gens, circ, w["label"], v, b, s, B)
                wall = time.perf_counter() - t0
                st = bp.last_prove_stats(lib)
                print(case, "device program B=%d rep %d: %.1f ms  phases %s msm %.1f ms / %d" % (B, rep, 1e3 * wall, [round(x, 1) for x in st["phase_ms"]], st["msm_ms"], st["msm_launches"]), flush=True)
        circ.close()
        gens.close()
        bp.release_cached_memory(lib)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
