#!/usr/bin/env python3
"""Proofs per call against ms per proof for the compiled depth-32 tree circuit (bpr1cs_gadget_compile once, then ONE bpr1cs_prove_batch
call per batch size on a handle with no options): the curve between the single proof and the benchmarked batch.
python tools/batch_sweep.py [--sizes 1,8,32,...]"""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
wl = importlib.import_module("bulletproofs-r1cs-gadgets_amd.workloads")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1,8,32,64,128,256,512,1024,2048,4096")
    args = ap.parse_args()
    lib = bp.load_library()
    bp.load_gadgets_library()
    sizes = [int(x) for x in args.sizes.split(",")]
    w = wl.vsmt4(bp, None, 32, max(sizes), max(sizes), 0)
    gens = bp.Gens(32768)
    circ = bp.CompiledGadget(w["gadget"], w["ip"], w["sp"])
    m = w["m"]
    print("# gadget_vsmt_4 depth 32, compiled circuit, one bpr1cs_prove_batch call per line (median of the 2nd and 3rd call)")
    for B in sizes:
        v, b, s = w["values"][:B * m * 32], w["blindings"][:B * m * 32], w["seeds"][:B * 32]
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            bp.prove_batch(gens, circ, w["label"], v, b, s, B)
            ts.append(time.perf_counter() - t0)
        st = bp.last_prove_stats(lib)
        t = sum(ts[1:]) / 2
        print("B=%5d  %8.1f ms per call  %8.3f ms per proof  %7.0f proofs/s   device phases (total, inputs, rng||witness, commit sums, polynomials, argument) %s"
              % (B, 1e3 * t, 1e3 * t / B, B / t, [round(x, 1) for x in st["phase_ms"]]), flush=True)


if __name__ == "__main__":
    main()
