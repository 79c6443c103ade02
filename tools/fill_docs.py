#!/usr/bin/env python3
"""Fills the @@PLACEHOLDER@@ fields of DESIGN.md / INTEGRATION.md / README.md from profiles/<tag>_bench_default.json and the tag's other
summaries, so that the figures quoted in the documents are the committed ones.  The four measured blocks of the documents live as templates
under tools/doc_templates/ (a document carries the marker @@NUMBERS@@ / @@ROUND6@@ / @@README_STATUS@@ / @@INTEGRATION_2A@@ where the block
goes; to refresh a filled document, cut the block back to its marker).   python tools/fill_docs.py r06h [files...]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
files = sys.argv[2:] or ["DESIGN.md", "INTEGRATION.md", "README.md"]
P = lambda name: os.path.join(ROOT, "profiles", "%s_%s" % (tag, name))
d = json.load(open(P("bench_default.json")))
lat, cfg, ro, rv, cb = d["latency"], d["configs"], d["roofline"], d["roofline_valu"], d["cpu_baseline"]


def f1(x): return "%.1f" % x
def f2(x): return "%.2f" % x
def k(x): return "%.1f k" % (x / 1e3) if x >= 20000 else "%.0f" % x


V = {"TAG": tag}
V["WALL"] = re.search(r"(\d+)", open(P("bench_default_wall.txt")).read()).group(1)
V["HEAD"] = "%.0f" % d["value"]; V["MSSTEP"] = f1(d["ms_per_step"])
V["LAUNCH"] = f1(ro["avg_launch_ms"])
ks = open(P("kernel_stats.txt")).read()
m = re.search(r"k_msm_fixed2 \[\d+ full-size jobs.*\| ([\d.]+)\s*$", ks, re.M)
V["LAUNCH_PROF"] = f1(float(m.group(1))) if m else "n/a"
V["ACH"] = f1(ro["achieved"]); V["FRAC"] = "%.4f" % ro["frac"]
V["TRAFFIC"] = f1(ro["traffic"] / 1e9) if ro.get("traffic") else "n/a"; V["TRATIO"] = f1(ro["traffic"] / ro["alg_bytes_per_launch"]) if ro.get("traffic") else "n/a"
V["GADDS"] = f1(rv["achieved"]); V["VFRAC"] = f2(rv["frac"]); V["VNOM"] = f2(rv["frac_nominal"]); V["VCHAIN"] = f2(rv["frac_of_madd_chain"])
V["CLK"] = f2(rv["effective_clock_ghz"]) if rv.get("effective_clock_ghz") else "n/a"
clk = open(P("pmc_clock.txt")).read()
m = re.search(r"k_probe_mad: ([\d.]+) GHz", clk)
V["CLKP"] = m.group(1) if m else "n/a"
V["CPU1"] = "%.3f" % cb["single_thread"]["value"]; V["CPUN"] = f1(cb["value"])
for key, name in (("C1", "c1"), ("C2", "c2"), ("C3", "c3"), ("C5", "c5"), ("D128T", "vsmt4_d128"), ("D253T", "vsmt2_d253")):
    V[key] = k(cfg[name]["value"])
    g = cfg[name].get("msm_table_adds_per_s")
    V[{"D128T": "D128G", "D253T": "D253G"}.get(key, key + "G")] = f1(g / 1e9) if g else "n/a"
c4, c1, d128, d253 = lat["c4"], lat["c1"], lat["d128"], lat["d253"]


def row(prefix, r):
    b = r["b1"]; s = b["stage_ms"]
    V[prefix + "COM"] = f2(s["commit"]); V[prefix + "GAD"] = f1(s["gadget"]) if s["gadget"] >= 1 else f2(s["gadget"])
    V[prefix + "CIR"] = f2(s["circuit"]); V[prefix + "PRV"] = f1(s["prove"]) if s["prove"] >= 10 else f2(s["prove"])
    V[prefix + "FIRST"] = f1(b["first_call_ms"])
    ph = b["device_phase_ms"]
    V[prefix + "IPA"] = f1(ph["ipa"]) if ph["ipa"] >= 3 else f2(ph["ipa"]); V[prefix + "DEV"] = f1(ph["total"]) if ph["total"] >= 3 else f2(ph["total"])
    V[prefix + "MID"] = f1(ph["commit_msm"] + ph["poly"]); V[prefix + "FRONT"] = f1(ph["inputs+commitV"] + ph["rng||witness"])
    v = r["verify_b1"]; vs = v["stage_ms"]
    return b, v, vs


b, v, vs = row("C4", c4)
V["C4B1"] = f1(b["ms_per_call"]); V["VC4"] = f1(v["ms_per_call"]); V["VC4G"] = f1(vs["gadget"]); V["VC4D"] = f1(vs["verify"])
e = c4["b1_eager_commits"]
V["C4B1E"] = f1(e["ms_per_call"]); V["C4ECOM"] = f1(e["stage_ms"]["commit"]); V["US_COMMIT"] = "%.0f" % e["us_per_commit"]
V["C4CPU"] = "%.0f" % c4["cpu_port_ms_per_proof"]
V["C1B1E"] = f2(c1["b1_eager_commits"]["ms_per_call"])
ul = open(P("ubench_latency.txt")).read()
m = re.search(r"ge_compress\s+([\d.]+) ns per op\s+([\d.]+) clock64", ul)
V["COMPRESS_US"] = "%.0f" % (float(m.group(2)) / 2390.0) if m else "n/a"
for B in (8, 64):
    r = c4["b%d" % B]; s = r["stage_ms"]
    V["C4B%d" % B] = f1(r["ms_per_call"]); V["C4B%dPP" % B] = f2(r["ms_per_proof"]); V["C4B%dGAD" % B] = f1(s["gadget"])
    V["C4B%dCIR" % B] = f2(s["circuit"]); V["C4B%dPRV" % B] = f1(s["prove"])
b, v, vs = row("C1", c1)
V["C1B1"] = f2(b["ms_per_call"]); V["VC1"] = f2(v["ms_per_call"]); V["VC1G"] = f2(vs["gadget"]); V["VC1D"] = f2(vs["verify"])
V["C1B8"] = f2(c1["b8"]["ms_per_call"]); V["C1B64"] = f2(c1["b64"]["ms_per_call"]); V["C1CPU"] = f1(c1["cpu_port_ms_per_proof"])
b, v, vs = row("D128", d128)
V["D128"] = f1(b["ms_per_call"]); V["VD128"] = f1(v["ms_per_call"]); V["VD128G"] = f1(vs["gadget"]); V["VD128D"] = f1(vs["verify"])
V["D128CPU"] = f1(d128["cpu_port_ms_per_proof"] / 1e3); V["D128CPUMS"] = "%.0f" % d128["cpu_port_ms_per_proof"]
b, v, vs = row("D253", d253)
V["D253"] = f1(b["ms_per_call"]); V["VD253"] = f1(v["ms_per_call"]); V["VD253G"] = f1(vs["gadget"]); V["VD253D"] = f1(vs["verify"])
V["D253CPU"] = f1(d253["cpu_port_ms_per_proof"] / 1e3); V["D253CPUMS"] = "%.0f" % d253["cpu_port_ms_per_proof"]
V["VB"] = k(d["verify"]["proofs_per_s"]); V["VBC"] = k(d["verify_batched"]["proofs_per_s"]); V["VBS"] = k(d["verify_batched"]["split_shared_base"]["proofs_per_s"])
V["CC"] = "%.0f" % json.load(open(P("c_caller_standalone.json")))["proofs_per_s"]
# INTEGRATION's names
V.update(COMMIT_LAZY=V["C4COM"], COMMIT_EAGER=V["C4ECOM"], GADGET=V["C4GAD"], CIRCUIT=V["C4CIR"], PROVE=V["C4PRV"], TOTAL=V["C4B1"], TOTAL_EAGER=V["C4B1E"],
         FIRST=V["C4FIRST"], B8=V["C4B8"], B64=V["C4B64"], VERIFY=V["VC4"], VGADGET=V["VC4G"], VDEV=V["VC4D"], VERIFY_C1=V["VC1"])
for fn in files:
    path = os.path.join(ROOT, fn)
    s = open(path).read()
    T = lambda n: os.path.join(ROOT, "tools", "doc_templates", n)
    for blk, src in (("@@NUMBERS@@", T("design_numbers.md")), ("@@ROUND6@@", T("design_round6.md")), ("@@README_STATUS@@", T("readme_status.md")),
                     ("@@INTEGRATION_2A@@", T("integration_2a.md"))):
        if blk in s and os.path.exists(src):
            s = s.replace(blk, open(src).read().rstrip("\n"))
    missing = sorted(set(re.findall(r"@@([A-Z0-9_]+)@@", s)) - set(V))
    if missing:
        print(fn, "unknown placeholders:", missing)
    s = re.sub(r"@@([A-Z0-9_]+)@@", lambda m: V.get(m.group(1), m.group(0)), s)
    open(path, "w").write(s)
    print("filled", fn)
