#!/usr/bin/env python3
"""Per-batch breakdown of the shared "back" stream from a rocprofv3 --kernel-trace database of `bench.py` (two batches in
flight): steady-state period between consecutive K_assemble launches, how busy the stream is, and the time every kernel
takes per batch on it; plus the front kernels (TranscriptRng chain, witness synthesis) and the MSM launches of one batch,
each with the share of its duration a front kernel of the NEXT batch was running.
  rocprofv3 --kernel-trace -d DIR -o out -- python bench.py --steps 6 --warmup 1 --cpu-proofs 0 ; trace_perjob.py DIR [rows]"""
import sqlite3, sys, glob, collections
db = sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1]
con = sqlite3.connect(db)
kt = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')") if r[0].startswith("kernels")][0]
def short(n):
    n = n.split("(")[0]
    for k in ("k_rng_stream", "k_witness_team", "k_msm_fixed2", "k_poseidon_team", "k_probe_madd", "k_probe_mad"):
        if k in n: return k
    if "k_functor_wave<" in n: return n.split("k_functor_wave<")[1].split(">")[0]
    if "k_functor<" in n: return n.split("k_functor<")[1].split(">")[0]
    return n[:40]
rows = [(short(n), s, e, q) for n, s, e, q in con.execute("select name, start, end, queue_id from %s order by start" % kt)]
msm = [r for r in rows if r[0] == "k_msm_fixed2"]
hq = collections.Counter(r[3] for r in msm).most_common(1)[0][0]
on = [r for r in rows if r[3] == hq]
# job boundaries on the heavy stream: K_transcript_A runs once per device job, right after its commitment sums (K_assemble is on
# the job's tail stream when the last IPA rounds are handed off, and its completion time says nothing about the heavy stream)
asm = [r for r in on if r[0] == "K_transcript_A"]
if len(asm) < 4:
    sys.exit("need at least 4 device jobs in the trace")
lo, hi, nb = asm[1][1], asm[-2][1], len(asm) - 3
w = [r for r in on if r[1] >= lo and r[2] <= hi]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in w:
    a = agg[r[0]]; a[0] += 1; a[1] += (r[2] - r[1]) / 1e6
tot = sum(a[1] for a in agg.values())
print("# steady state over %d device jobs: period %.1f ms per job, back stream busy %.1f ms per job (%.1f %%)" % (nb, (hi - lo) / 1e6 / nb, tot / nb, 100 * tot / ((hi - lo) / 1e6)))
print("# kernel | launches per job | ms per job | share of the back stream")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print("%-28s | %5.1f | %8.2f | %5.1f %%" % (k, a[0] / nb, a[1] / nb, 100 * a[1] / tot))
# idle time of the heavy stream inside the window, by the pair of launches it lies between
gaps = collections.defaultdict(lambda: [0, 0.0])
ws = sorted(w, key=lambda r: r[1])
for r0, r1 in zip(ws, ws[1:]):
    gp = (r1[1] - r0[2]) / 1e6
    if gp > 0.004:
        a = gaps[r0[0] + " -> " + r1[0]]; a[0] += 1; a[1] += gp
print("# idle time of the back stream (gaps > 4 us), per job, by neighbour pair: count ms   [sum of all gaps %.2f ms per job]" % (sum(max(0, r1[1] - r0[2]) for r0, r1 in zip(ws, ws[1:])) / 1e6 / nb))
for k, a in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:10]:
    print("  %-58s %5.1f %8.2f" % (k, a[0] / nb, a[1] / nb))
for r0, r1 in zip(ws, ws[1:]):
    gp = (r1[1] - r0[2]) / 1e6
    if gp > 0.5:
        ji = max(i for i, a in enumerate(asm) if a[1] <= r1[1] + 200e6)
        fr = [f for f in rows if f[0] in ("k_rng_stream", "k_witness_team") and f[2] > r0[2] - 300e6 and f[1] < r1[1] + 50e6]
        print("#   gap of %.2f ms between %s and %s, %.1f ms before K_transcript_A number %d; front kernels around it: %s" % (gp, r0[0], r1[0], (asm[ji][1] - r1[1]) / 1e6, ji,
              ", ".join("%s [%.1f .. %.1f]" % (f[0], (f[1] - r1[1]) / 1e6, (f[2] - r1[1]) / 1e6) for f in fr)))
tail = [r for r in rows if r[3] != hq and r[0].startswith("K_") and r[0] not in ("K_rng_reduce", "K_transcript_init", "K_commit_v", "K_load_inputs") and r[1] >= lo and r[2] <= hi]
if tail:
    first = {}
    spans = []
    ends = [r for r in tail if r[0] == "K_assemble"]
    for a0, a1 in zip([None] + ends[:-1], ends):
        t = [r for r in tail if (a0 is None or r[1] > a0[2]) and r[2] <= a1[2]]
        if t: spans.append((t[-1][2] - t[0][1]) / 1e6)
    print("# IPA tail on the jobs' own streams (next to the following job's sums): %d launches per job, %.2f ms of kernel time per job, %.1f ms from its first launch to K_assemble's end" %
          (len(tail) // nb, sum(r[2] - r[1] for r in tail) / 1e6 / nb, sum(spans) / max(1, len(spans))))
front = [r for r in rows if r[0] in ("k_rng_stream", "k_witness_team") and r[1] >= lo and r[2] <= hi]
fa = collections.defaultdict(lambda: [0, 0.0])
for r in front:
    a = fa[r[0]]; a[0] += 1; a[1] += (r[2] - r[1]) / 1e6
print("# front kernels (own streams, overlapping the back stream of the batch before): " + ", ".join("%s %.1f ms x %d" % (k, a[1] / a[0], a[0]) for k, a in fa.items()))
rng = [r for r in rows if r[0] in ("k_rng_stream",)]
one = [r for r in msm if r[1] >= asm[2][1] and r[2] <= asm[3][1]]
print("# MSM launches between two jobs' K_transcript_A (IPA of one job, commitment sums of the next): ms (share of it a TranscriptRng chain was running)")
print("  " + "  ".join("%.1f (%.0f %%)" % ((r[2] - r[1]) / 1e6, 100.0 * sum(max(0, min(r[2], f[2]) - max(r[1], f[1])) for f in rng) / (r[2] - r[1])) for r in one))
