#!/usr/bin/env python3
"""Dump every launch (all queues) of a rocprofv3 --kernel-trace database around the hand-over between two device jobs on the heavy
stream: from `before` ms ahead of the n-th K_transcript_A to `after` ms past it.   trace_window.py DIR [n before after]"""
import sqlite3, sys, glob, collections
db = sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith("kernels")][0]
def short(n):
    n = n.split("(")[0]
    for k in ("k_rng_stream", "k_witness_team", "k_msm_fixed2", "k_poseidon_team"):
        if k in n: return k
    if "k_functor_wave<" in n: return n.split("k_functor_wave<")[1].split(">")[0]
    if "k_functor<" in n: return n.split("k_functor<")[1].split(">")[0]
    return n[:40]
rows = [(short(n), s, e, q) for n, s, e, q in con.execute("select name, start, end, queue_id from %s order by start" % kt)]
msm = [r for r in rows if r[0] == "k_msm_fixed2"]
hq = collections.Counter(r[3] for r in msm).most_common(1)[0][0]
asm = [r for r in rows if r[0] == "K_transcript_A" and r[3] == hq]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
before = float(sys.argv[3]) if len(sys.argv) > 3 else 120.0
after = float(sys.argv[4]) if len(sys.argv) > 4 else 30.0
t0 = asm[n][1]
mc = [t for t in tabs if t.startswith("memory_copies")]
ev = [(r[0], r[1], r[2], "q%d%s" % (r[3], "*" if r[3] == hq else "")) for r in rows]
if mc:
    cols = [r[1] for r in con.execute("pragma table_info(%s)" % mc[0])]
    nm = "name" if "name" in cols else cols[0]
    for r in con.execute("select %s, start, end from %s" % (nm, mc[0])):
        ev.append(("copy:" + str(r[0]), r[1], r[2], "dma"))
print("# t = 0 at the start of K_transcript_A number %d on the heavy queue q%d (marked *)" % (n, hq))
last_end = {}
for name, s, e, q in sorted(ev, key=lambda r: r[1]):
    if s < t0 - before * 1e6 or s > t0 + after * 1e6: continue
    gap = (s - last_end[q]) / 1e6 if q in last_end else 0.0
    last_end[q] = e
    if (e - s) < 30e3 and gap < 0.05 and not q.endswith("*"): continue   # the tail's small kernels
    print("%9.3f  %-5s %-28s %9.3f ms%s" % ((s - t0) / 1e6, q, name, (e - s) / 1e6, "   <- idle %.2f ms before" % gap if gap > 0.05 else ""))
