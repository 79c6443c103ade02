#!/usr/bin/env python3
"""Timeline of the LAST prove call of a rocprofv3 --kernel-trace run (the single-proof latency probe): every launch after the
last idle gap longer than `gap_ms` on all queues, with its offset, duration and the idle time before it on its queue, then the
per-kernel totals of that window.   trace_lastcall.py DIR [gap_ms=20] [max_rows=400] [marker=K_assemble]
(marker: the kernel that ends the call looked for - K_assemble for a prove call, K_verify_finish for a verify call)"""
import sqlite3, sys, glob, collections
db = sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1]
gap_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
max_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 400
marker = sys.argv[4] if len(sys.argv) > 4 else "K_assemble"
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith("kernels")][0]
def short(n):
    n = n.split("(")[0]
    for k in ("k_rng_stream", "k_witness_team", "k_msm_fixed2", "k_poseidon_team"):
        if k in n: return k
    if "k_functor_wave<" in n: return n.split("k_functor_wave<")[1].rsplit(">", 1)[0]
    if "k_functor<" in n: return n.split("k_functor<")[1].rsplit(">", 1)[0]
    return n[:40]
rows = [(short(n), s, e, q) for n, s, e, q in con.execute("select name, start, end, queue_id from %s order by start" % kt)]
# the window of the last prove call: it ends with the last K_assemble (proof bytes); walk back from there to the first launch
# that follows an idle gap longer than gap_ms on all queues
last = max(i for i, r in enumerate(rows) if marker in r[0])
rows = rows[:last + 1]
cut = 0
busy_until = 0
for i, r in enumerate(rows):
    if i and r[1] - busy_until > gap_ms * 1e6: cut = i
    busy_until = max(busy_until, r[2])
win = rows[cut:]
t0 = win[0][1]
print("# last call: %d launches, %.3f ms from the first start to the last end" % (len(win), (max(r[2] for r in win) - t0) / 1e6))
last_end = {}
tot = collections.defaultdict(lambda: [0, 0.0])
for k, (name, s, e, q) in enumerate(win):
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    tot[name][0] += 1; tot[name][1] += (e - s) / 1e6
    if k < max_rows:
        print("%9.3f ms  q%-3d %-44s %9.1f us   idle before %8.1f us" % ((s - t0) / 1e6, q, name[:44], (e - s) / 1e3, gap))
print("# per kernel (calls, total ms)")
for name, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-50s %5d %9.3f" % (name[:50], c, ms))
