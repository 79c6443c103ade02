#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace SQLite database: per-launch timeline of the latency-bound front kernels
and the durations of the heavy kernels split by whether a front kernel was running at the same time."""
import sqlite3, sys, glob, collections
db = sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel_dispatch" in t]
cols = [r[1] for r in con.execute("pragma table_info(%s)" % kt[0])]
print("# table", kt[0], cols, file=sys.stderr)
rows = con.execute("select name, start, end, queue_id from %s order by start" % kt[0]).fetchall()
t0 = rows[0][1]
front = [(n, s, e, q) for n, s, e, q in rows if "k_rng_stream" in n or "k_witness_team" in n]
def short(n):
    n = n.split("(")[0]
    for k in ("k_rng_stream", "k_witness_team"):
        if k in n: return k
    if "k_msm_fixed2" in n: return "k_msm_fixed2"
    if "k_functor_wave<" in n: return n.split("k_functor_wave<")[1].split(">")[0]
    if "k_functor<" in n: return n.split("k_functor<")[1].split(">")[0]
    return n[:40]
print("# front kernels: name queue start_ms end_ms dur_ms")
for n, s, e, q in front:
    print("%-16s q%-3d %9.1f %9.1f %8.1f" % (short(n), q, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
def overlapped(s, e):
    return any(fs < e and fe > s for _, fs, fe, _ in front)
agg = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
for n, s, e, q in rows:
    k = short(n)
    if k in ("k_rng_stream", "k_witness_team"): continue
    a = agg[k]
    if overlapped(s, e): a[2] += 1; a[3] += (e - s) / 1e6
    else: a[0] += 1; a[1] += (e - s) / 1e6
print("# heavy kernels: name | alone: launches avg_ms | with a front kernel running: launches avg_ms")
for k, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][3]))[:14]:
    print("%-24s %5d %9.3f | %5d %9.3f" % (k, a[0], a[1] / max(a[0], 1), a[2], a[3] / max(a[2], 1)))
# busy / idle time of the queue the MSM kernel runs on (the shared "back" stream), between the first and the last MSM launch
msm = [(s, e, q) for n, s, e, q in rows if "k_msm_fixed2" in n]
if msm:
    hq = collections.Counter(q for _, _, q in msm).most_common(1)[0][0]
    lo, hi = msm[len(msm) // 4][0], msm[-1][1]   # skip the setup / warm-up quarter
    on = sorted((s, e, short(n)) for n, s, e, q in rows if q == hq and s >= lo and e <= hi)
    busy = sum(e - s for s, e, _ in on)
    gaps = collections.defaultdict(lambda: [0, 0.0])
    for (s0, e0, n0), (s1, e1, n1) in zip(on, on[1:]):
        g = (s1 - e0) / 1e6
        if g > 0.02:
            a = gaps[n0 + " -> " + n1]; a[0] += 1; a[1] += g
    print("# back stream q%d over %.1f ms: busy %.1f ms (%.1f %%), %d launches" % (hq, (hi - lo) / 1e6, busy / 1e6, 100.0 * busy / (hi - lo), len(on)))
    print("# gaps > 20 us on it, by neighbour pair: count total_ms")
    for k, a in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
        print("%-60s %4d %8.2f" % (k, a[0], a[1]))
