#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2) SQLite output.
  rocprof_summary.py stats <dir>            kernel | calls | total_ms | avg_ms | percent   (from --kernel-trace)
  rocprof_summary.py pmc <dir> [<dir> ...]  kernel | dispatches | <counter> avg per dispatch ... | avg duration ms
"""
import sqlite3, sys, glob, collections


def open_db(d):
    db = sorted(glob.glob(d + "/**/*.db", recursive=True))[-1]
    return sqlite3.connect(db)


def short(n):
    n = n.split("(")[0]
    for k in ("k_rng_stream", "k_witness_team", "k_msm_fixed2", "k_poseidon_team", "k_probe_mad", "k_probe_madd"):
        if k in n:
            return k
    if "k_functor" in n and "<" in n:
        return n.split("<", 1)[1].rsplit(">", 1)[0]
    return n[:48]


def kernel_table(con):
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    return [t for t in tabs if t.startswith("kernels")][0], tabs


def stats(d):
    con = open_db(d)
    kt, _ = kernel_table(con)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, s, e in con.execute("select name, start, end from %s" % kt):
        a = agg[short(n)]
        a[0] += 1
        a[1] += (e - s) / 1e6
    tot = sum(a[1] for a in agg.values())
    print("# kernel | calls | total_ms | avg_ms | percent")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%s | %d | %.3f | %.4f | %.2f" % (k, a[0], a[1], a[1] / a[0], 100 * a[1] / tot))
    # the dominant kernel by launch geometry: the trace mixes the prove jobs' launches (7 per 4096-proof job: A_I/A_O, S, four un-folded
    # rounds, the fold) with the warm-up's, the verifier's and the parity batch's - bench.py's roofline.avg_launch_ms is the mean over
    # the launches of the TIMED prove call only, i.e. over the geometries that occur once per full-size job
    cols = [r[1] for r in con.execute("pragma table_info(%s)" % kt)]
    gcol = next((c for c in ("grid_size_x", "grid_x", "grid_size") if c in cols), None)
    if gcol:
        by = collections.defaultdict(lambda: [0, 0.0])
        for n, s, e, g in con.execute("select name, start, end, %s from %s" % (gcol, kt)):
            if "k_msm_fixed2" in n:
                a = by[int(g)]
                a[0] += 1
                a[1] += (e - s) / 1e6
        if by:
            print("# k_msm_fixed2 by launch geometry: %s | calls | total_ms | avg_ms" % gcol)
            for g, a in sorted(by.items(), key=lambda kv: -kv[1][1]):
                print("k_msm_fixed2 %s=%d | %d | %.3f | %.4f" % (gcol, g, a[0], a[1], a[1] / a[0]))
            # launches per full-size job: J = calls of the rarest geometry that matters (>= 5 % of the kernel's time) = number of such
            # jobs in the trace; a geometry with k J (+ stragglers of other callers) calls is launched k times per job
            tot_ms = sum(a[1] for a in by.values())
            big = {g: a for g, a in by.items() if a[1] >= 0.05 * tot_ms}
            J = min(a[0] for a in big.values())
            per_job = {g: max(1, round(a[0] / J)) for g, a in big.items()}
            n = sum(per_job.values())
            avg = sum(per_job[g] * big[g][1] / big[g][0] for g in big) / n
            print("k_msm_fixed2 [%d full-size jobs x %d launches per job: mean over a job's launches, what bench.py's roofline.avg_launch_ms is] | %d | %.3f | %.4f" % (
                J, n, J * n, J * n * avg, avg))


def pmc(dirs):
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    names = []
    for d in dirs:
        con = open_db(d)
        kt, tabs = kernel_table(con)
        ct = [t for t in tabs if t.startswith("counters_collection")][0]
        cols = [r[1] for r in con.execute("pragma table_info(%s)" % ct)]
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        ncol = "counter_name" if "counter_name" in cols else "counter"
        rows = con.execute("select %s, %s, value, dispatch_id from %s" % (kcol, ncol, ct)).fetchall()
        seen = collections.defaultdict(float)
        for kn, cn, v, did in rows:
            seen[(short(kn), cn, did)] += v   # sum over XCDs / instances of one dispatch
            if cn not in names:
                names.append(cn)
        for (k, cn, did), v in seen.items():
            a = per[k][cn]
            a[0] += 1
            a[1] += v
        for n, s, e in con.execute("select name, start, end from %s" % kt):
            a = dur[short(n)]
            a[0] += 1
            a[1] += (e - s) / 1e6
    print("# kernel | dispatches | " + " | ".join("%s avg/dispatch" % c for c in names) + " | avg duration ms (profiled)")
    order = sorted(per, key=lambda k: -max(per[k][c][1] for c in per[k]))
    for k in order:
        nd = max(per[k][c][0] for c in per[k])
        vals = ["%.1f" % (per[k][c][1] / per[k][c][0]) if per[k][c][0] else "-" for c in names]
        print("%s | %d | %s | %.3f" % (k, nd, " | ".join(vals), dur[k][1] / max(dur[k][0], 1)))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
