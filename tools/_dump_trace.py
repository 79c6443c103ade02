import sqlite3, sys, glob
db = sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith("kernels")][0]
cols = [r[1] for r in con.execute("pragma table_info(%s)" % kt)]
print("#", cols)
rows = con.execute("select name, start, end, queue_id, stream_id from %s order by start" % kt).fetchall() if "stream_id" in cols else con.execute("select name, start, end, queue_id, 0 from %s order by start" % kt).fetchall()
t0 = rows[0][1]
def short(n):
    n = n.split("(")[0]
    for k in ("k_rng_stream", "k_witness_team", "k_msm_fixed2", "k_poseidon_team"):
        if k in n: return k
    if "k_functor_wave<" in n: return n.split("k_functor_wave<")[1].split(">")[0]
    if "k_functor<" in n: return n.split("k_functor<")[1].split(">")[0]
    return n[:40]
for n, s, e, q, st in rows:
    print("%s,%d,%d,%d,%d" % (short(n), s - t0, e - t0, q, st))
