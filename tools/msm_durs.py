import sqlite3, glob, sys
db=sorted(glob.glob(sys.argv[1]+'/**/*.db', recursive=True))[-1]
con=sqlite3.connect(db)
kt=[r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')") if r[0].startswith('kernels')][0]
rows=con.execute("select name,start,end,grid_x from %s order by start"%kt).fetchall()
msm=[(round((e-s)/1e6,2),g) for n,s,e,g in rows if "k_msm_fixed2" in n or "K_msm_fixed" in n]
print(msm[-12:])
