import collections, sys
rows=[l.strip().split(',') for l in open(sys.argv[1]) if not l.startswith('#')]
rows=[(n,int(s),int(e),int(q),int(st)) for n,s,e,q,st in rows]
msm=[r for r in rows if r[0]=='k_msm_fixed2']
hq=collections.Counter(r[3] for r in msm).most_common(1)[0][0]
on=[r for r in rows if r[3]==hq]
asm=[r for r in on if r[0]=='K_assemble']
lo,hi=asm[1][2],asm[6][2]
w=[r for r in on if r[1]>=lo and r[2]<=hi]
agg=collections.defaultdict(lambda:[0,0.0])
for r in w: a=agg[r[0]]; a[0]+=1; a[1]+=(r[2]-r[1])/1e6
tot=sum(a[1] for a in agg.values())
print('steady-state period %.1f ms, back stream busy %.1f ms per job'%((hi-lo)/5e6, tot/5))
for k,a in sorted(agg.items(), key=lambda kv:-kv[1][1])[:int(sys.argv[2]) if len(sys.argv)>2 else 12]: print('%-28s %5.1f calls/job %8.2f ms/job %5.1f %%'%(k,a[0]/5,a[1]/5,100*a[1]/tot))
