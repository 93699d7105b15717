cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d /tmp/pg -o g -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-spectral --no-launch-count --no-f32-leg > /tmp/pg.log 2>&1
f=$(find /tmp/pg -name "*.db" | head -1)
python - <<PY
import sqlite3,re,collections
db=sqlite3.connect("$f")
rows=db.execute("select name, start, end from kernels order by start").fetchall()
# find graph-replay steady region: take kernels between 30% and 55% of the trace (timed steps are replayed graphs after warmup, eager profiling comes later)
n=len(rows)
seg=rows[int(n*0.25):int(n*0.45)]
busy=sum(e-s for _,s,e in seg); span=seg[-1][2]-seg[0][1]
gaps=[seg[i+1][1]-seg[i][2] for i in range(len(seg)-1)]
pos=[g for g in gaps if g>0]
print(len(seg),"kernels span %.3f ms busy %.3f ms (%.1f%%) gaps %.3f ms; median gap %.2f us, mean %.2f us" % (span/1e6,busy/1e6,100*busy/span,sum(pos)/1e6, sorted(pos)[len(pos)//2]/1e3, sum(pos)/len(pos)/1e3))
big=[(g,seg[i+1][0]) for i,g in enumerate(gaps) if g>20000]
print("gaps > 20 us:", len(big), "total %.3f ms" % (sum(g for g,_ in big)/1e6))
c=collections.Counter()
for g,nm in big: c[re.sub(r"\(.*","",nm)[:60]]+=g
for k,v in c.most_common(8): print("   after-gap kernel %-60s %.3f ms" % (k, v/1e6))
PY
