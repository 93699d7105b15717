# kernel time by family INSIDE the replayed graphs (rocprofv3 kernel trace of a graph-mode bench run): one timed iteration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pg
rocprofv3 --kernel-trace -d /tmp/pg -o g -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-spectral --no-launch-count --no-f32-leg > /tmp/pg.log 2>&1
f=$(find /tmp/pg -name "*.db" | head -1)
python - <<PY
import sqlite3,re,collections
db=sqlite3.connect("$f")
rows=db.execute("select name, start, end from kernels order by start").fetchall()
# locate the timed region: the longest stretch of kernels whose Adam launches are a constant number of kernels apart (graph replays)
idx=[i for i,(n,s,e) in enumerate(rows) if "adam_tf_kernel" in n]
# an iteration = two adam launches (D then G); take iterations 6..10 of the replays after warm-up
per=[]
for a,b in zip(idx[:-2:2], idx[2::2]):
    per.append((a,b,rows[b][1]-rows[a][1], b-a))
# graph iterations have the smallest kernel count jitter; pick the most common count
cnt=collections.Counter(p[3] for p in per).most_common(1)[0][0]
sel=[p for p in per if p[3]==cnt][2:8]
fam=collections.Counter(); calls=collections.Counter(); tot=0; span=0
def family(n):
    for k in ("conv_igemm","conv_wgrad","wgrad_sk_reduce","wgrad_reduce","pixel_norm","thin_","dense_","channel_sum","weight_prep","adam_tf","act_bwd","axpby","bias_act","batch_stddev","at::native","rocclr","upscale","blocksum","sumsq","row_scale","gan_","embedding","conv_direct"):
        if k in n: return k
    return re.sub(r"[<(].*","",n)[:30]
for a,b,dt,c in sel:
    span+=dt
    for n,s,e in rows[a:b]:
        fam[family(n)]+=e-s; calls[family(n)]+=1; tot+=e-s
k=len(sel)
print("%d iterations of %d kernels: %.3f ms per iteration, kernel-busy %.3f ms" % (k,cnt,span/k/1e6,tot/k/1e6))
for f,v in fam.most_common(30): print("  %-24s %8.1f us  %5.1f launches  %5.1f %%" % (f, v/k/1e3, calls[f]/k, 100*v/tot))
PY
