# the kernels of ONE replayed iteration in launch order (name, grid, duration, gap to the previous kernel): where the chains are.
# usage (GPU box): bash scripts/graph_sequence.sh <out-file>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=${1:-$R/gpurun_out/graph_sequence.txt}
rm -rf /tmp/pgs
rocprofv3 --kernel-trace -d /tmp/pgs -o g -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-spectral --no-launch-count --no-f32-leg > /tmp/pgs.log 2>&1
f=$(find /tmp/pgs -name "*.db" | head -1)
python - > $OUT <<PY
import sqlite3,re,collections
db=sqlite3.connect("$f")
cols=[r[1] for r in db.execute("pragma table_info(kernels)")]
gx="grid_x" if "grid_x" in cols else "grid_size_x"
wx="workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else "0")
rows=db.execute(f"select name, start, end, {gx}, {wx} from kernels order by start").fetchall()
idx=[i for i,r in enumerate(rows) if "adam_tf_kernel" in r[0]]
per=[(a,b,b-a) for a,b in zip(idx[:-2:2], idx[2::2])]
cnt=collections.Counter(p[2] for p in per).most_common(1)[0][0]
sel=[p for p in per if p[2]==cnt]
a,b,_=sel[len(sel)//2]
def short(n):
    n=re.sub(r"^void ","",n); n=re.sub(r"\(.*$","",n)
    return n.replace("gs::","").replace("__hip_bfloat16","bf16")[:100]
# how much of the iteration runs a few-block kernel with nothing beside it (the gaps a forked branch can fill)
ev=sorted([(s,1,g//w if w else g) for n,s,e,g,w in rows[a:b]]+[(e,-1,g//w if w else g) for n,s,e,g,w in rows[a:b]])
alone_small=alone_big=multi=idle=0.0; running=[]; last=ev[0][0]
for t,kind,blk in ev:
    dt=(t-last)/1e3
    if not running: idle+=dt
    if running:
        if len(running)==1 and running[0]<256: alone_small+=dt
        elif len(running)==1: alone_big+=dt
        else: multi+=dt
    last=t
    if kind==1: running.append(blk)
    else: running.remove(blk)
print("# kernel time: %.0f us one kernel of < 256 blocks alone, %.0f us one kernel of >= 256 blocks alone, %.0f us two or more kernels, %.0f us idle" % (alone_small, alone_big, multi, idle))
prev=None; t0=rows[a][1]
print("# one replayed iteration: %d kernels, %.3f ms" % (b-a,(rows[b][1]-rows[a][1])/1e6))
for n,s,e,g,w in rows[a:b]:
    gap=(s-prev)/1e3 if prev else 0.0
    blocks = g//w if w else g
    print("%8.1f us  +%5.1f  %7.1f us  %6d blk  %s" % ((s-t0)/1e3, gap, (e-s)/1e3, blocks, short(n)))
    prev=e
PY
echo wrote $OUT
