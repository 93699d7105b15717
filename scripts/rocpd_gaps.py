"""Idle time between consecutive kernels of a rocprofv3 rocpd kernel trace (graph replay or eager): takes the last
`--tail` fraction of the trace (steady state), prints busy time, total gap time, the gap histogram and the kernels that
most often follow a long gap.  usage: rocpd_gaps.py <db> [tail fraction=0.5]"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("gs::", "")[:80]


db = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = db.execute("select name, start, end from kernels order by start").fetchall()
rows = rows[int(len(rows) * (1 - frac)):]
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - rows[0][1]
gaps = []
after = defaultdict(lambda: [0, 0.0])
prev_end = rows[0][2]
overlap = 0
for n, s, e in rows[1:]:
    g = s - prev_end
    if g < 0:
        overlap += 1
    gaps.append(g)
    a = after[short(n)]
    a[0] += 1
    a[1] += max(g, 0)
    prev_end = max(prev_end, e)
pos = [g for g in gaps if g > 0]
print(f"{len(rows)} kernels over {span/1e6:.3f} ms: busy {busy/1e6:.3f} ms ({100*busy/span:.1f} %), gaps {sum(pos)/1e6:.3f} ms ({100*sum(pos)/span:.1f} %), "
      f"{overlap} launches started before the previous one ended")
edges = [0, 500, 1000, 2000, 3000, 5000, 10000, 50000, 10**12]
for lo, hi in zip(edges, edges[1:]):
    sel = [g for g in pos if lo <= g < hi]
    print(f"  gap {lo/1e3:6.1f} .. {hi/1e3 if hi < 10**11 else float('inf'):8.1f} us: {len(sel):6d} gaps, {sum(sel)/1e6:8.3f} ms")
print("kernels by total gap time in front of them:")
for n, (c, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {n:80s} x{c:5d}  gap total {t/1e6:7.3f} ms  mean {t/c/1e3:6.2f} us")
