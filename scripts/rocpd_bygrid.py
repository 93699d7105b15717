"""Aggregate a rocprofv3 rocpd kernel trace by (kernel, grid) in first-seen order.  usage: rocpd_bygrid.py <db> [name-filter]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("gs::", "")[:90]


db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else "grid_size_x"
gy = "grid_y" if "grid_y" in cols else ("grid_size_y" if "grid_size_y" in cols else "0")
agg, order = {}, []
for n, s, e, a, b in db.execute(f"select name, start, end, {gx}, {gy} from kernels order by start"):
    k = (short(n), a, b)
    if flt and flt not in k[0]:
        continue
    if k not in agg:
        agg[k] = []
        order.append(k)
    agg[k].append(e - s)
for k in order:
    d = agg[k]
    print(f"{k[0]:60s} grid {k[1]:7d} x {k[2]:5d}  n {len(d):3d}  mean {sum(d)/len(d)/1e3:7.1f} us  min {min(d)/1e3:7.1f} us")
