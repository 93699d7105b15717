"""List a rocprofv3 rocpd kernel trace in dispatch order, collapsing consecutive launches of the same kernel
(name + grid) into one line with the mean / min duration.  usage: rocpd_runs.py <db> [name-filter]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("gs::", "")[:100]


db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gcol = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
sel = f"name, start, end, {gcol}" if gcol else "name, start, end, 0"
rows = db.execute(f"select {sel} from kernels order by start").fetchall()
runs = []
for n, s, e, g in rows:
    k = (short(n), g)
    if flt and flt not in k[0]:
        continue
    if runs and runs[-1][0] == k:
        runs[-1][1].append(e - s)
    else:
        runs.append((k, [e - s]))
for (n, g), d in runs:
    print(f"{n:70s} grid {g:7d} x{len(d):3d}  mean {sum(d)/len(d)/1e3:7.1f} us  min {min(d)/1e3:7.1f} us")
