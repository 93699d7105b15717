"""Per-kernel averages of the PMC counters stored in rocprofv3 rocpd databases.  usage: pmc_summary.py db [db ...]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n.replace("gs::", "")[:64]


for path in sys.argv[1:]:
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection").fetchall()
    agg = {}
    for k, c, v, d in rows:
        a = agg.setdefault((short(k), c), {})
        a[d] = a.get(d, 0.0) + v
    for (k, c), per in sorted(agg.items()):
        if "igemm" in k or "wgrad" in k:
            vals = list(per.values())
            print(f"{k:64s} {c:26s} avg/dispatch {sum(vals)/len(vals):14.4g}  (n={len(vals)})")
