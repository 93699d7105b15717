"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel table (markdown)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("gs::", "")
    return name[:110]


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else cols[0]
    rows = cur.execute(f"select {namecol}, (end - start) from kernels").fetchall()
    agg = {}
    for n, d in rows:
        a = agg.setdefault(short(n), [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print(f"total kernel time {total/1e6:.3f} ms over {len(rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"| {n} | {a[0]} | {a[1]/1e6:.3f} | {a[1]/a[0]/1e3:.1f} | {a[2]/1e3:.1f} | {a[3]/1e3:.1f} | {100*a[1]/total:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
