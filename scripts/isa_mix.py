"""Instruction mix per kernel of a hipcc -S listing: usage isa_mix.py file.s [name-substring]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end", s, re.S):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    ins = [l.split()[0] for l in (x.strip() for x in body.split("\n")) if l and not l.startswith((".", ";")) and not l.endswith(":")]
    c = collections.Counter(ins)
    groups = collections.Counter()
    for k, v in c.items():
        g = ("lane" if k in ("v_readlane_b32", "v_writelane_b32") else "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else
             "ds" if k.startswith("ds_") else "vmem" if k.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other")
        groups[g] += v
    print(name[:70], "total", sum(c.values()), dict(groups))
    print("  ", c.most_common(30))
