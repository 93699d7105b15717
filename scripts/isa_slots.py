"""Per-MFMA-slot instruction budget of a kernel from a hipcc -S listing.
For every v_mfma of the kernel body: the instructions issued since the previous one, by class (VALU / SALU / LDS / VMEM / waitcnt / other), and the
mnemonics themselves for the slots of the steady-state loop.  usage: isa_slots.py file.s '<demangled-name substring>' [--full]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2]
full = "--full" in sys.argv
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    name = m.group(1)
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("gs::", "")
    if flt not in dn:
        continue
    a = s.index("\n" + name + ":")
    b = s.index(".Lfunc_end", a)
    body = s[a:b].split("\n")
    print("#", re.sub(r"\(.*$", "", dn))
    slot, slots, labels = [], [], []
    for l in body:
        t = l.strip()
        if not t or t.startswith((";", ".")) and not t.endswith(":"):
            continue
        if t.endswith(":") and not l.startswith("\t"):
            slot.append("@" + t[:-1])
            continue
        if not l.startswith("\t"):
            continue
        op = t.split()[0]
        if op.startswith("v_mfma"):
            slots.append(slot)
            slot = []
        else:
            slot.append(op if not op.startswith("s_waitcnt") else "s_waitcnt " + " ".join(t.split()[1:]).split(";")[0].strip())
    slots.append(slot)

    def cls(op):
        if op.startswith("@"):
            return "label"
        if op.startswith("s_waitcnt"):
            return "wait"
        if op.startswith(("s_barrier",)):
            return "barrier"
        if op.startswith("v_"):
            return "valu"
        if op.startswith("s_"):
            return "salu"
        if op.startswith("ds_"):
            return "lds"
        if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
            return "vmem"
        return "other"
    tot = {}
    print("slot  valu salu lds vmem wait | instructions")
    for i, sl in enumerate(slots):
        c = {}
        for op in sl:
            c[cls(op)] = c.get(cls(op), 0) + 1
            tot[cls(op)] = tot.get(cls(op), 0) + 1
        n = sum(v for k, v in c.items() if k != "label")
        line = "%4d  %4d %4d %3d %4d %4d | " % (i, c.get("valu", 0), c.get("salu", 0), c.get("lds", 0), c.get("vmem", 0), c.get("wait", 0))
        if full or n > 6 or any(op.startswith("@") for op in sl):
            line += " ".join(sl)
        else:
            line += " ".join(sl)
        print(line)
    nm = len(slots) - 1
    print("total: %d MFMAs; non-MFMA per MFMA: %.2f (%s)" % (nm, sum(v for k, v in tot.items() if k != "label") / max(nm, 1), tot))
