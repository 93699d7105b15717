"""Per-kernel ISA statistics from a hipcc -save-temps .s file: registers, scratch, instruction mix, stray M0 uses.
usage: isa_stats.py file.s [name-filter]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    name, meta = m.group(1), m.group(2)
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("gs::", "")
    dn = re.sub(r"\(.*$", "", re.sub(r"^void ", "", dn))
    if flt and flt not in dn:
        continue
    a = s.index("\n" + name + ":")
    b = s.index(".Lfunc_end", a)
    body = s[a:b].split("\n")
    ops = [l.strip().split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    cnt = {}
    for o in ops:
        cnt[o] = cnt.get(o, 0) + 1
    grp = lambda pre: sum(v for k, v in cnt.items() if k.startswith(pre))
    stray_m0 = [l.strip() for l in body if re.search(r"\bm0\b", l) and "s_mov_b32 m0" not in l]
    vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1)
    sg = re.search(r"\.amdhsa_next_free_sgpr (\d+)", meta).group(1)
    sc = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1)
    print(f"{dn[:60]:60s} vgpr {vg:>3} sgpr {sg:>3} scratch {sc:>4} | insts {len(ops):5d} mfma {grp('v_mfma'):4d} ds_read {grp('ds_read'):4d} "
          f"dma {cnt.get('buffer_load_dwordx4', 0):3d} salu {grp('s_'):5d} valu {grp('v_') - grp('v_mfma'):5d} lane-spill {grp('v_writelane') + grp('v_readlane'):3d} "
          f"accvgpr {grp('v_accvgpr'):4d} waitcnt {cnt.get('s_waitcnt', 0):3d} barrier {cnt.get('s_barrier', 0)} stray-m0 {len(stray_m0)}")
