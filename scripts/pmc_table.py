"""Per-kernel (name + grid) PMC table from rocprofv3 rocpd databases: per-wave instruction mix and where the wave cycles go.
usage: pmc_table.py db [db ...]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n.replace("gs::", "")[:56]


agg = {}
for path in sys.argv[1:]:
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    g = "grid_size" if "grid_size" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
    for k, c, v, d, gs in cur.execute(f"select kernel_name, counter_name, value, dispatch_id, {g} from counters_collection"):
        a = agg.setdefault((short(k), gs), {}).setdefault(c, {})
        a[d] = a.get(d, 0.0) + v
for (k, gs), cs in sorted(agg.items()):
    m = {c: sum(v.values()) / len(v) for c, v in cs.items()}
    w = m.get("SQ_WAVES", 0) or 1
    wc = m.get("SQ_WAVE_CYCLES", 0) * 4 / w  # quad-cycles -> cycles per wave
    if wc == 0:
        continue
    f = lambda c: m.get(c, 0) / w
    print(f"{k:56s} grid {gs:7d} waves {int(w):5d} | cyc/wave {wc:8.0f} mfma-busy {100*f('SQ_VALU_MFMA_BUSY_CYCLES')/wc:5.1f}% "
          f"active {100*4*f('SQ_ACTIVE_INST_ANY')/wc:5.1f}% wait {100*4*f('SQ_WAIT_ANY')/wc:5.1f}% inst-stall {100*4*f('SQ_WAIT_INST_ANY')/wc:5.1f}% "
          f"(lds {100*4*f('SQ_WAIT_INST_LDS')/wc:4.1f}%) | per wave: mfma {f('SQ_INSTS_MFMA'):6.0f} valu {f('SQ_INSTS_VALU'):6.0f} salu {f('SQ_INSTS_SALU'):6.0f} "
          f"lds {f('SQ_INSTS_LDS'):6.0f} vmrd {f('SQ_INSTS_VMEM_RD'):5.0f} vmwr {f('SQ_INSTS_VMEM_WR'):5.0f} | lds-active/wave {f('SQ_LDS_IDX_ACTIVE'):7.0f} conflicts {f('SQ_LDS_BANK_CONFLICT'):6.0f}")
