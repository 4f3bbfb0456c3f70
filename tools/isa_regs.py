"""Print (vgpr, agpr, sgpr, spills, lds) per kernel from a hipcc -S assembly file.  usage: isa_regs.py file.s [filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if flt in name:
        print(f"{name[:90]:90s} agpr {blk.split()[0]:>3s} vgpr {g('vgpr_count'):>3s} sgpr {g('sgpr_count'):>3s} spill {g('vgpr_spill_count')} lds {g('group_segment_fixed_size')}")
