"""Compressed instruction-class listing of the MFMA loops of one kernel in a hipcc -S file.
usage: isa_loop.py file.s <kernel-substring> [--full]
M mfma, D ds_read, W ds_write, G global/buffer load, T store, v VALU, s SALU, b branch, [wait...] s_waitcnt, | barrier"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
full = "--full" in sys.argv
start = next(i for i, l in enumerate(src) if re.match(r"^_Z\w+:", l) and key in l)
end = next(i for i in range(start, len(src)) if src[i].strip().startswith("s_endpgm"))
body = src[start:end]
def cls(t):
    op = t.split()[0]
    if "mfma" in op: return "M"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "D"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "W"
    if op.startswith(("global_load", "buffer_load", "flat_load")): return "G"
    if op.startswith(("global_store", "buffer_store", "flat_store", "global_atomic")): return "T"
    if op == "s_waitcnt": return "[" + t.split(None, 1)[1].replace(" ", "") + "]"
    if op == "s_barrier": return "|"
    if op.startswith(("s_cbranch", "s_branch")): return "b"
    if op.startswith("v_exp"): return "x"
    if op.startswith("s_"): return "s"
    if op.startswith("v_"): return "v"
    return "?"
# blocks by label
blocks, cur, name = [], [], "entry"
for l in body:
    t = l.strip()
    if not t or t.startswith((";", ".")) and not t.endswith(":"): continue
    if t.endswith(":") and not t.startswith("s_") and not t.startswith("v_"):
        blocks.append((name, cur)); cur, name = [], t[:-1]; continue
    t = t.split(";")[0].strip()
    if t: cur.append(t)
blocks.append((name, cur))
for name, ins in blocks:
    s = "".join(cls(t) for t in ins)
    nm = s.count("M")
    if nm or full:
        print(f"== {name}: {len(ins)} instrs, {nm} mfma, valu {s.count('v')}, salu {s.count('s')}, ds_r {s.count('D')}, ds_w {s.count('W')}, gload {s.count('G')}")
        print(s)
