"""Instruction count per kernel of a hipcc -S file (finds bloated epilogues).  usage: isa_size.py file.s [min]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
mn = int(sys.argv[2]) if len(sys.argv) > 2 else 0
name, n, out = None, 0, []
for ln in lines:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        name, n = m.group(1), 0
        continue
    if name:
        t = ln.strip()
        if t.startswith("s_endpgm"):
            out.append((n, name)); name = None
        elif t and not t.startswith((";", ".")) and not t.endswith(":"):
            n += 1
for n, name in sorted(out, reverse=True):
    if n >= mn:
        print(f"{n:7d}  {name[:120]}")
