#!/usr/bin/env python3
"""Instruction mix of a kernel's main loop from hipcc -S output: python tools/isa_loop_mix.py <file.s> <mangled kernel name substring> ...
(the innermost loop with the most MFMAs: from its '=>This Inner Loop Header' label to the branch back to it)."""
import re, sys
from collections import Counter


def loops(lines):
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):.*Inner Loop Header", l)
        if m:
            start = (m.group(1), i)
        elif start and re.search(r"s_c?branch\w* " + re.escape(start[0]) + r"\b", l):
            yield lines[start[1]:i + 1]
            start = None


def classify(ins):
    if ins.startswith("v_mfma"): return "mfma"
    if ins.startswith("ds_read"): return "lds_read"
    if ins.startswith("ds_write"): return "lds_write"
    if ins.startswith("buffer_load") and "lds" in ins: return "lds_dma"
    if ins.startswith(("global_load", "buffer_load")): return "vmem_load"
    if ins.startswith(("global_store", "buffer_store")): return "vmem_store"
    if ins.startswith("s_waitcnt"): return "s_waitcnt"
    if ins.startswith("s_barrier"): return "s_barrier"
    if ins.startswith("s_"): return "salu"
    if ins.startswith("v_"): return "valu"
    return "other"


def main():
    text = open(sys.argv[1]).read().splitlines()
    for name in sys.argv[2:]:
        idx = [i for i, l in enumerate(text) if l.startswith("_Z") and name in l and l.rstrip().endswith(":") or (l.startswith("_Z") and name in l and ": " in l)]
        if not idx:
            print(name, "not found"); continue
        body = []
        for l in text[idx[0] + 1:]:
            if l.startswith(".Lfunc_end"):
                break
            body.append(l)
        best = max(loops(body), key=lambda b: sum("v_mfma" in x for x in b), default=None)
        if best is None:
            print(name, "no loop"); continue
        c = Counter(classify(l.strip()) for l in best if l.startswith("\t") and not l.strip().startswith((";", ".")))
        n = c.get("mfma", 1)
        print(f"{name}: main loop {sum(c.values())} instructions, {n} MFMAs")
        print("   " + "  ".join(f"{k} {v} ({v / n:.2f}/mfma)" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
