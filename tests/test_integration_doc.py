"""INTEGRATION.md is executable: every fenced python block of its section 2 (the ctypes stub a reference maintainer would add) is
extracted from the document, executed against the built library, and its result compared with the oracle's restatement of the
replaced reference function (CorrBlock.__call__, core/corr.py:47-71).  The CPU half checks the stub's argtypes against the header."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def section2_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2."):text.index("## 3.")]
    return re.findall(r"```python\n(.*?)```", sec, flags=re.S)


def header_arg_count(name):
    hdr = open(os.path.join(ROOT, "include", "craft_hip.h")).read()
    m = re.search(r"\bint\s+" + name + r"\s*\(([^;]*?)\)\s*;", hdr, flags=re.S)
    assert m, name
    return len([a for a in m.group(1).split(",") if a.strip()])


def test_doc_stub_matches_header_arity():
    """(CPU) the documented argtypes list and call of craft_corr_lookup have exactly the header's number of parameters."""
    blocks = section2_blocks()
    assert blocks, "INTEGRATION.md section 2 lost its python block"
    src = blocks[0]
    n = header_arg_count("craft_corr_lookup")
    import ctypes
    ns = {}
    # evaluate only the argtypes expression (no library load on a box without the .so's dependencies)
    m = re.search(r"_lib\.craft_corr_lookup\.argtypes = (.*?\])\n", src.replace("\\\n", " "), flags=re.S)
    assert m
    argtypes = eval(m.group(1), {"ctypes": ctypes}, ns)        # noqa: S307 -- our own document
    assert len(argtypes) == n == 16
    call = re.search(r"_lib\.craft_corr_lookup\((.*?)\)\n\s+assert", src, flags=re.S).group(1)
    call = re.sub(r"#[^\n]*", "", call)
    depth, parts, cur = 0, [], ""
    for ch in call:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    parts.append(cur)
    assert len([p for p in parts if p.strip()]) == n


@pytest.mark.gpu
@pytest.mark.parametrize("r", [4, 3, 1])
def test_doc_stub_runs_and_matches_oracle(device, r):
    """(GPU) exec the documented stub, build a reference-style CorrBlock holder (normalised pyramid), compare with the oracle.
    r = 4 is the reference's radius (the kernel's compile-time form); 3 and 1 take its run-time-radius form."""
    from oracle import craft_oracle as O
    from craft_amd import hip
    os.environ["CRAFT_HIP_LIB"] = hip.lib_path()
    ns = {}
    for src in section2_blocks():
        exec(compile(src, "INTEGRATION.md", "exec"), ns)       # noqa: S102 -- our own document
    CorrBlock = ns["CorrBlock"]
    g = torch.Generator().manual_seed(7)
    B, H8, W8 = 2, 16, 24
    N = H8 * W8
    c = torch.randn(B, N, N, generator=g)
    mu, rstd = O.global_stats(c)
    chat = (c - mu[:, None, None]) * rstd[:, None, None]
    pyr = O.build_pyramid(chat, H8, W8, 4)
    coords = O.coords_grid(B, H8, W8) + torch.randn(B, 2, H8, W8, generator=g) * 6.0      # some windows leave the image
    want = O.corr_lookup(pyr, coords, r)
    blk = CorrBlock.__new__(CorrBlock)
    blk.corr_pyramid = [p.to(device) for p in pyr]
    blk.radius = r
    got = blk(coords.to(device)).cpu()
    assert got.shape == want.shape == (B, 4 * (2 * r + 1) ** 2, H8, W8)
    assert (got - want).abs().max().item() < 2e-5
