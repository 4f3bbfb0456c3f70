#!/usr/bin/env python3
"""tests/golden/flowio_ref.npz: the reference's own file-format functions run from its source.

core/utils/frame_utils.py imports cv2 at the top, but ``readFlow``, ``writeFlow`` (.flo) and ``readPFM`` are numpy / re only: they are compiled
out of the file's AST (with its module constant TAG_CHAR) and run on seeded arrays.  The fixture holds the arrays, the BYTES the reference's
writeFlow produced for them (both call forms), what its readFlow returns for those bytes, and hand-made PFM files (colour / grey, little /
big endian, negative and non-unit scale) with what its readPFM returns.  tests/test_flow_io.py holds craft_amd.flow_io to them.

Only runs in the build container.      python tools/make_golden_flowio.py [--check]
"""
import ast
import os
import re
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/core/utils/frame_utils.py"


def ref_functions():
    tree = ast.parse(open(REF).read(), REF)
    ns = {"np": np, "re": re}
    body = [n for n in tree.body if (isinstance(n, ast.FunctionDef) and n.name in ("readFlow", "writeFlow", "readPFM"))
            or (isinstance(n, ast.Assign) and getattr(n.targets[0], "id", None) == "TAG_CHAR")]
    exec(compile(ast.Module(body=body, type_ignores=[]), REF, "exec"), ns)
    return ns


def pfm_bytes(img, little, scale):
    color = img.ndim == 3
    h, w = img.shape[:2]
    head = (b"PF\n" if color else b"Pf\n") + f"{w} {h}\n".encode() + f"{-scale if little else scale}\n".encode()
    return head + np.flipud(img).astype("<f4" if little else ">f4").tobytes()


def main():
    ns = ref_functions()
    rs = np.random.RandomState(3)
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for k, (h, w) in enumerate([(5, 7), (1, 1), (16, 3)]):
            uv = (rs.randn(h, w, 2) * 30).astype(np.float32)
            out[f"flo.{k}.uv"] = uv
            p = os.path.join(td, "a.flo")
            ns["writeFlow"](p, uv)
            out[f"flo.{k}.bytes"] = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
            ns["writeFlow"](p, uv[..., 0], uv[..., 1])
            assert np.array_equal(np.frombuffer(open(p, "rb").read(), dtype=np.uint8), out[f"flo.{k}.bytes"])
            out[f"flo.{k}.read"] = ns["readFlow"](p)
        # writeFlow casts through float64 -> float32: a float64 input with values not representable in float32
        uv64 = rs.randn(4, 6, 2) * 1e3
        p = os.path.join(td, "b.flo")
        ns["writeFlow"](p, uv64)
        out["flo.f64.uv"], out["flo.f64.bytes"] = uv64, np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
        cases = [((6, 5, 3), True, 1.0), ((6, 5, 3), False, 1.0), ((4, 9), True, 1.0), ((4, 9), False, 2.5), ((3, 2, 3), True, 0.5)]
        for k, (shape, little, scale) in enumerate(cases):
            img = (rs.randn(*shape) * 10).astype(np.float32)
            raw = pfm_bytes(img, little, scale)
            p = os.path.join(td, "c.pfm")
            open(p, "wb").write(raw)
            out[f"pfm.{k}.bytes"] = np.frombuffer(raw, dtype=np.uint8)
            out[f"pfm.{k}.read"] = np.ascontiguousarray(ns["readPFM"](p)).astype(np.float32)
        out["pfm.n"] = np.array(len(cases))
    path = os.path.join(ROOT, "tests", "golden", "flowio_ref.npz")
    if "--check" in sys.argv:
        z = np.load(path)
        bad = [k for k in out if not np.array_equal(z[k], out[k])]
        print("differs:", bad if bad else "nothing")
        return 1 if bad else 0
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
