#!/usr/bin/env python
"""tests/golden/photo_pil.npz: the four ColorJitter operations of the reference's photometric augmentation (core/utils/augmentor.py:104,
:111-123 -- torchvision.transforms.ColorJitter on PIL images) computed by PILLOW ITSELF, the library torchvision's PIL path wraps
(_functional_pil.py: ImageEnhance.Brightness / Contrast / Color; the HSV round trip with a wrapped uint8 hue).  torchvision is absent from
this image, Pillow is present; the four wrappers are restated below (a few lines each).  Inputs: small uint8 images (random, nearly grey,
dark, saturated primaries + exact greys); outputs per (operation, factor) and for three full jitter chains.

    python tools/make_golden_photo.py          (re-running it on the committed tree reproduces the file bit for bit)
"""
import os
import sys

import numpy as np
from PIL import Image, ImageEnhance

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pil_op(a, op, f):
    img = Image.fromarray(a)
    if op == 0:
        return np.array(ImageEnhance.Brightness(img).enhance(f))
    if op == 1:
        return np.array(ImageEnhance.Contrast(img).enhance(f))
    if op == 2:
        return np.array(ImageEnhance.Color(img).enhance(f))
    h, s, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        np_h += np.int32(f * 255).astype(np.uint8)
    return np.array(Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB"))


FACTORS = {0: [0.6, 0.77, 1.0, 1.234, 1.4], 1: [0.6, 0.93, 1.0, 1.31, 1.4], 2: [0.6, 0.85, 1.0, 1.17, 1.4, 0.0],
           3: [-0.159, -0.05, -0.001, 0.0, 0.003, 0.1, 0.159]}
CHAINS = [((2, 0, 3, 1), (1.21, 0.83, 1.37, -0.113)), ((3, 1, 0, 2), (0.66, 1.39, 0.61, 0.158)), ((0, 1, 2, 3), (1.0, 1.0, 1.0, 0.0))]


def images():
    rs = np.random.RandomState(20260930)
    H, W = 32, 40
    a = rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
    base = rs.randint(0, 256, size=(H, W, 1))
    b = np.clip(base + rs.randint(-5, 6, size=(H, W, 3)), 0, 255).astype(np.uint8)
    b[::4, ::5] = base[::4, ::5]                                           # exact greys (s = 0)
    c = rs.randint(0, 14, size=(H, W, 3)).astype(np.uint8)                  # dark
    d = rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
    d[:, ::2, 0] = 255; d[::2, :, 1] = 0; d[1::3, 1::3] = (0, 0, 255); d[0, :8] = (255, 255, 255); d[1, :8] = 0      # primaries, white, black
    return {"rand": a, "grey": b, "dark": c, "prim": d}


def main():
    out = {}
    for name, a in images().items():
        out[f"{name}.in"] = a
        for op, facs in FACTORS.items():
            for k, f in enumerate(facs):
                out[f"{name}.op{op}.{k}"] = pil_op(a, op, f)
        for k, (order, fac) in enumerate(CHAINS):
            x = a
            for op in order:
                x = pil_op(x, op, fac[op])
            out[f"{name}.chain{k}"] = x
    for op, facs in FACTORS.items():
        out[f"factors{op}"] = np.array(facs, dtype=np.float64)
    out["chain_orders"] = np.array([c[0] for c in CHAINS], dtype=np.int64)
    out["chain_factors"] = np.array([c[1] for c in CHAINS], dtype=np.float64)
    path = os.path.join(ROOT, "tests", "golden", "photo_pil.npz")
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
