#!/usr/bin/env python3
"""Generate tests/golden/forward_interpolate.npz by RUNNING THE REFERENCE's warm-start helper
(/root/reference/core/utils/utils.py:34-62, scipy griddata 'nearest').  Build container only; the fixture is data:
seeded input flows and the reference's outputs.

    python tools/make_golden_interp.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference/core")

from utils.utils import forward_interpolate  # noqa: E402  (the reference)


def flows():
    g = torch.Generator().manual_seed(77)
    out = {}
    # smooth + noise, small / large displacements, one with many out-of-frame targets, one all-zero
    for name, (H, W, amp) in {"small_24x32": (24, 32, 2.0), "large_30x40": (30, 40, 25.0), "kitti_47x156": (47, 156, 6.0)}.items():
        base = torch.randn(2, H // 6 + 2, W // 6 + 2, generator=g) * amp
        f = torch.nn.functional.interpolate(base[None], size=(H, W), mode="bilinear", align_corners=True)[0]
        out[name] = (f + torch.randn(2, H, W, generator=g) * 0.37).float()
    out["zero_8x8"] = torch.zeros(2, 8, 8)
    return out


def main():
    data = {}
    for name, f in flows().items():
        ref = forward_interpolate(f)
        data[name + ".in"] = f.numpy()
        data[name + ".out"] = ref.numpy()
        print(name, tuple(f.shape), "changed pixels:", int((ref != f).any(0).sum()))
    path = os.path.join(ROOT, "tests", "golden", "forward_interpolate.npz")
    np.savez_compressed(path, **data)
    print("->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
