#!/usr/bin/env python3
"""GPU experiment: k_conv_halo time versus K-loop length (fixed cost vs per-K-tile cost)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from craft_amd import ops
from craft_amd.hip import PREC_F16X3, PREC_F32

def t(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

dev = torch.device("cuda"); B, H8, W8 = 4, 56, 128; N = H8 * W8
for prec, name in ((PREC_F16X3, "f16x3"), (PREC_F32, "fp32")):
    for (KH, KW) in ((1, 5), (3, 3)):
        for cout in (128, 256):
            row = []
            for cin in (32, 64, 128, 256, 512):
                x = torch.randn(B, N, cin, device=dev)
                w = torch.randn(cout, cin, KH, KW, device=dev) * 0.02
                b = torch.zeros(cout, device=dev)
                wp = ops.pack_conv_prec(w, prec)
                y = torch.empty(B, N, cout, device=dev)
                us = t(lambda: ops.conv2d_tokens(x, (H8, W8), wp, b, cout, KH, KW, 0, prec, packed=prec != PREC_F32, out=y))
                row.append((cin * KH * KW // 32, us))
            print(f"{name} {KH}x{KW} cout={cout}: " + "  ".join(f"ktiles={k}: {u:.0f}us" for k, u in row), flush=True)
