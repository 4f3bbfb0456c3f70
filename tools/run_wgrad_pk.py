#!/usr/bin/env python3
"""Launch the packed-operand weight-gradient kernel at one of the training step's launch shapes, alone (for rocprofv3 --pmc passes),
the 12 calls of a pass as one launch.  "x16": the X packs as one fp16 plane (role wgx: 2 MFMAs per product); "fp16": dY too (roles wgx + wgy of the "mixed" policy: 1 MFMA).  Default: the shape bench.py reports as dominant at configs[3] (flow head / mask head
conv1: 3x3, 128 -> 256 channels over 8 x 46 x 62 pixels); "gru": SepConvGRU z|r gates, 1x5, [h | mf | mfg] (384) -> 256.
usage: python tools/run_wgrad_pk.py [reps] [gru] [x16|fp16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from craft_amd import autograd as AG
from craft_amd.hip import PREC_F16, PREC_F16X3
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda")
B, H, W, calls = 8, 46, 62, 12
cin, cout, KH, KW = (384, 256, 1, 5) if "gru" in sys.argv[2:] else (128, 256, 3, 3)
geom = (B, H, W, KH // 2, KW // 2)
pairs = []
for _ in range(calls):
    x = torch.randn(B, H * W, cin, device=dev)
    dy = torch.randn(B, H * W, cout, device=dev) * 1e-2
    both = "fp16" in sys.argv[2:]
    pairs.append((AG.Packed(dy, PREC_F16 if both else PREC_F16X3, geom), AG.Packed(x, PREC_F16 if (both or "x16" in sys.argv[2:]) else PREC_F16X3, geom)))
acc = torch.zeros(cout, KH, KW, cin, device=dev)
for _ in range(reps):
    AG.wgrad_pk(pairs, KH, KW, acc)
torch.cuda.synchronize()
print("ok", float(acc.abs().mean()))
