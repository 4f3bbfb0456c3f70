#!/usr/bin/env python3
"""Bandwidth of craft_pack_operand(s) at the training step's shapes (configs[3]: 8 x 46 x 62 pixels): one launch per tensor and the
batched launch of one update iteration's conv inputs.  GB/s = (fp32 bytes read + 16-bit bytes written) / time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from craft_amd import autograd as AG  # noqa: E402
from craft_amd.hip import PREC_F16, PREC_F16X3  # noqa: E402


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda")
    B, H, W = 8, 46, 62
    N = H * W
    g3, g15, g51 = (B, H, W, 1, 1), (B, H, W, 0, 2), (B, H, W, 2, 0)
    for prec, nm, wb in ((PREC_F16, "fp16", 2), (PREC_F16X3, "f16x3", 4)):
        for C, geom in ((256, g3), (128, g3), (256, g15), (352, None), (32, g3)):
            x = torch.randn(B, N, C, device=dev)
            cs = torch.zeros(C, device=dev)
            for colsum in (None, cs):
                us = timeit(lambda: AG.Packed(x, prec, geom, colsum=colsum))
                byts = x.numel() * (4 + wb)
                print(f"{nm:6s} C={C:4d} geom={'rows' if geom is None else geom[3:]} colsum={colsum is not None!s:5s} {us:7.1f} us  {byts / us / 1e3:7.1f} GB/s")
        # one update iteration's forward conv inputs in ONE launch (train_update.py:171-258)
        chans = [(352, None), (256, g3), (32, (B, H, W, 3, 3)), (128, g3), (256, g3), (128, None), (128, g15), (128, g15), (256, g15), (128, g51), (128, g51),
                 (256, g51), (128, g3), (256, g3), (256, None)]
        xs = [torch.randn(B, N, c, device=dev) for c, _ in chans]

        def batch():
            pb = AG.PackBatch()
            for x, (c, g) in zip(xs, chans):
                AG.Packed(x, prec, g, batch=pb)
            pb.flush()
        us = timeit(batch)
        byts = sum(x.numel() for x in xs) * (4 + wb)
        print(f"{nm:6s} batched iteration ({len(xs)} tensors, {byts / 1e6:.0f} MB moved) {us:7.1f} us  {byts / us / 1e3:7.1f} GB/s")


if __name__ == "__main__":
    main()
