#!/usr/bin/env python3
"""GPU experiment: end-point error and speed of per-role precision policies against the fp32 HIP path.
Writes a table to stdout (run on the GPU box; results are recorded in DESIGN.md)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from craft_amd import CRAFT, default_args  # noqa: E402
from craft_amd.synth import synth_pair, synth_state_dict  # noqa: E402

POLICIES = ["fp32", "mixed", "proj=f16x3,score=f16x3,pv=fp16,conv=fp16,enc=f16x3", "proj=f16x3,score=f16x3,pv=fp16,conv=bf16,enc=f16x3",
            "proj=f16x3,score=f16x3,pv=fp16,conv=f16x3,enc=fp16", "mixed_fp32conv", "f16x3", "conv=f16x3", "proj=f16x3,score=f16x3", "fp16", "bf16",
            "score=fp16,pv=fp16", "score=bf16,pv=bf16", "score=bf16,pv=fp16",
            "proj=fp16,score=fp16,pv=fp16", "conv=fp16", "conv=bf16",
            "score=fp16,pv=fp16,conv=fp16", "pv=fp16", "score=fp16", "proj=fp16"]


def run(policy, im1, im2, iters, sd, reps=2):
    m = CRAFT(default_args(hip_precision=policy))
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        out = m(im1, im2, iters=iters, test_mode=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = m(im1, im2, iters=iters, test_mode=1)
        torch.cuda.synchronize()
    return out[1], (time.perf_counter() - t0) / reps * 1e3


def main():
    sd = synth_state_dict(CRAFT(default_args()).state_dict(), seed=1234)
    for (B, H, W, iters) in ((1, 128, 256, 4), (1, 448, 1024, 12)):
        im1, im2, _ = synth_pair(B, H, W, seed=0)
        im1, im2 = im1.cuda(), im2.cuda()
        ref, t_ref = run("fp32", im1, im2, iters, sd)
        print(f"--- {H}x{W} B={B} iters={iters}   (fp32: {t_ref:.2f} ms/forward; |flow| max {ref.abs().max():.2f} px)")
        for pol in POLICIES[1:]:
            up, t = run(pol, im1, im2, iters, sd)
            epe = (up - ref).pow(2).sum(1).sqrt()
            print(f"{pol:34s} mean EPE d {epe.mean().item():.5f}  max {epe.max().item():.4f}  {t:8.2f} ms/forward", flush=True)


if __name__ == "__main__":
    main()
