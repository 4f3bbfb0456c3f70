#!/usr/bin/env python3
"""BASELINE.json configs[2]: correlation-volume stress at 768x1024 (KITTI-size), radius 4 -- achieved HBM GB/s of the
correlation build (4-mode cross-attention scores -> pooled volume + pyramid) and of the radius-4 lookup against the
chip's peak, timed with HIP events on the launch stream.  Prints one JSON line.

    python tools/bench_corr.py [--height 768 --width 1024 --batch 1 --reps 10 --precision mixed]
"""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from craft_amd import ops  # noqa: E402
from craft_amd.hip import Precision  # noqa: E402

HBM_PEAK_GBS = 8000.0


def timed(fn, reps):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def measure(height=768, width=1024, batch=1, reps=10, precision="mixed"):
    """-> the configs[2] record (dict); bench.py's `corr_cfg2` leg calls this in-process."""
    a = argparse.Namespace(height=height, width=width, batch=batch, reps=reps, precision=precision)
    prec = Precision.parse(a.precision)
    dev = torch.device("cuda")
    B, H8, W8 = a.batch, a.height // 8, a.width // 8
    N, C, M, R, L, r = H8 * W8, 256, 4, 7, 4, 4
    g = torch.Generator(device="cpu").manual_seed(0)
    q = torch.randn(B, N, C, generator=g).to(dev)
    k = torch.randn(B, N, C, generator=g).to(dev)
    tab = (torch.randn(2 * R + 1, 2 * R + 1, generator=g) * 0.5).to(dev)
    pyr = ops.CorrPyramid(B, H8, W8, L, dev, tiled=ops.fused_pyramid(C, M, prec, L, H8, W8) and not os.environ.get("CRAFT_NO_TILED_PYRAMID"))
    scale = 1.0 / math.sqrt(C // M)

    def build():
        pyr.sums.zero_()
        ops.corr_build(q, k, H8, W8, M, scale, tab, 0.5, 0.7, None, pyr, True, prec)

    ms_build = timed(build, a.reps)
    # algorithmic bytes: read Q, K; write every pyramid level once.  The unfused path (CRAFT_NO_FUSED_PYRAMID,
    # other level counts, images under 64 px) reads level 0 back in its pooling pass: those bytes are counted only when that pass runs
    lvl = [pyr.lv[l].shape[0] * h * w * 4 for l, (h, w) in enumerate(pyr.dims)]           # (image sizes: a tiled level's padding is not counted)
    fused = L == 4 and min(H8, W8) >= 8 and not os.environ.get("CRAFT_NO_FUSED_PYRAMID")
    bytes_build = 2 * B * N * C * 4 + sum(lvl) + (0 if fused else lvl[0])
    coords = (torch.rand(B, N, 2, generator=g) * torch.tensor([W8 - 1.0, H8 - 1.0])).to(dev)
    out = torch.empty(B, N, L * (2 * r + 1) ** 2, device=dev)
    ms_look = timed(lambda: ops.corr_lookup(pyr, coords, r, out=out), a.reps)
    # lookup: <= (2r+2)^2 taps per level and query gathered, (2r+1)^2 * L outputs written
    bytes_look = B * N * (L * (2 * r + 2) ** 2 * 4 + L * (2 * r + 1) ** 2 * 4 + 8)
    line = {
        "workload": f"configs[2]: corr-volume stress {a.height}x{a.width}, batch {B}, radius {r}, {L} levels, 4-mode cross-attention scores",
        "volume_bytes": sum(lvl), "N": N,
        "corr_build": {"ms": round(ms_build, 4), "bytes": bytes_build, "pyramid_fused": bool(fused), "achieved_GBs": round(bytes_build / ms_build / 1e6, 1),
                       "frac_of_hbm_peak": round(bytes_build / ms_build / 1e6 / HBM_PEAK_GBS, 4),
                       "tflops_algorithmic": round(2.0 * B * N * N * C / ms_build / 1e9, 1)},
        "corr_lookup": {"ms": round(ms_look, 4), "bytes": bytes_look, "achieved_GBs": round(bytes_look / ms_look / 1e6, 1),
                        "frac_of_hbm_peak": round(bytes_look / ms_look / 1e6 / HBM_PEAK_GBS, 4)},
        "precision": repr(prec), "hbm_peak_GBs": HBM_PEAK_GBS,
    }
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=768)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--precision", default="mixed")
    a = ap.parse_args()
    print(json.dumps(measure(a.height, a.width, a.batch, a.reps, a.precision)))


if __name__ == "__main__":
    main()
