#!/usr/bin/env python3
"""Yardstick for the "power-capped MFMA roof" claim: the vendor GEMM (torch.matmul -> hipBLASLt) and this
library's two MFMA engines run back to back in ONE process on ONE box, each for >= 2 s on random data, with sclk / package power
sampled from rocm-smi.  Reports executed PFLOP/s (MFMA work actually issued: 3 x the algorithmic flops for f16x3), sclk and W.

The vendor GEMM never enters the product path; this file is a measurement tool only.

    python tools/roof_calibrate.py [--seconds 2.5] [--out gpurun_out/roof_calibrate.txt]
"""
import argparse
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def smi_sampler(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        except Exception:  # noqa: BLE001
            r = ""
        sclk = [ln.split("(")[-1].rstrip(")") for ln in r.splitlines() if "sclk" in ln]
        pw = [ln.split(":")[-1].strip() for ln in r.splitlines() if "Power" in ln and "W" in ln]
        try:
            out.append((float(sclk[0].lower().replace("mhz", "")), float(pw[0])))
        except (IndexError, ValueError):
            pass
        time.sleep(0.3)


def run_leg(name, fn, flops_exec, seconds, chunk=10):
    """fn() enqueues ONE launch; run it back to back for `seconds`; -> dict."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=smi_sampler, args=(stop, samples))
    th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t0 = 0, time.time()
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(chunk):
            fn()
        n += chunk
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    us = e0.elapsed_time(e1) / n * 1e3
    mid = samples[1:-1] or samples
    sclk = sum(s for s, _ in mid) / max(1, len(mid))
    pw = sum(p for _, p in mid) / max(1, len(mid))
    pf = flops_exec / (us * 1e-6) / 1e15
    return {"name": name, "us": us, "launches": n, "PF_exec": pf, "sclk_MHz": sclk, "W": pw, "samples": len(mid)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    dev = torch.device("cuda")
    from craft_amd import autograd as AG, ops
    from craft_amd.hip import PREC_BF16, PREC_F16, PREC_F16X3

    legs = []

    def matmul_leg(M, N, K, dt, zeros=False):
        A = torch.zeros(M, K, device=dev, dtype=dt) if zeros else torch.randn(M, K, device=dev, dtype=dt)
        Bm = torch.zeros(N, K, device=dev, dtype=dt) if zeros else torch.randn(N, K, device=dev, dtype=dt)
        C = torch.empty(M, N, device=dev, dtype=dt)
        nm = f"torch.matmul {str(dt).split('.')[-1]:8s} M={M} N={N} K={K} NT" + (" ZERO data" if zeros else "")
        legs.append(run_leg(nm, lambda: torch.matmul(A, Bm.t(), out=C), 2.0 * M * N * K, a.seconds))

    # the vendor GEMM at a large square and at the conv engine's GEMM-equivalent shape (GRU z|r conv at 448x1024 B=4: M = pixels
    # = 28 672, N = Cout = 256, K = 5 taps x 384 = 1 920)
    for dt in (torch.float16, torch.bfloat16):
        matmul_leg(8192, 8192, 8192, dt)
    matmul_leg(8192, 8192, 8192, torch.float16, zeros=True)
    for dt in (torch.float16, torch.bfloat16):
        matmul_leg(28672, 256, 1920, dt)
    # the training step's dominant weight-gradient GEMM shape as a plain GEMM: M = Cout 256, N = 9 x 128, K = pixels x 12 calls
    matmul_leg(256, 1152, 8 * 46 * 62 * 12, torch.float16)

    # this library: halo convolution (bench.py roofline_conv shape) in f16x3 (3 MFMAs / product) and in plain fp16 / bf16 (1 MFMA)
    B, H8, W8 = 4, 56, 128
    N = H8 * W8
    x = torch.randn(B, N, 384, device=dev)
    w = torch.randn(256, 384, 1, 5, device=dev) * 0.02
    bias = torch.zeros(256, device=dev)
    y = torch.empty(B, N, 256, device=dev)
    for nm, cp, mult in (("f16x3", PREC_F16X3, 3), ("fp16", PREC_F16, 1), ("bf16", PREC_BF16, 1)):
        wp = ops.pack_conv_prec(w, cp)
        fl = 2.0 * B * N * 256 * 5 * 384 * mult
        legs.append(run_leg(f"k_conv_halo_wf {nm:6s} 1x5 384->256 at 56x128 B=4 (executed = {mult} x algorithmic)",
                            lambda wp=wp, cp=cp: ops.conv2d_tokens(x, (H8, W8), wp, bias, 256, 1, 5, 0, cp, packed=True, out=y), fl, a.seconds))

    # this library: packed weight-gradient GEMM (bench.py train roofline shape: 3x3, 128 -> 256, 8 x 46 x 62 pixels, 12 segments)
    b, h, ww = 8, 46, 62
    xs = torch.randn(b, h * ww, 128, device=dev)
    dy = torch.randn(b, h * ww, 256, device=dev) * 1e-3
    acc = torch.zeros(256, 3, 3, 128, device=dev)
    geom = (b, h, ww, 1, 1)
    for nm, cp, mult in (("f16x3", PREC_F16X3, 3), ("fp16", PREC_F16, 1), ("bf16", PREC_BF16, 1)):
        gp, xp = AG.Packed(dy, cp, geom), AG.Packed(xs, cp, geom)
        fl = 2.0 * b * h * ww * 128 * 256 * 9 * 12 * mult
        legs.append(run_leg(f"k_gemm_pk      {nm:6s} wgrad 3x3 128->256, 8x46x62 px x 12 segments (executed = {mult} x algorithmic)",
                            lambda gp=gp, xp=xp: AG.wgrad_pk([(gp, xp)] * 12, 3, 3, acc), fl, a.seconds, chunk=5))

    lines = [f"# roof calibration: every leg back to back for >= {a.seconds} s, random data unless noted; one process, one box",
             f"# {torch.cuda.get_device_name(0)}; torch {torch.__version__}",
             f"{'leg':100s} {'us/launch':>10s} {'PF/s exec':>10s} {'sclk MHz':>9s} {'W':>7s} {'launches':>8s}"]
    for r in legs:
        lines.append(f"{r['name']:100s} {r['us']:10.1f} {r['PF_exec']:10.3f} {r['sclk_MHz']:9.0f} {r['W']:7.0f} {r['launches']:8d}")
    txt = "\n".join(lines)
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as fh:
            fh.write(txt + "\n")


if __name__ == "__main__":
    main()
