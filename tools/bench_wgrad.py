"""Weight-gradient kernels at the training step's shapes: round-2 craft_conv2d_wgrad (fp32 operands split in the K loop) against
round-3 craft_pack_operand x 2 + craft_wgrad_pk (packed operands).  usage: python tools/bench_wgrad.py [--cfg 3|4] [--prec f16x3]"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from craft_amd import autograd as AG
from craft_amd.hip import PREC_NAMES, call

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", type=int, default=3); ap.add_argument("--prec", default="f16x3"); ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
B, H, W = {3: (8, 46, 62), 4: (4, 46, 96)}[a.cfg]
prec = PREC_NAMES[a.prec]
dev = torch.device("cuda")
SHAPES = [("gru z|r 1x5", 512, 256, 1, 5), ("gru z|r 5x1", 512, 256, 5, 1), ("gru q 1x5", 512, 128, 1, 5), ("convc2 3x3", 256, 192, 3, 3),
          ("menc conv 3x3", 256, 128, 3, 3), ("fh/mask conv1 3x3", 128, 256, 3, 3), ("convf2 3x3", 128, 64, 3, 3), ("fh conv2 3x3", 256, 32, 3, 3),
          ("convf1 7x7", 32, 128, 7, 7), ("enc 64 3x3 @184x248", 64, 64, 3, 3)]


def timed(fn, reps):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

rows = []
for name, cin, cout, KH, KW in SHAPES:
    b, h, w = (B, H, W) if "@" not in name else (B, 184, 248)
    x = torch.randn(b, h * w, cin, device=dev); dy = torch.randn(b, h * w, cout, device=dev) * 1e-3
    acc = torch.zeros(cout, KH, KW, cin, device=dev); db = torch.zeros(cout, device=dev)
    geom = (b, h, w, KH // 2, KW // 2)
    t_old = timed(lambda: call("craft_conv2d_wgrad", x, cin, cin, dy, cout, cout, KH, KW, b, h, w, acc, db, None, 0, prec), a.reps)
    t_pack = timed(lambda: (AG.Packed(dy, prec, geom, colsum=db), AG.Packed(x, prec, geom)), a.reps)
    gp, xp = AG.Packed(dy, prec, geom), AG.Packed(x, prec, geom)
    t_gemm = timed(lambda: AG.wgrad_pk([(gp, xp)], KH, KW, acc), a.reps)
    t_g12 = timed(lambda: AG.wgrad_pk([(gp, xp)] * 12, KH, KW, acc), max(2, a.reps // 4)) / 12
    fl = 2.0 * b * h * w * cin * cout * KH * KW
    rows.append({"layer": name, "cin": cin, "cout": cout, "taps": KH * KW, "old_us": round(t_old, 1), "pack_us": round(t_pack, 1), "gemm_us": round(t_gemm, 1), "gemm12_us_per_call": round(t_g12, 1),
                 "new_us": round(t_pack + t_gemm, 1), "old_TF": round(fl / t_old / 1e6, 1), "gemm_TF": round(fl / t_gemm / 1e6, 1), "new_TF": round(fl / (t_pack + t_gemm) / 1e6, 1)})
    print(json.dumps(rows[-1]), flush=True)
