"""Fixed cost of a halo-convolution launch: the same 1x5 / 3x3 convolution at the refinement loop's shape with 32 .. 384 input channels
(the K loop shrinks, prologue / epilogue / launch stay).  usage: python tools/bench_conv_fixed.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from craft_amd import hip, ops
from craft_amd.hip import call, PREC_F16X3, ACT_NONE, W_PACKED

dev = torch.device("cuda")
cp = PREC_F16X3
B, H8, W8 = int(os.environ.get("B", 4)), int(os.environ.get("H8", 56)), int(os.environ.get("W8", 128))
N = H8 * W8


def timeit(fn, reps=100):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for KH, KW, cout in ((1, 5, 256), (1, 5, 128), (3, 3, 256), (3, 3, 128)):
    row = []
    for cin in (32, 64, 128, 256, 384):
        x = torch.randn(B, N, cin, device=dev)
        w = torch.randn(cout, cin, KH, KW, device=dev) / (cin * KH * KW) ** 0.5
        wp = ops.pack_conv_weights(w, cp)
        zb = torch.zeros(cout, device=dev)
        y = torch.empty(B, N, cout, device=dev)
        t = timeit(lambda: call("craft_conv2d_nhwc", x, cin, cin, wp, zb, cout, KH, KW, ACT_NONE, y, cout, B, H8, W8, cp | W_PACKED))
        row.append(f"cin {cin:3d}: {t:6.1f} us")
    print(f"{KH}x{KW} -> {cout:3d} ({B}x{H8}x{W8}): " + " | ".join(row))
