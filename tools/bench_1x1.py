"""1x1 convolutions / small-K products on fp32 activations: the fp32-source engines (craft_gemm -> k_gemm_gen, craft_linear -> k_gemm_rows)
against the halo-convolution kernel with KH = KW = 1 and fragment-order weights (craft_conv2d_nhwc | W_PACKED: weights never touch LDS).
usage: python tools/bench_1x1.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from craft_amd import hip, ops
from craft_amd import autograd as AG
from craft_amd.hip import call, PREC_F16X3, ACT_NONE, ACT_RELU, W_PACKED

dev = torch.device("cuda")
cp = PREC_F16X3


def timeit(fn, reps=50):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


# (name, B, H8, W8, cin, cout, w16)
CASES = [("train d_mh  576->256", 8, 46, 62, 576, 256, True), ("train d_corr 256->324", 8, 46, 62, 256, 324, True),
         ("train d_mf3 512->128", 8, 46, 62, 512, 128, True), ("infer convc1 352->256 relu", 4, 56, 128, 352, 256, False),
         ("infer linear 128->512", 4, 56, 128, 128, 512, False)]
for name, B, H8, W8, cin, cout, w16 in CASES:
    N = H8 * W8
    rows = B * N
    x = torch.randn(B, N, cin, device=dev)
    w = torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5
    cout_p = ops.round_up(cout, 32)
    # (a) craft_gemm on fp32 sources (what the backward's 1x1 input gradients run)
    y0 = torch.empty(B, N, cout, device=dev)
    w2 = w.view(cout, cin).contiguous()
    t_gemm = timeit(lambda: AG.gemm(x, cin, 1, 0, 0, w2, cin, 1, 0, 0, y0, cout, 0, 0, 1, 1, rows, cout, cin, prec=cp))
    # (b) craft_linear (k_gemm_rows)
    y1 = torch.empty(B, N, cout, device=dev)
    t_lin = timeit(lambda: ops.linear(x, w2, None, cp, out=y1)) if cout % 4 == 0 else float("nan")
    # (c) the halo kernel with 1 tap and fragment-order weights
    wp = ops.pack_conv_weights(w, cp)
    zb = torch.zeros(cout_p, device=dev)
    y2 = torch.empty(B, N, cout_p, device=dev)
    flag = W_PACKED | (hip.CONV_W16 if w16 else 0)
    t_wf = timeit(lambda: call("craft_conv2d_nhwc", x, cin, cin, wp, zb, cout_p, 1, 1, ACT_NONE, y2, cout_p, B, H8, W8, cp | flag))
    err = ((y2[..., :cout] - y0).norm() / y0.norm()).item()
    print(f"{name:28s} rows {rows:6d}: craft_gemm {t_gemm:6.1f} us | craft_linear {t_lin:6.1f} us | halo 1x1 packed{' w16' if w16 else ''} {t_wf:6.1f} us   (rel diff {err:.1e})")
