"""Does an MFMA-bound convolution hide under the HBM-bound P.V kernel when both are in flight on two HIP streams?  (round 5: before
splitting the GRU's z|r convolution into an [h | mf] part that could run beside k_pv16.)  usage: python tools/overlap_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from craft_amd import ops
from craft_amd.hip import call, ACT_NONE, W_PACKED, PREC_F16X3, PREC_F16, PROB_DTYPE

dev = torch.device("cuda")
B, H8, W8 = 4, 56, 128
N, M, Dv = H8 * W8, 4, 128
ldp = ops.round_up(N, 64)
P = ops.probs_tiled(torch.rand(B, M, N, ldp, device=dev).div_(N / 2).to(torch.float16))
vT = torch.randn(B, M * Dv, ops.round_up(N, 32), device=dev).to(torch.float16)
O = torch.empty(B, M, N, Dv, device=dev)
side = torch.cuda.Stream()


def pv():
    ops.attn_apply(P, vT, Dv, PREC_F16, out=O)


def conv_fn(cin, cout, KH=1, KW=5):
    x = torch.randn(B, N, cin, device=dev)
    w = torch.randn(cout, cin, KH, KW, device=dev) / (cin * KH * KW) ** 0.5
    wp = ops.pack_conv_weights(w, PREC_F16X3)
    zb = torch.zeros(cout, device=dev)
    y = torch.empty(B, N, cout, device=dev)
    return lambda: call("craft_conv2d_nhwc", x, cin, cin, wp, zb, cout, KH, KW, ACT_NONE, y, cout, B, H8, W8, PREC_F16X3 | W_PACKED)


def timeit(fn, reps=30):
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


for cin, cout in ((256, 256), (128, 768), (256, 768)):
    cv = conv_fn(cin, cout)

    def both(first_conv):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        if first_conv:
            with torch.cuda.stream(side):
                cv()
            pv()
        else:
            pv()
            with torch.cuda.stream(side):
                cv()
        main.wait_stream(side)
    t_pv, t_cv = timeit(pv), timeit(cv)
    t_a, t_b = timeit(lambda: both(True)), timeit(lambda: both(False))
    print(f"conv 1x5 {cin}->{cout}: pv16 {t_pv:.1f} us, conv {t_cv:.1f} us, serial {t_pv + t_cv:.1f}; two streams: conv enqueued first {t_a:.1f}, pv16 first {t_b:.1f}")
