"""Round 5: the halo convolution over plane-packed activations (craft_conv2d_pk: LDS-DMA halo, no in-loop conversion) beside the fp32-token
kernel (craft_conv2d_nhwc: k_conv_halo_wf) at the refinement loop's shapes with 32 .. 384 input channels: fixed cost per launch and
incremental MFMA rate of both (a least-squares line over the channel counts).  The packs are made outside the timed region (in the
product their producers write them).  usage: python tools/bench_conv_pk.py [f16x3|fp16|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from craft_amd import hip, ops
from craft_amd.autograd import Packed
from craft_amd.hip import call, ACT_NONE, W_PACKED

dev = torch.device("cuda")
cp = hip.PREC_NAMES[sys.argv[1]] if len(sys.argv) > 1 else hip.PREC_F16X3
nm = 3 if cp == hip.PREC_F16X3 else 1
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


CINS = (32, 64, 128, 256, 384)
for KH, KW, cout in ((1, 5, 256), (5, 1, 256), (1, 5, 128), (3, 3, 256), (3, 3, 192), (3, 3, 128), (3, 3, 64)):
    rows = {"tokens": [], "packed": []}
    for cin in CINS:
        x = torch.randn(B, N, cin, device=dev)
        w = torch.randn(cout, cin, KH, KW, device=dev) / (cin * KH * KW) ** 0.5
        wp = ops.pack_conv_weights(w, cp)
        zb = torch.zeros(cout, device=dev)
        y = torch.empty(B, N, cout, device=dev)
        y2 = torch.empty(B, N, cout, device=dev)
        pk = Packed(x, cp, spatial=(B, H8, W8, 2, 2))
        t0 = timeit(lambda: call("craft_conv2d_nhwc", x, cin, cin, wp, zb, cout, KH, KW, ACT_NONE, y, cout, B, H8, W8, cp | W_PACKED))
        t1 = timeit(lambda: call("craft_conv2d_pk", pk.buf, pk.rows_p, pk.C_p // 32, 0, cin, None, 0, 0, 0, 0, pk.guard, 2, 2, 0, wp, zb, None, 0, cout,
                                 KH, KW, ACT_NONE, y2, cout, B, H8, W8, cp | W_PACKED))
        if not (KH == 3 and cin == 64 and cout == 64):          # (that shape runs k_conv3x3_c64 on the token side: another summation order)
            assert torch.equal(y, y2), float((y - y2).abs().max())
        rows["tokens"].append(t0)
        rows["packed"].append(t1)
    for k, ts in rows.items():
        slope, icpt = np.polyfit(np.array(CINS, float), np.array(ts), 1)
        pf = 2.0 * B * N * cout * KH * KW * nm / (slope * 1e-6) / 1e15           # executed PF/s of the incremental K loop
        print(f"{KH}x{KW} -> {cout:3d} ({B}x{H8}x{W8}) {k:6s}: " + " | ".join(f"cin {c:3d}: {t:6.1f} us" for c, t in zip(CINS, ts))
              + f"  => {icpt:5.1f} us fixed + {slope:.4f} us/channel = {pf:.3f} PF/s executed")
