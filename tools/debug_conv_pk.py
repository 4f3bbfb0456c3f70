"""Developer check of craft_conv2d_pk against craft_conv2d_nhwc2 and a float64 convolution on a few shapes."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from craft_amd import hip, ops
from craft_amd.autograd import Packed
from craft_amd.hip import call, ACT_NONE, W_PACKED, PREC_F16X3
dev = torch.device("cuda")
cp = PREC_F16X3
for (KH, KW, cin, cout, pad, B, H8, W8) in [(1, 5, 32, 256, (2, 2), 4, 56, 128), (1, 5, 32, 256, (0, 2), 4, 56, 128), (1, 5, 64, 256, (2, 2), 1, 16, 32), (1, 5, 32, 256, (2, 2), 1, 8, 16),
                                            (5, 1, 128, 126, (2, 0), 2, 11, 21), (5, 1, 128, 128, (2, 0), 1, 8, 16), (3, 3, 64, 192, (1, 1), 2, 11, 21), (1, 5, 128, 128, (0, 2), 1, 8, 16),
                                            (1, 5, 384, 256, (2, 2), 4, 56, 128), (3, 3, 256, 192, (2, 2), 4, 56, 128)]:
    torch.manual_seed(0)
    x = torch.randn(B, H8 * W8, cin, device=dev)
    w = torch.randn(cout, cin, KH, KW, device=dev) / math.sqrt(cin * KH * KW)
    wp = ops.pack_conv_weights(w, cp)
    zb = torch.zeros(cout, device=dev)
    y = torch.empty(B, H8 * W8, cout, device=dev)
    y2 = torch.full((B, H8 * W8, cout), float("nan"), device=dev)
    pk = Packed(x, cp, spatial=(B, H8, W8, pad[0], pad[1]))
    call("craft_conv2d_nhwc", x, cin, cin, wp, zb, cout, KH, KW, ACT_NONE, y, cout, B, H8, W8, cp | W_PACKED)
    call("craft_conv2d_pk", pk.buf, pk.rows_p, pk.C_p // 32, 0, cin, None, 0, 0, 0, 0, pk.guard, pad[0], pad[1], 0, wp, zb, None, 0, cout,
         KH, KW, ACT_NONE, y2, cout, B, H8, W8, cp | W_PACKED)
    torch.cuda.synchronize()
    xn = x.view(B, H8, W8, cin).permute(0, 3, 1, 2).double()
    ref = F.conv2d(xn, w.double(), None, padding=(KH // 2, KW // 2)).permute(0, 2, 3, 1).reshape(B, H8 * W8, cout)
    e1, e2 = (y.double() - ref).abs().max().item(), (y2.double() - ref).abs().max().item()
    d = (y - y2).abs()
    bad = (d > 0).nonzero()
    print(f"{KH}x{KW} {cin}->{cout} pad {pad} {B}x{H8}x{W8}: tokens err {e1:.2e}  packed err {e2:.2e}  max|diff| {d.max().item():.2e}  differing {bad.shape[0]} of {d.numel()}"
          + (f" first {bad[0].tolist()} last {bad[-1].tolist()}" if bad.shape[0] else ""))
