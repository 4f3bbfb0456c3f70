"""Developer check: k_conv3x3_c64 with the permuted GEMM-row -> pixel map (round 5) against the linear one (build_variant c64lin): the convolution
outputs must be bit-identical; only the fp32 summation order of the per-(image, channel) statistics changes.  Run once per library:
    CRAFT_HIP_LIB=... python tools/c64_perm_check.py out.pt ; then python tools/c64_perm_check.py --cmp a.pt b.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if sys.argv[1] == "--cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        d = (a[k].double() - b[k].double()).abs()
        print(f"{k:12s} equal={torch.equal(a[k], b[k])}  max|diff| {d.max().item():.3e}  max rel {(d / a[k].double().abs().clamp_min(1e-30)).max().item():.3e}")
    sys.exit(0)
from craft_amd import ops
from craft_amd.hip import call, ACT_NONE, W_PACKED, STATS_REPLICAS
dev = torch.device("cuda")
out = {}
for prec in (3, 2):
    torch.manual_seed(0)
    B, H, W = 4, 64, 96
    x = torch.randn(B, H * W, 64, device=dev)
    w = torch.randn(64, 64, 3, 3, device=dev) / 24
    b = torch.randn(64, device=dev)
    wp = ops.pack_conv_prec(w, prec)
    y = torch.empty(B, H * W, 64, device=dev)
    stats = torch.zeros(STATS_REPLICAS, B, 64, 2, device=dev, dtype=torch.float64)
    call("craft_conv2d_nhwc_ex", x, 64, 64, H, W, None, wp, b, 64, 3, 3, 1, ACT_NONE, y, 64, B, H, W, stats, prec | W_PACKED)
    torch.cuda.synchronize()
    out[f"y_prec{prec}"] = y.cpu()
    out[f"stats_prec{prec}"] = stats.sum(0).cpu()
torch.save(out, sys.argv[1])
