"""craft_attn_softmax_fwd / _bwd at BASELINE configs[3]'s attention shape (8 x 4 modes x 2852 x 2852): time and HBM rate against the bytes
each kernel has to move (diagnostics).  usage: python tools/bench_softmax.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from craft_amd import autograd as AG, hip
from craft_amd.hip import round_up

dev = torch.device("cuda")
B, M, H8, W8 = 8, 4, 46, 62
N = H8 * W8
ld = round_up(N, 32)
S0 = torch.randn(B, M, N, ld, device=dev) * 2.0
tab = torch.randn(15, 15, device=dev) * 0.5
G = torch.randn(B, M, N, ld, device=dev)
bits = torch.empty(B * M * N * (ld // 32), device=dev, dtype=torch.int32)
gb = S0.numel() * 4 / 1e9


def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


S = S0.clone()
pk = AG.PkMat(B * M, N, ld, hip.PREC_F16X3, dev)
Pd = torch.empty_like(S)
for name, fn, nb in (
    ("fwd plain (in place)", lambda: AG.call("craft_attn_softmax_fwd", S, ld, B, M, H8, W8, tab, 7, 0.5, -1, None, bits, None, 0.0, 0, None, 0, 0, 0), 2),
    ("fwd + dropout -> fp32 copy", lambda: AG.call("craft_attn_softmax_fwd", S, ld, B, M, H8, W8, tab, 7, 0.5, -1, None, bits, Pd, 0.2, 5, None, 0, 0, 0), 3),
    ("fwd + dropout -> packed f16x3", lambda: AG.call("craft_attn_softmax_fwd", S, ld, B, M, H8, W8, tab, 7, 0.5, -1, None, bits, None, 0.2, 5, pk.buf, pk.rows_total, pk.np_, pk.prec), 3),
):
    t = timeit(fn)
    print(f"{name:34s} {t:8.1f} us   {nb * gb / t * 1e3:5.2f} TB/s of the {nb} x {gb:.2f} GB it has to move")
P = torch.softmax(S0[..., :N], dim=-1)
Pp = torch.zeros_like(S0); Pp[..., :N] = P
rep = torch.zeros(hip.STATS_REPLICAS, 225, device=dev)
dS = G.clone()
for name, fn, nb in (
    ("bwd (dS over dP)", lambda: AG.call("craft_attn_softmax_bwd", Pp, dS, ld, B, M, H8, W8, 7, 0.5, None, bits, rep, 0.0, 0, None, 0, 0, 0), 3),
    ("bwd + dropout mask", lambda: AG.call("craft_attn_softmax_bwd", Pp, dS, ld, B, M, H8, W8, 7, 0.5, None, bits, rep, 0.2, 5, None, 0, 0, 0), 3),
    ("bwd + dropout mask -> packed f16x3 dS", lambda: AG.call("craft_attn_softmax_bwd", Pp, dS, ld, B, M, H8, W8, 7, 0.5, None, bits, rep, 0.2, 5, pk.buf, pk.rows_total, pk.np_, pk.prec), 3),
    ("bwd, no positional table", lambda: AG.call("craft_attn_softmax_bwd", Pp, dS, ld, B, M, H8, W8, 0, 0.0, None, bits, None, 0.2, 5, None, 0, 0, 0), 3),
):
    t = timeit(fn)
    print(f"{name:34s} {t:8.1f} us   {nb * gb / t * 1e3:5.2f} TB/s of the {nb} x {gb:.2f} GB it has to move")
