"""Attention products of the training pass at BASELINE configs[3]'s shape on craft_gemm_pk vs the fp32-source engine craft_gemm
(diagnostics).  usage: python tools/bench_gemm_pkb.py [prec]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from craft_amd import autograd as AG, hip
from craft_amd.hip import round_up

prec = {"f16x3": hip.PREC_F16X3, "f16": hip.PREC_F16, "bf16": hip.PREC_BF16}[sys.argv[1] if len(sys.argv) > 1 else "f16x3"]
dev = torch.device("cuda")
B, Mh, N, C, T = 8, 4, 46 * 62, 128, 12
ld = round_up(N, 32)
P = torch.zeros(B, Mh, N, ld, device=dev)
P[..., :N] = torch.softmax(torch.randn(B, Mh, N, N, device=dev), dim=-1)
V = torch.randn(B, N, Mh * C, device=dev)
dO = torch.randn(B, Mh, N, T * C, device=dev)


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


Ppk = AG.PkMat(B * Mh, N, ld, prec, dev).fill(P)
Vpk = AG.PkMat(B, N, Mh * T * C, prec, dev)                 # channels (m, t, c)
for m in range(Mh):
    for t in range(T):
        Vpk.fill(V[..., m * C:(m + 1) * C], cg_off=(m * T + t) * (C // 32))
dOpk = AG.PkMat(B * Mh, N, T * C, prec, dev).fill(dO)
cg = C // 32
O = torch.empty(B, Mh, N, C, device=dev)
O2 = torch.empty_like(O)
planes = 3 if prec == hip.PREC_F16X3 else 1
fl = 2.0 * B * Mh * N * N * C
t = timeit(lambda: AG.gemm_pk(Ppk, Ppk.desc(AG.PK_CH, Mh, 1), Vpk, Vpk.desc(AG.PK_ROWS, 1, 0, 0, 0, T * cg), O, C, Mh * N * C, N * C, Mh, B * Mh, N, C, ld))
t0 = timeit(lambda: AG.gemm(P, ld, 1, Mh * N * ld, N * ld, V, 1, Mh * C, N * Mh * C, C, O2, C, Mh * N * C, N * C, Mh, B * Mh, N, C, N, prec=prec))
print(f"O = P V      pk {t:8.1f} us ({P.numel() * 4 / t / 1e6:5.2f} TB/s of P, {planes * fl / t / 1e6:6.1f} TF MFMA)   craft_gemm {t0:8.1f} us   max diff {float((O - O2).abs().max()):.2e}")
TC = T * C
dV = torch.empty(B, N, Mh, TC, device=dev)
dV2 = torch.empty_like(dV)
t = timeit(lambda: AG.gemm_pk(Ppk, Ppk.desc(AG.PK_ROWS, Mh, 1), dOpk, dOpk.desc(AG.PK_ROWS, Mh, 1), dV, Mh * TC, N * Mh * TC, TC, Mh, B * Mh, N, TC, N), 5)
t0 = timeit(lambda: AG.gemm(P, 1, ld, Mh * N * ld, N * ld, dO, 1, TC, Mh * N * TC, N * TC, dV2, Mh * TC, N * Mh * TC, TC, Mh, B * Mh, N, TC, N, prec=prec), 5)
print(f"dV = P^T dO  pk {t:8.1f} us ({planes * fl * T / t / 1e6:6.1f} TF MFMA)   craft_gemm {t0:8.1f} us   rel diff {float((dV - dV2).norm() / dV2.norm()):.2e}")
dP = torch.empty(B, Mh, N, ld, device=dev)
dP2 = torch.zeros_like(dP)
Vcat = V.view(B, N, Mh, 1, C).expand(B, N, Mh, T, C).permute(0, 2, 1, 3, 4).reshape(B, Mh, N, TC).contiguous()
t = timeit(lambda: AG.gemm_pk(dOpk, dOpk.desc(AG.PK_CH, Mh, 1), Vpk, Vpk.desc(AG.PK_CH, 1, 0, 0, 0, T * cg), dP, ld, Mh * N * ld, N * ld, Mh, B * Mh, N, N, TC), 5)
t0 = timeit(lambda: AG.gemm(dO, TC, 1, Mh * N * TC, N * TC, Vcat, TC, 1, Mh * N * TC, N * TC, dP2, ld, Mh * N * ld, N * ld, Mh, B * Mh, N, N, TC, prec=prec), 5)
print(f"dP = dO V^T  pk {t:8.1f} us ({planes * fl * T / t / 1e6:6.1f} TF MFMA)   craft_gemm {t0:8.1f} us   rel diff {float((dP[..., :N] - dP2[..., :N]).norm() / dP2[..., :N].norm()):.2e}")
tp = timeit(lambda: AG.PkMat(B * Mh, N, T * C, prec, dev).fill(dO), 5)
print(f"pack dO_cat ({dO.numel() * 4 / 1e6:.0f} MB): {tp:.1f} us")
S = torch.randn(B, Mh, N, ld, device=dev)
pk = AG.PkMat(B * Mh, N, ld, prec, dev)
Pd = torch.empty_like(S)
t = timeit(lambda: AG.call("craft_attn_softmax_fwd", S, ld, B, Mh, 46, 62, None, 0, 0.0, -1, None, None, None, 0.2, 5, pk.buf, pk.rows_total, pk.np_, prec), 5)
t0 = timeit(lambda: AG.call("craft_attn_softmax_fwd", S, ld, B, Mh, 46, 62, None, 0, 0.0, -1, None, None, Pd, 0.2, 5, None, 0, 0, 0), 5)
print(f"softmax fwd + dropout: packed output {t:.1f} us, fp32 Pdrop output {t0:.1f} us")
# ---- the scores' three products (d = 64 per mode): S = scale Q K^T, dQ = dS K, dK = dS^T Q
d_ = 64
Cq = Mh * d_
q = torch.randn(B, N, Cq, device=dev)
k = torch.randn(B, N, Cq, device=dev)
qpk, kpk = AG.PkMat(B, N, Cq, prec, dev).fill(q), AG.PkMat(B, N, Cq, prec, dev).fill(k)
Sx, Sy = torch.empty(B, Mh, N, ld, device=dev), torch.empty(B, Mh, N, ld, device=dev)
cgq = d_ // 32
t = timeit(lambda: AG.gemm_pk(qpk, qpk.desc(AG.PK_CH, 1, 0, 0, 0, cgq), kpk, kpk.desc(AG.PK_CH, 1, 0, 0, 0, cgq), Sx, ld, Mh * N * ld, N * ld, Mh, B * Mh, N, N, d_, alpha=0.125))
t0 = timeit(lambda: AG.gemm(q, Cq, 1, N * Cq, d_, k, Cq, 1, N * Cq, d_, Sy, ld, Mh * N * ld, N * ld, Mh, B * Mh, N, N, d_, alpha=0.125, prec=prec))
print(f"S = Q K^T    pk {t:8.1f} us ({Sx.numel() * 4 / t / 1e6:5.2f} TB/s of S written)   craft_gemm {t0:8.1f} us   rel diff {float((Sx[..., :N] - Sy[..., :N]).norm() / Sy[..., :N].norm()):.2e}")
dS = torch.zeros(B, Mh, N, ld, device=dev)
dS[..., :N] = torch.randn(B, Mh, N, N, device=dev)
dSpk = AG.PkMat(B * Mh, N, ld, prec, dev).fill(dS)
dq, dq2 = torch.empty(B, N, Cq, device=dev), torch.empty(B, N, Cq, device=dev)
t = timeit(lambda: AG.gemm_pk(dSpk, dSpk.desc(AG.PK_CH, Mh, 1), kpk, kpk.desc(AG.PK_ROWS, 1, 0, 0, 0, cgq), dq, Cq, N * Cq, d_, Mh, B * Mh, N, d_, N, alpha=0.125))
t0 = timeit(lambda: AG.gemm(dS, ld, 1, Mh * N * ld, N * ld, k, 1, Cq, N * Cq, d_, dq2, Cq, N * Cq, d_, Mh, B * Mh, N, d_, N, alpha=0.125, prec=prec))
print(f"dQ = dS K    pk {t:8.1f} us   craft_gemm {t0:8.1f} us   rel diff {float((dq - dq2).norm() / dq2.norm()):.2e}")
t = timeit(lambda: AG.gemm_pk(dSpk, dSpk.desc(AG.PK_ROWS, Mh, 1), qpk, qpk.desc(AG.PK_ROWS, 1, 0, 0, 0, cgq), dq, Cq, N * Cq, d_, Mh, B * Mh, N, d_, N, alpha=0.125))
t0 = timeit(lambda: AG.gemm(dS, 1, ld, Mh * N * ld, N * ld, q, 1, Cq, N * Cq, d_, dq2, Cq, N * Cq, d_, Mh, B * Mh, N, d_, N, alpha=0.125, prec=prec))
print(f"dK = dS^T Q  pk {t:8.1f} us   craft_gemm {t0:8.1f} us   rel diff {float((dq - dq2).norm() / dq2.norm()):.2e}")
