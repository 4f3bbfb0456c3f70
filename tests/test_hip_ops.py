"""Per-kernel parity tests: every C-ABI entry point against the CPU oracle on seeded inputs.

All tests need the GPU (``-m gpu``) and go through ``libcraft_hip.so``.  fp32 mode must match the
oracle to fp32 rounding (rtol 1e-4 / atol 1e-5 on O(1) tensors); the 16-bit MFMA modes
are checked against the same oracle with the tolerance their operand rounding implies
(bf16: 2^-9 relative per operand, fp16: 2^-12), stated per test.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from craft_amd import hip, ops
from craft_amd.hip import ACT_NONE, ACT_RELU, ACT_TANH, PREC_BF16, PREC_F16, PREC_F16X3, PREC_F32
from oracle import craft_oracle as O

pytestmark = pytest.mark.gpu

PRECS = [PREC_F32, PREC_BF16, PREC_F16, PREC_F16X3]
# (rtol, atol multiplier) for a K-long dot product of O(1) operands.  F16X3 (split-fp16 emulation of fp32
# products, 2^-22 per product) is held to the fp32 tolerance.
TOL = {PREC_F32: (1e-4, 1e-5), PREC_BF16: (2e-2, 2e-2), PREC_F16: (3e-3, 3e-3), PREC_F16X3: (1e-4, 1e-5)}


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(got, ref, rtol, atol, what):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{what}: non-finite values in HIP output"
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any():
        i = int(torch.argmax(err - tol))
        idx = np.unravel_index(i, got.shape)
        raise AssertionError(f"{what}: {int(bad.sum())}/{got.numel()} elements out of tolerance; worst at {idx}: "
                             f"got {got.flatten()[i]:.6f} ref {ref.flatten()[i]:.6f} |d|={err.flatten()[i]:.3e} "
                             f"(rtol={rtol}, atol={atol}; max|ref|={ref.abs().max():.3f})")


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("rows,cin,cout", [(2 * 301, 324, 256), (640, 256, 126), (130, 128, 64), (257, 64, 192), (96, 32, 576)])
def test_linear(device, prec, rows, cin, cout):
    """craft_linear: asymmetric operands so a transposed / permuted output can not pass."""
    x = gen(2, rows // 2, cin, seed=1)
    w = gen(cout, cin, seed=2) / math.sqrt(cin)
    w[3, 5] += 2.0                                  # break any accidental symmetry
    b = gen(cout, seed=3)
    ref = F.linear(x, w, b)
    y = ops.linear(x.to(device), w.to(device), b.to(device), prec)
    rt, at = TOL[prec]
    close(y, ref, rt, at, f"linear prec={prec}")
    y2 = ops.linear(x.to(device), w.to(device), None, prec)
    close(y2, F.linear(x, w), rt, at, f"linear(no bias) prec={prec}")


@pytest.mark.parametrize("prec", PRECS)
def test_linear_identity_weight_detects_transpose(device, prec):
    x = torch.arange(2 * 70 * 64, dtype=torch.float32).reshape(2, 70, 64) / 64.0 if prec == PREC_F32 else gen(2, 70, 64, seed=4)
    w = torch.eye(64)
    y = ops.linear(x.to(device), w.to(device), None, prec)
    rt, at = TOL[prec]
    close(y, x, rt, at, "identity projection")


@pytest.mark.parametrize("prec", PRECS)
def test_linear_t_and_column_views(device, prec):
    B, N, C, Cout = 2, 203, 128, 512
    buf = torch.zeros(B, N, 512)
    buf[..., 256:384] = gen(B, N, C, seed=5)
    w = gen(Cout, C, seed=6) / math.sqrt(C)
    ldt = ops.round_up(N, 32)
    yT = ops.linear_t(buf.to(device)[..., 256:384], w.to(device), ldt, hip.Precision(proj=prec, pv=PREC_F32))
    ref = F.linear(buf[..., 256:384], w).transpose(1, 2)
    rt, at = TOL[prec]
    close(yT[..., :N], ref, rt, at, "linear_t")
    y16 = ops.linear_t(buf.to(device)[..., 256:384], w.to(device), ldt, hip.Precision(proj=prec, pv=PREC_F16))
    assert y16.dtype == torch.float16
    close(y16[..., :N], ref, max(rt, 2e-3), max(at, 2e-3), "linear_t (fp16 output)")
    assert float(yT[..., N:].abs().max()) == 0.0
    # fragment order (the V^T operand of the 16-bit craft_attn_apply): a pure permutation of the plain layout
    Dv = 128
    yf = ops.linear_t(buf.to(device)[..., 256:384], w.to(device), ldt, hip.Precision(proj=prec, pv=PREC_F16), Dv=Dv)
    un = yf.view(B, Cout // Dv, ldt // 16, Dv // 32, 2, 32, 8).permute(0, 1, 3, 5, 2, 4, 6).reshape(B, Cout, ldt)
    # (the two layouts run the GEMM in opposite operand roles: same products, different fp32 summation order)
    close(un.float(), y16.float(), 2e-3, 2e-3, "fragment-order V^T vs row-major")
    close(un[..., :N], ref, max(rt, 2e-3), max(at, 2e-3), "linear_t (fragment order)")


@pytest.mark.parametrize("prec", [p for p in PRECS if p != PREC_F32])
@pytest.mark.parametrize("rows,cin,cout", [(406, 324, 256), (300, 128, 512), (77, 256, 256), (130, 36, 40), (128, 64, 96)])
def test_linear_with_packed_weights(device, prec, rows, cin, cout):
    """craft_linear / craft_linear_t with CRAFT_W_PACKED (k_gemm_rows_wf: weights in MFMA fragment order, K padded to 32) against the
    fp32 reference and against the same product with raw weights (same operand planes, another fp32 summation order); row views with
    a column offset, ragged row / column / K tails, bias."""
    B = 2
    buf = torch.zeros(B, rows // 2, cin + 24)
    buf[..., 8:8 + cin] = gen(B, rows // 2, cin, seed=31)
    w = gen(cout, cin, seed=32) / math.sqrt(cin)
    w[3, 5] += 2.0
    b = gen(cout, seed=33)
    xd = buf.to(device)[..., 8:8 + cin]
    pk = ops.pack_linear_weight(w.to(device), prec)
    y = ops.linear(xd, w.to(device), b.to(device), prec, packed=pk)
    y0 = ops.linear(xd, w.to(device), b.to(device), prec)
    ref = F.linear(buf[..., 8:8 + cin], w, b)
    rt, at = TOL[prec]
    close(y, ref, rt, at, f"packed linear prec={prec}")
    close(y, y0, 2e-5 if prec == PREC_F16X3 else rt, 2e-5 if prec == PREC_F16X3 else at, "packed vs raw weights")
    if cout % 128 == 0:                                  # the V^T projection of the aggregators (fragment order, both key orders)
        N = rows // 2
        ldt = ops.round_up(N, 32)
        P = hip.Precision(proj=prec, pv=PREC_F16)
        for acc in (False, True):
            a = ops.linear_t(xd, w.to(device), ldt, P, Dv=128, acc_order=acc, packed=pk)
            a0 = ops.linear_t(xd, w.to(device), ldt, P, Dv=128, acc_order=acc)
            assert a.dtype == torch.float16 and a.shape == a0.shape
            close(a.float(), a0.float(), 2e-3, 2e-3, f"packed V^T (acc_order={acc})")
            # (N a multiple of 32 or not: the tail keys must read as zero in both)
            assert int((a == 0).sum()) >= int(B * cout * (ldt - N))


MIXED = hip.Precision.parse("mixed")


def test_linear_pack_cache_follows_the_parameter(device):
    """ops.linear_pack: one packed copy per (module, name, precision); an in-place update or a replaced parameter re-packs; fp32 and
    autograd-tracked calls get None (the raw-weight path)."""
    lin = torch.nn.Linear(96, 64).to(device)
    with torch.no_grad():
        a = ops.linear_pack(lin, "w", lin.weight, MIXED)
        assert a is not None and ops.linear_pack(lin, "w", lin.weight, MIXED) is a
        x = gen(1, 50, 96, seed=34).to(device)
        y1 = ops.linear(x, lin.weight, lin.bias, MIXED, packed=a)
        lin.weight.mul_(2.0)
        b = ops.linear_pack(lin, "w", lin.weight, MIXED)
        assert b is not a
        y2 = ops.linear(x, lin.weight, lin.bias, MIXED, packed=b)
        close(y2 - lin.bias, 2 * (y1 - lin.bias), 1e-5, 1e-5, "re-packed after an in-place update")
        assert ops.linear_pack(lin, "w", lin.weight, PREC_F32) is None
    assert ops.linear_pack(lin, "w", lin.weight, MIXED) is None      # grad mode, requires_grad


@pytest.mark.parametrize("nchw", [True, False])
def test_tokens(device, nchw):
    B, C, H, W = 2, 256, 9, 13
    x = gen(B, 300, H, W, seed=7) * 3 + 0.5
    if nchw:
        t = ops.tokens_from_nchw(x.to(device), c_off=20, C=C, act=ACT_RELU, ln=True)
        ref = O.layernorm_lastdim(torch.relu(x[:, 20:20 + C]).reshape(B, C, H * W).transpose(1, 2))
    else:
        xt = x[:, :C].reshape(B, C, H * W).transpose(1, 2).contiguous()
        t = ops.tokens_norm(xt.to(device))
        ref = O.layernorm_lastdim(xt)
    close(t, ref, 1e-4, 1e-5, "tokens LN")
    t2 = ops.tokens_from_nchw(x.to(device), c_off=0, C=128, act=ACT_TANH, ln=False)
    close(t2, torch.tanh(x[:, :128]).reshape(B, 128, H * W).transpose(1, 2), 1e-5, 1e-6, "tokens tanh")
    back = ops.tokens_to_nchw(t2, H, W)
    close(back, torch.tanh(x[:, :128]), 1e-5, 1e-6, "tokens_to_nchw")


def _qk(B, H8, W8, C, seed, gain=2.5):
    N = H8 * W8
    x1 = O.layernorm_lastdim(gen(B, N, C, seed=seed))
    x2 = O.layernorm_lastdim(gen(B, N, C, seed=seed + 1) + 0.5 * x1)
    Wq = gen(C, C, seed=seed + 2) * math.sqrt(gain / C)
    Wk = gen(C, C, seed=seed + 3) * math.sqrt(gain / C)
    bq = gen(C, seed=seed + 4) * 0.3
    return x1, x2, Wq, Wk, bq


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("gain", [2.5, 60.0])
@pytest.mark.parametrize("tiled,H8,W8", [(False, 18, 21), (True, 18, 21), (True, 16, 48), (True, 9, 37)])
def test_corr_build_pyramid_lookup(device, prec, gain, tiled, H8, W8):
    """craft_score_max + craft_corr_build + craft_corr_finish + craft_corr_lookup vs the oracle
    (inter-frame attention, tied projection with bias; gain=60 triggers the global clamp).  ``tiled``: the pyramid in the fused
    build's tiled layout (CRAFT_PYR_TILED: 8x16 / 4x8 tiles of levels 0 / 1; only the f16x3 build writes it) -- ragged tile grids
    (18x21, 9x37) and exact ones (16x48)."""
    if tiled and not ops.fused_pyramid(256, 4, prec, 4, H8, W8):
        pytest.skip("the tiled layout is written by the fused f16x3 build only")
    B, C, M = 2, 256, 4                       # odd sizes: ragged tiles, floor pooling
    N = H8 * W8
    x1, x2, Wq, _, bq = _qk(B, H8, W8, C, seed=10, gain=gain)
    tab = gen(15, 15, seed=20) * 0.5
    w_aggr = 0.8
    sd = {"corr_fn.setrans.query.weight": Wq, "corr_fn.setrans.query.bias": bq,
          "corr_fn.vispos_encoder.pos_coder.biases": tab,
          "corr_fn.setrans.attn_softaggr.feat2score.weight": torch.tensor([[w_aggr]])}
    S = O.mm_scores(x1, x2, Wq, bq, Wq, bq, M)
    if gain > 10:
        assert float(S.max()) > 100, "test input does not trigger the clamp"
    c_ref = O.softaggr_scores(O.clamp_rule(S) + 0.5 * O.pos_bias_matrix(tab, H8, W8), sd["corr_fn.setrans.attn_softaggr.feat2score.weight"])
    mu_ref, rstd_ref = O.global_stats(c_ref)
    pyr_ref = O.build_pyramid(c_ref, H8, W8, 4)

    q = ops.linear(x1.to(device), Wq.to(device), bq.to(device), PREC_F32)
    k = ops.linear(x2.to(device), Wq.to(device), bq.to(device), PREC_F32)
    scale = 1.0 / math.sqrt(C // M)
    mx = ops.score_max(q, k, H8, W8, M, scale, prec)
    rt, at = TOL[prec]
    got_max = ops.decode_ord(mx)
    if got_max != got_max:      # NaN: the Cauchy-Schwarz norm bound proved that no score reaches the clip threshold
        assert float(S.max()) <= 100.0, "norm bound skipped the exact pass although a score exceeds the threshold"
    else:
        assert abs(got_max - float(S.max())) <= at * 20 + rt * abs(float(S.max())), f"score max {got_max} vs {float(S.max())}"
    pyr = ops.CorrPyramid(B, H8, W8, 4, device, tiled=tiled)
    assert pyr.tiled == tiled
    ops.corr_build(q, k, H8, W8, M, scale, tab.to(device), 0.5, w_aggr, mx, pyr, True, prec)
    sc = float(c_ref.abs().max())
    close(pyr.dense(0).reshape(B, N, N), c_ref, rt, at * max(1.0, sc), f"corr level 0 prec={prec}")
    for l in range(1, 4):
        close(pyr.dense(l), pyr_ref[l][:, 0], rt, at * max(1.0, sc), f"corr level {l}")
    close(pyr.mu_rstd[:, 0], mu_ref, rt, at * max(1.0, sc), "global mean")
    close(pyr.mu_rstd[:, 1], rstd_ref, max(rt, 1e-4), 1e-6, "global rstd")

    # lookups: identity grid, smooth offsets, far out-of-bounds
    c0 = O.coords_grid(B, H8, W8)
    wild = c0 + gen(B, 2, H8, W8, seed=30) * torch.tensor([W8 / 2.0, H8 / 2.0]).view(1, 2, 1, 1)
    frac = c0 + gen(B, 2, H8, W8, seed=31) * 2.0
    for name, cc in (("identity", c0), ("frac", frac), ("wild", wild)):
        ref = O.corr_lookup(pyr_ref, cc, 4, mu_ref, rstd_ref)
        got = ops.corr_lookup(pyr, ops.tokens_from_nchw(cc.to(device)), 4)
        if prec in (PREC_F32, PREC_F16X3):
            close(ops.tokens_to_nchw(got, H8, W8), ref, 2e-4, 2e-4, f"lookup {name}")
        else:
            close(ops.tokens_to_nchw(got, H8, W8), ref, rt, at * 10, f"lookup {name} prec={prec}")


@pytest.mark.parametrize("prec", PRECS)
def test_plain_corr(device, prec):
    """CorrBlock.corr variant: one mode over all 256 channels, no bias, no norm."""
    B, H8, W8, C = 1, 16, 20, 256
    f1, f2 = gen(B, C, H8, W8, seed=40), gen(B, C, H8, W8, seed=41)
    ref = O.plain_corr_raw(f1, f2)
    pyr = ops.CorrPyramid(B, H8, W8, 4, device)
    ops.corr_build(ops.tokens_from_nchw(f1.to(device)), ops.tokens_from_nchw(f2.to(device)), H8, W8, 1, 1.0 / 16.0, None, 0.0,
                   1.0, None, pyr, False, prec)
    rt, at = TOL[prec]
    close(pyr.lv[0].reshape(B, H8 * W8, H8 * W8), ref, rt, at * 4, "plain corr")
    assert pyr.mu_rstd.cpu().tolist() == [[0.0, 1.0]]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("C,M,mask_radius,gain", [(128, 4, -1, 2.5), (256, 4, 5, 2.5), (128, 4, -1, 80.0), (128, 1, -1, 2.5)])
def test_attn_probs(device, prec, C, M, mask_radius, gain):
    B, H8, W8 = 2, 13, 19
    N = H8 * W8
    x, _, Wq, Wk, _ = _qk(B, H8, W8, C, seed=50, gain=gain)
    tab = gen(15, 15, seed=51) * 0.5
    ref = O.self_attn_probs(x, Wq, Wk, tab, 1.0, M, H8, W8, mask_radius)
    q = ops.linear(x.to(device), Wq.to(device), None, PREC_F32)
    k = ops.linear(x.to(device), Wk.to(device), None, PREC_F32)
    scale = 1.0 / math.sqrt(C // M)
    mx = ops.score_max(q, k, H8, W8, M, scale, prec)
    if prec == PREC_F16X3:
        prec = hip.Precision(score=PREC_F16X3, pv=PREC_F32)
    P = ops.attn_probs(q, k, H8, W8, M, scale, tab.to(device), 1.0, mask_radius, mx, prec)
    if prec == PREC_BF16:      # mixed roles: fp32 logits, fp16 storage
        P2 = ops.attn_probs(q, k, H8, W8, M, scale, tab.to(device), 1.0, mask_radius, mx, hip.Precision(score=PREC_F32, pv=PREC_F16))
        assert P2.dtype == torch.float16
        close(P2[..., :H8 * W8].float(), ref, 2e-3, 1e-6, "fp32 logits stored as fp16")
    assert P.shape == (B, M, N, ops.round_up(N, 32))
    assert float(P[..., N:].float().abs().max()) == 0.0, "padding columns must be zero"
    got = P[..., :N].float()
    rsum = got.sum(-1).cpu()
    exact = prec == PREC_F32 or isinstance(prec, hip.Precision)
    assert (rsum - 1).abs().max() < (1e-5 if exact else 1e-2), "rows must sum to 1"
    if exact:
        close(got, ref, 2e-4, 1e-6, "attention probs")
    else:
        # logits carry the operand rounding (|S| up to ~|x||y| 2^-8), probabilities are <= 1
        close(got, ref, 0.0, 0.08 if prec == PREC_BF16 else 0.02, f"attention probs prec={prec}")
    # deferred normalisation: exp(logit - rowmax) in (0, 1] plus row sums; P' / rowsum is the same softmax
    Pd = ops.attn_probs(q, k, H8, W8, M, scale, tab.to(device), 1.0, mask_radius, mx, prec, defer=True, tiled=False)
    rs = Pd.craft_rowsum
    assert rs.shape == (B, M, N) and float(Pd.float().max()) <= 1.0 and float(Pd[..., :N].float().amax(-1).min()) > 0.99
    store_tol = 1e-6 if Pd.dtype == torch.float32 else (8e-3 if Pd.dtype == torch.bfloat16 else 1e-3)
    close(Pd[..., :N].float() / rs[..., None], got, store_tol, store_tol, "deferred vs normalised probabilities")
    assert float(Pd[..., N:].float().abs().max()) == 0.0


@pytest.mark.parametrize("H8,W8", [(13, 19), (16, 64), (24, 128), (9, 70), (5, 32)])
@pytest.mark.parametrize("mask_radius,gain", [(-1, 2.5), (5, 2.5), (-1, 80.0)])
@pytest.mark.parametrize("score", ["f16x3", "fp16"])
def test_attn_probs_fused_single_launch(device, monkeypatch, H8, W8, mask_radius, gain, score):
    """craft_attn_probs_fused (one launch of independent waves: exact row maxima, then P' + row sums; keys pre-split in MFMA fragment
    order) against the oracle's softmax (setrans.py:507-557) and against the two-launch path it replaces: ragged N, rows that straddle
    64-key tiles (W8 = 19 / 70), tiles inside one row (W8 = 64 / 128), Chebyshev mask, the clamp-active case (gain 80)."""
    B, C, M = 2, 128, 4
    N = H8 * W8
    x, _, Wq, Wk, _ = _qk(B, H8, W8, C, seed=52, gain=gain)
    tab = gen(15, 15, seed=53) * 0.5
    ref = O.self_attn_probs(x, Wq, Wk, tab, 1.0, M, H8, W8, mask_radius)
    q = ops.linear(x.to(device), Wq.to(device), None, PREC_F32)
    k = ops.linear(x.to(device), Wk.to(device), None, PREC_F32)
    scale = 1.0 / math.sqrt(C // M)
    sp = PREC_F16X3 if score == "f16x3" else PREC_F16
    prec = hip.Precision(score=sp, pv=PREC_F16)
    mx = ops.score_max(q, k, H8, W8, M, scale, prec)
    calls = []
    orig = ops.call
    monkeypatch.setattr(ops, "call", lambda name, *a: (calls.append(name), orig(name, *a))[1])
    Pt = ops.attn_probs(q, k, H8, W8, M, scale, tab.to(device), 1.0, mask_radius, mx, prec, defer=True)       # default layout: tiled
    Pd = ops.attn_probs(q, k, H8, W8, M, scale, tab.to(device), 1.0, mask_radius, mx, prec, defer=True, tiled=False)
    assert calls == ["craft_attn_probs_fused"] * 2
    # CRAFT_P_TILED: the same values in 32-query x 64-key tiles, zero key padding up to a multiple of 64
    assert Pt.craft_tiled and Pt.craft_n == N and tuple(Pt.shape) == (B, M, ops.round_up(N, 32), ops.round_up(N, 64))
    assert torch.equal(ops.probs_rowmajor(Pt), Pd[..., :N]) and torch.equal(Pt.craft_rowsum, Pd.craft_rowsum)
    unt = Pt.view(B, M, -1, Pt.shape[-1] // 64, 32, 64).permute(0, 1, 2, 4, 3, 5).reshape(B, M, Pt.shape[2], Pt.shape[3])
    assert float(unt[:, :, :N, N:].float().abs().max() if Pt.shape[-1] > N else 0.0) == 0.0, "tiled padding columns must be zero"
    rs = Pd.craft_rowsum
    assert Pd.dtype == torch.float16 and rs.shape == (B, M, N)
    assert float(Pd.float().max()) <= 1.0 and float(Pd[..., :N].float().amax(-1).min()) == 1.0      # exact maxima: the largest entry is 2^0
    assert float(Pd[..., N:].float().abs().max() if Pd.shape[-1] > N else 0.0) == 0.0, "padding columns must be zero"
    got = Pd[..., :N].float() / rs[..., None]
    assert (got.sum(-1) - 1).abs().max() < 2e-3
    close(got, ref, 0.0, 2e-3 if score == "f16x3" else 0.02, "fused deferred probabilities vs oracle")
    # the two-launch path on the same inputs
    monkeypatch.setenv("CRAFT_NO_FUSED_PROBS", "1")
    Po = ops.attn_probs(q, k, H8, W8, M, scale, tab.to(device), 1.0, mask_radius, mx, prec, defer=True)
    assert calls[-1] == "craft_attn_probs"
    close(Pd.float(), Po.float(), 2e-3, 1e-6, "fused vs two-launch P'")
    close(rs, Po.craft_rowsum, 1e-4, 1e-6, "fused vs two-launch row sums")


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("C", [128, 256])
def test_expanded_feat_trans(device, prec, C):
    """craft_linear_t + craft_attn_apply + craft_mode_pool_ln vs ExpandedFeatTrans."""
    if prec == PREC_F16X3:
        pytest.skip("P is stored as fp32 / bf16 / fp16; the split mode applies to fp32 operands only")
    B, H8, W8, M = 2, 11, 17, 4
    N = H8 * W8
    x = gen(B, N, C, seed=60)
    Pf = torch.softmax(gen(B, M, N, N, seed=61) * 2.0, dim=-1)
    Wv = gen(M * C, C, seed=62) / math.sqrt(C)
    w_agg = gen(1, C, seed=63) * 0.3
    skip = torch.tensor([0.7])
    ldp = ops.round_up(N, 32)
    P = torch.zeros(B, M, N, ldp, dtype=hip.PROB_DTYPE[prec])
    P[..., :N] = Pf.to(P.dtype)
    ref = O.expanded_feat_trans(x, P[..., :N].float(), Wv, w_agg, skip)
    xd = x.to(device)
    vT = ops.linear_t(xd, Wv.to(device), ldp, prec, Dv=C)
    Od = ops.attn_apply(P.to(device), vT, C, prec)
    # the same product from un-normalised rows + row sums (deferred softmax normalisation)
    scl = torch.rand(B, M, N) * 3 + 0.5
    Pun = (P.float() * scl[..., None]).to(P.dtype).to(device)
    Pun.craft_rowsum = scl.to(device)
    Od2 = ops.attn_apply(Pun, vT, C, prec)
    y = ops.mode_pool_ln(Od, xd, w_agg.to(device), skip.to(device))
    rt, at = TOL[prec]
    Oref = torch.matmul(P[..., :N].float(), F.linear(x, Wv).reshape(B, N, M, C).permute(0, 2, 1, 3))
    close(Od, Oref, rt, at, "P.V")
    close(Od2, Oref, max(rt, 1e-2 if prec != PREC_F32 else rt), max(at, 1e-2 if prec != PREC_F32 else at), "P.V with deferred row sums")
    close(y, ref, rt * 5, at * 5, "ExpandedFeatTrans")


@pytest.mark.parametrize("score", [PREC_F16X3, PREC_F16])
@pytest.mark.parametrize("H8,W8,mask_radius,gain", [(13, 19, -1, 2.5), (16, 32, 5, 2.5), (9, 40, -1, 80.0), (5, 6, -1, 2.5)])
def test_flash_attention(device, score, H8, W8, mask_radius, gain):
    """craft_flash_attention (scores -> online softmax -> P.V in one pass) vs the oracle's probabilities times V, and vs
    the two-kernel path (craft_attn_probs + craft_attn_apply) it replaces.  Ragged N, the positional window, the
    Chebyshev mask, and (gain 80) the score clamp are all exercised."""
    B, C, M, Dv = 2, 256, 4, 256
    N = H8 * W8
    x, _, Wq, Wk, _ = _qk(B, H8, W8, C, seed=150, gain=gain)
    tab = gen(15, 15, seed=151) * 0.5
    Wv = gen(M * Dv, C, seed=152) / math.sqrt(C)
    Pref = O.self_attn_probs(x, Wq, Wk, tab, 1.0, M, H8, W8, mask_radius)                       # [B, M, N, N]
    V = F.linear(x, Wv).reshape(B, N, M, Dv).permute(0, 2, 1, 3)
    Oref = torch.matmul(Pref, V)
    prec = hip.Precision(proj=PREC_F32, score=score, pv=PREC_F16)
    xd = x.to(device)
    q = ops.linear(xd, Wq.to(device), None, PREC_F32)
    k = ops.linear(xd, Wk.to(device), None, PREC_F32)
    scale = 1.0 / math.sqrt(C // M)
    mx = ops.score_max(q, k, H8, W8, M, scale, prec)
    assert ops.flash_supported(N, W8, C // M, Dv, prec)
    ldt = ops.round_up(N, 32)
    vT = ops.linear_t(xd, Wv.to(device), ldt, prec, Dv=Dv, acc_order=True)
    Of = ops.flash_attention(q, k, vT, H8, W8, M, Dv, scale, tab.to(device), 1.0, mask_radius, mx, prec)
    assert Of.shape == (B, M, N, Dv)
    # two-kernel path with the same roles
    P = ops.attn_probs(q, k, H8, W8, M, scale, tab.to(device), 1.0, mask_radius, mx, prec, defer=True)
    O2 = ops.attn_apply(P, ops.linear_t(xd, Wv.to(device), ldt, prec, Dv=Dv), Dv, prec)
    # fp16 P and V: 2^-11 relative per operand of a convex combination of O(1) values
    tol = 2e-3 if score == PREC_F16X3 else 2e-2
    close(Of, Oref, tol, tol, "flash attention vs oracle")
    close(Of, O2, tol, tol, "flash attention vs attn_probs + attn_apply")


def _conv_sd(seed=70):
    from craft_amd import CRAFT, default_args
    from craft_amd.synth import synth_state_dict
    m = CRAFT(default_args())
    sd = synth_state_dict(m.state_dict(), seed=seed)
    m.load_state_dict(sd)
    return m, sd


@pytest.mark.parametrize("prec", PRECS)
def test_update_block_pieces(device, prec):
    """craft_motion_encoder / craft_sepconv_gru / craft_flow_head / craft_mask_head vs update.py."""
    m, sd = _conv_sd()
    m = m.to(device).eval()
    ub = m.update_block
    B, H8, W8 = 2, 10, 14
    N = H8 * W8
    corr = gen(B, 324, H8, W8, seed=71)
    flow = gen(B, 2, H8, W8, seed=72) * 3
    net = torch.tanh(gen(B, 128, H8, W8, seed=73))
    inp = torch.relu(gen(B, 128, H8, W8, seed=74))
    mfg = gen(B, 128, H8, W8, seed=75)
    rt, at = TOL[prec]
    ws = ub.workspace(B, N, device)

    mf_ref = O.motion_encoder(flow, corr, sd)
    hx = torch.zeros(B, N, 512, device=device)
    corr_t = ops.tokens_from_nchw_wide(corr.to(device))
    flow_t = ops.tokens_from_nchw(flow.to(device))
    ub.encoder.forward_tokens(flow_t, corr_t, (H8, W8), hx[..., 256:384], ws, prec)
    close(ops.tokens_to_nchw(hx[..., 256:384], H8, W8), mf_ref, rt * 3, at * 10, f"motion encoder prec={prec}")
    assert float(hx[..., :256].abs().max()) == 0 and float(hx[..., 384:].abs().max()) == 0, "encoder wrote outside its columns"

    x_ref = torch.cat([inp, mf_ref, mfg], dim=1)
    h_ref = O.sepconv_gru(net, x_ref, sd)
    ops.tokens_from_nchw(net.to(device), out=hx[..., 0:128])
    ops.tokens_from_nchw(inp.to(device), out=hx[..., 128:256])
    ops.tokens_from_nchw(mf_ref.to(device), out=hx[..., 256:384])
    ops.tokens_from_nchw(mfg.to(device), out=hx[..., 384:512])
    ub.gru.forward_tokens(hx, (H8, W8), ws, prec)
    close(ops.tokens_to_nchw(hx[..., 0:128], H8, W8), h_ref, rt * 3, at * 10, f"SepConvGRU prec={prec}")
    close(ops.tokens_to_nchw(hx[..., 128:256], H8, W8), inp, 0, 0, "GRU must not touch x")
    # hoisted-context form: the inp channels enter through precomputed per-pixel bias fields
    ops.tokens_from_nchw(net.to(device), out=hx[..., 0:128])
    fields = ub.gru.context_tokens(hx[..., 128:256], (H8, W8), prec)
    ub.gru.step_tokens(hx, (H8, W8), ws, prec, fields, 128, 256)
    close(ops.tokens_to_nchw(hx[..., 0:128], H8, W8), h_ref, rt * 3, at * 10, f"SepConvGRU (hoisted context) prec={prec}")

    ops.tokens_from_nchw(h_ref.to(device), out=hx[..., 0:128])
    c0, c1, fl = ops.coords_init(flow.to(device), B, H8, W8, device)
    delta = torch.empty(B, N, 2, device=device)
    ub.flow_head_tokens(hx, (H8, W8), c1, c0, fl, delta, ws, prec)
    d_ref = O.flow_head(h_ref, sd)
    close(ops.tokens_to_nchw(delta, H8, W8), d_ref, rt * 3, at * 10, f"flow head prec={prec}")
    close(ops.tokens_to_nchw(fl, H8, W8), flow + d_ref, rt * 3, at * 10, "flow = coords1 - coords0")
    close(ops.tokens_to_nchw(c1, H8, W8), O.coords_grid(B, H8, W8) + flow + d_ref, rt * 3, at * 10, "coords1 += delta")

    mask = ub.mask_tokens(hx, (H8, W8), ws, prec)
    mk_ref = O.mask_head(h_ref, sd)
    close(ops.tokens_to_nchw(mask, H8, W8), mk_ref, rt * 3, at * 10, f"mask head prec={prec}")

    up = ops.convex_upsample(ops.tokens_from_nchw_wide(mk_ref.to(device)).contiguous(), ops.tokens_from_nchw((flow + d_ref).to(device)), H8, W8)
    close(up, O.convex_upsample(flow + d_ref, mk_ref), 1e-4, 1e-4, "convex upsample")


def test_module_level_api_matches_reference_shapes(device):
    """The NCHW module interfaces of the reference (SelfAttVisPosTrans / TransCorrBlock / GMAUpdateBlock)."""
    m, sd = _conv_sd(seed=80)
    m = m.to(device).eval()
    cfg = O.OracleConfig()
    B, H8, W8 = 1, 16, 24
    N = H8 * W8
    fmap1, fmap2 = gen(B, 256, H8, W8, seed=81), gen(B, 256, H8, W8, seed=82)
    inp = torch.relu(gen(B, 128, H8, W8, seed=83))
    net = torch.tanh(gen(B, 128, H8, W8, seed=84))
    f2 = m.f2_trans(fmap2.to(device))
    f2_ref = O.f2_transform(fmap2, sd, cfg)
    close(f2, f2_ref, 2e-4, 5e-5, "f2_trans(fmap2)")
    att = m.att(inp.to(device))
    att_ref = O.intra_attention(inp, sd, cfg)
    close(att, att_ref, 2e-4, 1e-6, "att(inp)")
    m.corr_fn.update(fmap1.to(device), f2, None, None, None)
    c = O.inter_corr_raw(fmap1, f2_ref, sd, cfg)
    mu, rstd = O.global_stats(c)
    pyr = O.build_pyramid(c, H8, W8)
    coords = O.coords_grid(B, H8, W8) + gen(B, 2, H8, W8, seed=85) * 1.5
    look = m.corr_fn(coords.to(device))
    look_ref = O.corr_lookup(pyr, coords, 4, mu, rstd)
    close(look, look_ref, 5e-4, 5e-4, "corr_fn(coords)")
    flow = coords - O.coords_grid(B, H8, W8)
    n2, mk, df = m.update_block(net.to(device), inp.to(device), look, flow.to(device), att)
    n2r, mkr, dfr = O.update_block(net, inp, look_ref, flow, att_ref, sd, cfg)
    close(n2, n2r, 1e-3, 1e-3, "update_block net")
    close(df, dfr, 1e-3, 1e-3, "update_block delta_flow")
    close(mk, mkr, 1e-3, 1e-3, "update_block mask")


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("KH,KW,cin,cout,relu", [(3, 3, 64, 192, True), (1, 5, 96, 256, False), (5, 1, 128, 126, True), (1, 1, 64, 64, True),
                                                 (1, 3, 64, 128, False), (3, 1, 96, 64, True)])
def test_conv2d_tokens(device, prec, KH, KW, cin, cout, relu):
    """craft_conv2d_nhwc (halo-tile kernel for KxK, generic implicit GEMM for 1x1) vs F.conv2d on an image whose
    size is not a multiple of the 8x16 patch (ragged patches, zero padding at every border).  3x3 / 1x5 / 5x1 take the
    static-tap variants of the packed-weight kernel, 1x3 / 3x1 its run-time tap loop."""
    B, H8, W8 = 2, 11, 21
    x = gen(B, cin, H8, W8, seed=90)
    w = gen(cout, cin, KH, KW, seed=91) / math.sqrt(cin * KH * KW)
    b = gen(cout, seed=92)
    ref = F.conv2d(x, w, b, padding=(KH // 2, KW // 2))
    if relu:
        ref = torch.relu(ref)
    xt = ops.tokens_from_nchw(x.to(device))
    rt, at = TOL[prec]
    for packed in ([False, True] if KH * KW > 1 else [False]):
        wp = ops.pack_conv_prec(w.to(device), prec) if packed else ops.pack_conv(w.to(device))
        y = ops.conv2d_tokens(xt, (H8, W8), wp, b.to(device), cout, KH, KW, ACT_RELU if relu else ACT_NONE, prec, packed=packed)
        close(ops.tokens_to_nchw(y, H8, W8), ref, rt * 2, at * 4, f"conv {KH}x{KW} prec={prec} packed={packed}")


@pytest.mark.parametrize("KH,KW,cin,cout", [(3, 3, 64, 192), (1, 5, 96, 256), (5, 1, 128, 128), (3, 3, 64, 64)])      # (the last: k_conv3x3_c64)
def test_conv2d_w16_is_the_full_product_on_fp16_weights(device, KH, KW, cin, cout):
    """CRAFT_CONV_W16 (the input-gradient convolutions of the "mixed" training policy): f16x3 with only the hi plane of the packed
    weights -- two MFMAs per product.  On weights that ARE fp16 numbers the dropped term is exactly zero, so the result must equal the
    three-term kernel bit for bit; on arbitrary weights it is the convolution with the weights rounded to fp16 (~2e-4 relative)."""
    from craft_amd.hip import call, CONV_W16, W_PACKED
    B, H8, W8 = 2, 11, 21
    x = gen(B, cin, H8, W8, seed=93)
    w = gen(cout, cin, KH, KW, seed=94) / math.sqrt(cin * KH * KW)
    b = gen(cout, seed=95)
    xt = ops.tokens_from_nchw(x.to(device))

    def run(wt, flag):
        wp = ops.pack_conv_prec(wt.to(device), PREC_F16X3)
        y = torch.empty(B, H8 * W8, cout, device=device)
        call("craft_conv2d_nhwc", xt, xt.stride(-2), cin, wp, b.to(device), cout, KH, KW, ACT_NONE, y, cout, B, H8, W8, PREC_F16X3 | W_PACKED | flag)
        return y
    w16 = w.half().float()
    assert torch.equal(run(w16, CONV_W16), run(w16, 0))
    got = ops.tokens_to_nchw(run(w, CONV_W16), H8, W8).cpu()
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=(KH // 2, KW // 2)).float()
    ref16 = F.conv2d(x.double(), w16.double(), b.double(), padding=(KH // 2, KW // 2)).float()
    assert (got - ref16).abs().max().item() < 2e-5 * ref.abs().max().item()
    rel = ((got - ref).norm() / ref.norm()).item()
    assert 1e-5 < rel < 6e-4, rel


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("cin,cout,shape", [(64, 64, (2, 24, 40)), (96, 96, (2, 11, 21)), (128, 128, (1, 9, 17))])
def test_conv2d_residual_epilogue(device, prec, cin, cout, shape):
    """craft_conv2d_nhwc_res: relu(res + relu(conv3x3(x) + bias)) -- ResidualBlock's tail (extractor.py:56-63) in the epilogue of its second
    convolution -- against F.conv2d; 64 -> 64 runs the persistent layer-1 kernel, the others the halo kernel (ragged patches)."""
    from craft_amd.hip import call, W_PACKED
    B, H8, W8 = shape
    x = gen(B, cin, H8, W8, seed=110)
    r = gen(B, cout, H8, W8, seed=111)
    w = gen(cout, cin, 3, 3, seed=112) / math.sqrt(cin * 9)
    b = gen(cout, seed=113)
    ref = torch.relu(r + torch.relu(F.conv2d(x, w, b, padding=1)))
    xt, rtok = ops.tokens_from_nchw(x.to(device)), ops.tokens_from_nchw(r.to(device))
    packed = prec != PREC_F32
    wp = ops.pack_conv_prec(w.to(device), prec) if packed else ops.pack_conv(w.to(device))
    y = torch.empty(B, H8 * W8, cout, device=device)
    call("craft_conv2d_nhwc_res", xt, xt.stride(1), cin, wp, b.to(device), cout, 3, 3, ACT_RELU, rtok, rtok.stride(1), y, cout, B, H8, W8,
         prec | (W_PACKED if packed else 0))
    rt, at = TOL[prec]
    close(ops.tokens_to_nchw(y, H8, W8), ref, rt * 2, at * 4, f"conv + residual prec={prec}")


def test_conv2d_relu_mask_epilogue_and_multi_copy(device):
    """craft_conv2d_nhwc2_mask: the ReLU backward of the layer below an input-gradient convolution in its epilogue -- bit-equal to
    craft_conv2d_nhwc2 followed by craft_act_bwd (with and without a per-pixel bias field); craft_multi_copy: ragged tensors into a flat
    buffer in one launch."""
    from craft_amd.hip import call, carray, W_PACKED
    import ctypes
    B, H8, W8, cin, cout = 2, 11, 21, 64, 96
    x = gen(B, H8 * W8, cin, seed=120).to(device)
    ysave = gen(B, H8 * W8, cout, seed=121).to(device)
    field = gen(B, H8 * W8, cout, seed=122).to(device)
    w = gen(cout, cin, 3, 3, seed=123) / math.sqrt(cin * 9)
    wp = ops.pack_conv_prec(w.to(device), PREC_F16X3)
    zb = torch.zeros(cout, device=device)
    for f in (None, field):
        a = torch.empty(B, H8 * W8, cout, device=device)
        b = torch.empty_like(a)
        call("craft_conv2d_nhwc2", x, cin, cin, None, 0, 0, wp, zb if f is None else None, f, cout if f is not None else 0, cout, 3, 3, ACT_NONE, a, cout,
             B, H8, W8, PREC_F16X3 | W_PACKED)
        call("craft_act_bwd", a, cout, ysave, cout, a, cout, B * H8 * W8, cout, ACT_RELU, 1.0)
        call("craft_conv2d_nhwc2_mask", x, cin, cin, None, 0, 0, wp, zb if f is None else None, f, cout if f is not None else 0, cout, 3, 3, ysave, cout,
             b, cout, B, H8, W8, PREC_F16X3 | W_PACKED)
        assert torch.equal(a, b)
    sizes = [1, 7, 2048, 2049, 5000, 0, 33, 12345]
    srcs = [gen(max(n, 1), seed=130 + i)[:n].to(device) for i, n in enumerate(sizes)]
    offs, off = [], 3
    for n in sizes:
        offs.append(off)
        off += n + 5
    flat = torch.full((off,), -7.0, device=device)
    call("craft_multi_copy", carray(ctypes.c_void_p, [t.data_ptr() if t.numel() else 0 for t in srcs]), carray(ctypes.c_long, sizes),
         carray(ctypes.c_long, offs), None, len(sizes), flat)
    ref = torch.full((off,), -7.0)
    for t, o, n in zip(srcs, offs, sizes):
        ref[o:o + n] = t.cpu()
    assert torch.equal(flat.cpu(), ref)
    # conv weight gradients in the weight-gradient kernels' [cout][KH][KW][cin] layout land as [cout][cin][KH][KW] (chlast)
    shapes = [(96, 3, 3, 64), (7, 1, 5, 36), (128, 7, 7, 2), (5, 1, 1, 324)]
    cls = [gen(*sh, seed=140 + i).to(device) for i, sh in enumerate(shapes)]
    plain = gen(77, seed=150).to(device)
    srcs = [cls[0], plain] + cls[1:]
    sizes = [t.numel() for t in srcs]
    chl = [shapes[0][3] * 1024 + 9, 0, 36 * 1024 + 5, 2 * 1024 + 49, 324 * 1024 + 1]
    offs, off = [], 1
    for n in sizes:
        offs.append(off)
        off += n + 3
    flat = torch.full((off,), -7.0, device=device)
    call("craft_multi_copy", carray(ctypes.c_void_p, [t.data_ptr() for t in srcs]), carray(ctypes.c_long, sizes), carray(ctypes.c_long, offs),
         carray(ctypes.c_long, chl), len(srcs), flat)
    ref = torch.full((off,), -7.0)
    for t, o, n, c in zip(srcs, offs, sizes, chl):
        ref[o:o + n] = (t.permute(0, 3, 1, 2) if c else t).cpu().reshape(-1)
    assert torch.equal(flat.cpu(), ref)
    with pytest.raises(hip.CraftHipError):                 # a size that is not a whole number of [taps][cin] rows
        call("craft_multi_copy", carray(ctypes.c_void_p, [plain.data_ptr()]), carray(ctypes.c_long, [77]), carray(ctypes.c_long, [0]),
             carray(ctypes.c_long, [4 * 1024 + 9]), 1, flat)


def test_act_bwd2_and_field_column_start(device):
    """craft_act_bwd2 = craft_act_bwd on dy + dy2 with the last channels zeroed; craft_conv2d_nhwc2 with CRAFT_CONV_FIELD_COL0: the bias
    field reaches columns >= c only (bit-equal to the convolution with a field whose first c columns are zero)."""
    from craft_amd.hip import call, W_PACKED
    B, N, C = 2, 301, 128
    dy, dy2buf, y = gen(B, N, C, seed=160).to(device), gen(B, N, 384, seed=161).to(device), gen(B, N, 200, seed=162).to(device)
    dy2 = dy2buf[..., 128:256]
    ref = torch.empty(B, N, C, device=device)
    s_ = dy + dy2
    call("craft_act_bwd", s_, C, y, y.stride(-2), ref, C, B * N, C, ACT_RELU, 1.0)
    ref[..., 126:] = 0.0
    out = dy.clone()
    call("craft_act_bwd2", out, C, dy2, dy2.stride(-2), y, y.stride(-2), out, C, B * N, C, ACT_RELU, 1.0, 2)      # in place over dy
    assert torch.equal(out, ref)
    out = torch.empty_like(dy)
    call("craft_act_bwd2", dy, C, None, 0, y, y.stride(-2), out, C, B * N, C, ACT_RELU, 0.5, 0)
    ref2 = torch.empty_like(dy)
    call("craft_act_bwd", dy, C, y, y.stride(-2), ref2, C, B * N, C, ACT_RELU, 0.5)
    assert torch.equal(out, ref2)
    H8, W8, cin, cout = 9, 14, 128, 384
    x = gen(B, H8 * W8, cin, seed=163).to(device)
    w = (gen(cout, cin, 1, 5, seed=164) / math.sqrt(5 * cin)).to(device)
    wp = ops.pack_conv_weights(w, PREC_F16X3)
    field = gen(B, H8 * W8, cout, seed=165).to(device)
    fz = field.clone()
    fz[..., :128] = 0.0
    a, b = torch.empty(B, H8 * W8, cout, device=device), torch.empty(B, H8 * W8, cout, device=device)
    call("craft_conv2d_nhwc2", x, cin, cin, None, 0, 0, wp, None, field, cout, cout, 1, 5, ACT_NONE, a, cout, B, H8, W8, PREC_F16X3 | W_PACKED | ((128 // 32) << 16))
    call("craft_conv2d_nhwc2", x, cin, cin, None, 0, 0, wp, None, fz, cout, cout, 1, 5, ACT_NONE, b, cout, B, H8, W8, PREC_F16X3 | W_PACKED)
    assert torch.equal(a, b)


def test_forward_interpolate(device):
    """craft_forward_interpolate vs the reference's outputs (tests/golden/forward_interpolate.npz, generated by running
    utils.py:34-62) and vs the oracle on a batch of fresh flows, including one with no valid source."""
    import os
    from craft_amd.utils import forward_interpolate
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "forward_interpolate.npz"))
    for n in sorted(k[:-3] for k in z.files if k.endswith(".in")):
        got = forward_interpolate(torch.from_numpy(z[n + ".in"]).to(device)).cpu().numpy()
        assert np.array_equal(got, z[n + ".out"]), f"{n}: {int((got != z[n + '.out']).any(0).sum())} pixels differ"
    f = gen(3, 2, 21, 37, seed=300) * 4.0
    f[2] = 1000.0                                       # every source leaves the frame: zeros
    got = forward_interpolate(f.to(device)).cpu()
    for b in range(3):
        assert torch.equal(got[b], O.forward_interpolate(f[b])), f"sample {b}"
    assert float(got[2].abs().max()) == 0.0
