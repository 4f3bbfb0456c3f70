"""Tensor-level wrappers over the C ABI: allocate outputs with torch, enqueue the HIP kernels.

Every function maps 1:1 to an entry point of ``include/craft_hip.h``; "tokens" = channels-last fp32
``[B, N, C]`` (possibly a column slice of a wider buffer: pass ``t[..., a:b]`` views, their row stride
is forwarded as ``ld``).  No compute happens in Python.
"""
from __future__ import annotations

import math
import os
import threading
from typing import Optional, Sequence

import torch

from . import hip
from .hip import ACT_NONE, ACT_RELU, ACT_TANH, PREC_F32, PROB_DTYPE, call, pick, round_up


def _ld(t: torch.Tensor) -> int:
    """Row stride (in elements) of a tokens view [..., n, c] whose channel stride is 1."""
    if t.stride(-1) != 1:
        raise hip.CraftHipError("tokens view must have unit channel stride")
    return t.stride(-2)


def _check_rows(t: torch.Tensor):
    # rows of a [B, N, C] view must be uniformly strided: stride(0) == N * stride(1)
    if t.dim() == 3 and t.shape[0] > 1 and t.stride(0) != t.shape[1] * t.stride(1):
        raise hip.CraftHipError("tokens view must be uniformly row-strided")


def tokens_from_nchw(x: torch.Tensor, c_off: int = 0, C: Optional[int] = None, act: int = ACT_NONE, ln: bool = False,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """NCHW [B, Ctot, H, W] channels [c_off, c_off+C) -> tokens [B, H*W, C] (+activation, +LayerNorm)."""
    B, Ctot, H, W = x.shape
    C = Ctot - c_off if C is None else C
    x = x.contiguous()
    if out is None:
        out = torch.empty(B, H * W, C, device=x.device, dtype=torch.float32)
    _check_rows(out)
    call("craft_tokens", x, 1, B, Ctot, c_off, C, H * W, 0, act, int(ln), out, _ld(out))
    return out


def tokens_from_nchw_wide(x: torch.Tensor) -> torch.Tensor:
    """NCHW -> tokens for any channel count (chunks of <= 256 channels written into column slices)."""
    B, C, H, W = x.shape
    ld = round_up(C, 4)
    out = torch.empty(B, H * W, ld, device=x.device, dtype=torch.float32)
    if ld != C:
        out[..., C:].zero_()
    for c0 in range(0, C, 256):
        c1 = min(C, c0 + 256)
        tokens_from_nchw(x, c_off=c0, C=c1 - c0, out=out[..., c0:c1])
    return out[..., :C]


def tokens_norm(t: torch.Tensor, act: int = ACT_NONE, ln: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """tokens -> (activation) -> LayerNorm over C -> tokens."""
    B, N, C = t.shape
    _check_rows(t)
    if out is None:
        out = torch.empty(B, N, C, device=t.device, dtype=torch.float32)
    call("craft_tokens", t, 0, B, C, 0, C, N, _ld(t), act, int(ln), out, _ld(out))
    return out


def tokens_slice(t: torch.Tensor, c_off: int, C: int, act: int = ACT_NONE, ln: bool = False,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """tokens columns [c_off, c_off+C) -> (activation) -> (LayerNorm) -> tokens [B, N, C]."""
    B, N, Ctot = t.shape
    _check_rows(t)
    if out is None:
        out = torch.empty(B, N, C, device=t.device, dtype=torch.float32)
    _check_rows(out)
    call("craft_tokens", t, 0, B, Ctot, c_off, C, N, _ld(t), act, int(ln), out, _ld(out))
    return out


def tokens_to_nchw(t: torch.Tensor, H: int, W: int) -> torch.Tensor:
    B, N, C = t.shape
    _check_rows(t)
    out = torch.empty(B, C, H, W, device=t.device, dtype=torch.float32)
    call("craft_tokens_to_nchw", t, _ld(t), B, C, N, out)
    return out


def pack_linear_weight(w: torch.Tensor, prec: int) -> torch.Tensor:
    """[Cout, Cin] fp32 -> craft_pack_weights' MFMA fragment order with Cin padded to a multiple of 32 by zero columns: the weight
    operand craft_linear / craft_linear_t / craft_motion_encoder's convc1 take with CRAFT_W_PACKED / CRAFT_W1X1_PACKED (bf16 / fp16 /
    f16x3)."""
    cout, cin = w.shape
    Kp = round_up(cin, 32)
    m = w.detach().float()
    if Kp != cin:
        m = torch.cat([m, torch.zeros(cout, Kp - cin, device=w.device)], dim=1)
    m = m.contiguous()
    planes = 2 if prec == hip.PREC_F16X3 else 1
    out = torch.empty(planes * round_up(cout, 32) * Kp, device=w.device, dtype=torch.bfloat16 if prec == hip.PREC_BF16 else torch.float16)
    call("craft_pack_weights", m, cout, Kp, prec, out)
    return out


def linear_pack(owner, name: str, w: torch.Tensor, prec: int) -> Optional[torch.Tensor]:
    """The packed copy of the nn.Linear / 1x1-conv weight ``w`` (2-D view) for the ``proj`` role of ``prec``, cached on the module
    ``owner`` under ``name`` and rebuilt when the parameter is updated in place or replaced; None when the product runs in fp32, under
    autograd (a training step re-packs through its own registry) or with CRAFT_NO_LINEAR_PACK (A/B switch)."""
    pp = pick(prec, "proj")
    if pp == PREC_F32 or not w.is_cuda or torch.is_grad_enabled() and w.requires_grad or os.environ.get("CRAFT_NO_LINEAR_PACK"):
        return None
    cache = owner.__dict__.setdefault("_linear_packs", {})
    key = (pp, hip.weights_epoch(), w.data_ptr(), w._version, w.device, tuple(w.shape))
    e = cache.get(name)
    if e is None or e[0] != key:
        with torch.no_grad():
            e = cache[name] = (key, pack_linear_weight(w, pp))
        torch.cuda.current_stream().synchronize()       # (once per weight update; the copy may be consumed on other HIP streams)
    return e[1]


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], prec: int,
           out: Optional[torch.Tensor] = None, packed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Linear on tokens: [B, N, Cin] x [Cout, Cin]^T (+bias) -> [B, N, Cout].  ``packed``: ``linear_pack`` of the same weight (the
    product then reads the weights in MFMA fragment order, k_gemm_rows_wf)."""
    B, N, Cin = x.shape
    _check_rows(x)
    Cout = w.shape[0]
    if out is None:
        out = torch.empty(B, N, Cout, device=x.device, dtype=torch.float32)
    _check_rows(out)
    if packed is not None:
        call("craft_linear", x, _ld(x), packed, bias, out, _ld(out), B * N, Cin, Cout, pick(prec, "proj") | hip.W_PACKED)
    else:
        call("craft_linear", x, _ld(x), w.contiguous(), bias, out, _ld(out), B * N, Cin, Cout, pick(prec, "proj"))
    return out


def linear_t(x: torch.Tensor, w: torch.Tensor, ldt: int, prec: int, out: Optional[torch.Tensor] = None,
             Dv: int = 0, acc_order: bool = False, packed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Per-sample transposed projection: yT[b][o][n] (row stride ldt >= N, tail zero), stored in the
    element type of the pv role (the type attn_apply consumes).  With ``Dv`` (the per-mode value width) and a 16-bit
    pv type the result is in the MFMA fragment order craft_attn_apply requires (same shape / footprint);
    ``acc_order``: the key order ``flash_attention`` requires instead."""
    B, N, Cin = x.shape
    _check_rows(x)
    Cout = w.shape[0]
    pv = pick(prec, "pv")
    if out is None:
        # the tail columns N .. ldt-1 must read as zero; without a tail (N a multiple of 32: every BASELINE size) the kernel writes
        # every element and the 29 MB zero fill per aggregator call (12 per forward at 448x1024) is skipped
        alloc = torch.empty if ldt == N else torch.zeros
        out = alloc(B, Cout, ldt, device=x.device, dtype=PROB_DTYPE[pv])
    frag = Dv if (Dv and pv != PREC_F32) else 0
    if acc_order:
        if not frag:
            raise hip.CraftHipError("acc_order needs a 16-bit pv type and Dv")
        frag |= hip.FRAG_ACC_ORDER
    if packed is not None and frag:                     # (``packed``: linear_pack of the same weight; fragment-order results only)
        call("craft_linear_t", x, _ld(x), packed, out, ldt, B, N, Cin, Cout, pv, frag, pick(prec, "proj") | hip.W_PACKED)
    else:
        call("craft_linear_t", x, _ld(x), w.contiguous(), out, ldt, B, N, Cin, Cout, pv, frag, pick(prec, "proj"))
    return out


def score_max(q: torch.Tensor, k: torch.Tensor, H8: int, W8: int, M: int, scale: float, prec: int,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Device-side global max of the raw scaled scores as an ordered uint32 (setrans.py:520-529)."""
    B, N, C = q.shape
    if out is None:
        out = torch.zeros(32, device=q.device, dtype=torch.int32)
    call("craft_score_max", q, _ld(q), k, _ld(k), B, H8, W8, M, C // M, scale, out, pick(prec, "score"))
    return out


def decode_ord(o: torch.Tensor) -> float:
    """Host decode of the ordered-uint max (diagnostics / tests only; forces a sync)."""
    u = int(o[0].item()) & 0xFFFFFFFF
    import struct
    if u == 0:
        return float("nan")
    bits = (u & 0x7FFFFFFF) if (u & 0x80000000) else (~u & 0xFFFFFFFF)
    return struct.unpack("<f", struct.pack("<I", bits))[0]


class CorrPyramid:
    """Un-normalised correlation pyramid + lazy global-LayerNorm statistics.  ``tiled``: levels 0 / 1 are stored per query as 8 x 16 /
    4 x 8 tiles (CRAFT_PYR_TILED: the patch one workgroup of the fused build kernel writes -- whole 128-byte lines); ``dense(l)`` gives
    the row-major image of any level."""

    def __init__(self, B: int, H8: int, W8: int, levels: int, device, tiled: bool = False):
        self.B, self.H8, self.W8, self.levels = B, H8, W8, levels
        self.tiled = bool(tiled) and levels == 4 and min(H8, W8) >= 8
        N = H8 * W8
        self.lv, self.dims = [], []
        h, w = H8, W8
        for l in range(levels):
            self.dims.append((h, w))
            if self.tiled and l < 2:
                th, tw = (8, 16) if l == 0 else (4, 8)
                self.lv.append(torch.empty(B * N, -(-h // th) * -(-w // tw) * th * tw, device=device, dtype=torch.float32))
            else:
                self.lv.append(torch.empty(B * N, h, w, device=device, dtype=torch.float32))
            h, w = h // 2, w // 2
        self.sums = torch.zeros(B, 2, device=device, dtype=torch.float64)
        self.mu_rstd = torch.empty(B, 2, device=device, dtype=torch.float32)

    def nbytes(self) -> int:
        """Algorithmic bytes of the pyramid (the image sizes; a tiled level's padding is not counted)."""
        return sum(self.lv[l].shape[0] * h * w * 4 for l, (h, w) in enumerate(self.dims))

    def dense(self, l: int) -> torch.Tensor:
        """Level l as [B*N, h, w] row-major images (a copy for a tiled level: tests / debugging)."""
        h, w = self.dims[l]
        if not (self.tiled and l < 2):
            return self.lv[l]
        th, tw = (8, 16) if l == 0 else (4, 8)
        nty, ntx = -(-h // th), -(-w // tw)
        t = self.lv[l].view(-1, nty, ntx, th, tw).permute(0, 1, 3, 2, 4).reshape(-1, nty * th, ntx * tw)
        return t[:, :h, :w].contiguous()

    def batch_slice(self, b0: int, b1: int) -> "CorrPyramid":
        """View of samples [b0, b1) (no copy): the refinement loop of a batch slice can run on its own stream."""
        v = CorrPyramid.__new__(CorrPyramid)
        v.B, v.H8, v.W8, v.levels, v.tiled, v.dims = b1 - b0, self.H8, self.W8, self.levels, self.tiled, self.dims
        N = self.H8 * self.W8
        v.lv = [t[b0 * N:b1 * N] for t in self.lv]
        v.sums, v.mu_rstd = self.sums[b0:b1], self.mu_rstd[b0:b1]
        return v


def fused_pyramid(C: int, M: int, prec, levels: int, H8: int, W8: int) -> bool:
    """True when ``corr_build`` will run the fused build + pyramid kernel (craft_corr_build_pyramid) -- the caller may then allocate a
    tiled CorrPyramid (its stores fill whole 128-byte lines)."""
    return (pick(prec, "score") == hip.PREC_F16X3 and M == 4 and C == 256 and levels == 4 and min(H8, W8) >= 8
            and not os.environ.get("CRAFT_NO_FUSED_PYRAMID"))


def corr_build(q: torch.Tensor, k: torch.Tensor, H8: int, W8: int, M: int, scale: float, pos_tab: Optional[torch.Tensor],
               pos_w: float, w_aggr: float, clamp_ord: Optional[torch.Tensor], pyr: CorrPyramid, do_norm: bool,
               prec: int) -> CorrPyramid:
    B, N, C = q.shape
    R = 0 if pos_tab is None else (pos_tab.shape[0] - 1) // 2
    sp = pick(prec, "score")
    # f16x3, 4 modes of 64: scratch for the pre-split (hi / lo fp16) copies of Q and K
    ws = torch.empty(4 * B * N * C, device=q.device, dtype=torch.float16) if (sp == hip.PREC_F16X3 and M == 4 and C == 256) else None
    tab = None if pos_tab is None else pos_tab.contiguous()
    lv = pyr.lv + [None] * (4 - len(pyr.lv))
    if ws is not None and len(pyr.lv) == 4 and min(H8, W8) >= 8 and not os.environ.get("CRAFT_NO_FUSED_PYRAMID"):
        # pyramid written from the tile that produced level 0 (no re-read of the 604 MB level 0 at 768x1024)
        call("craft_corr_build_pyramid", q, _ld(q), k, _ld(k), B, H8, W8, M, C // M, scale, tab, R, pos_w, w_aggr, clamp_ord,
             lv[0], lv[1], lv[2], lv[3], pyr.sums, ws, sp | (hip.PYR_TILED if pyr.tiled else 0))
        call("craft_corr_finish", lv[0], None, None, None, pyr.sums, pyr.mu_rstd, B, H8, W8, int(do_norm))
        return pyr
    if pyr.tiled:
        raise hip.CraftHipError("a tiled CorrPyramid can only be filled by the fused build (f16x3 scores, 4 modes of 64, 4 levels, H8, W8 >= 8)")
    call("craft_corr_build", q, _ld(q), k, _ld(k), B, H8, W8, M, C // M, scale, tab, R, pos_w, w_aggr, clamp_ord, pyr.lv[0], pyr.sums, ws, sp)
    call("craft_corr_finish", lv[0], lv[1], lv[2], lv[3], pyr.sums, pyr.mu_rstd, B, H8, W8, int(do_norm))
    return pyr


def corr_lookup(pyr, coords: torch.Tensor, radius: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """coords tokens [B, N, 2] -> tokens [B, N, V*levels*(2r+1)^2].  ``pyr`` is one CorrPyramid or, for the two-way
    correlation of ``--f1`` (corr.py:164-171), a list of V of them: level l then holds [volume 0 | volume 1 | ...]."""
    pyrs = list(pyr) if isinstance(pyr, (list, tuple)) else [pyr]
    B, N, _ = coords.shape
    win2 = (2 * radius + 1) ** 2
    V = len(pyrs)
    nch = pyrs[0].levels * win2 * V
    if out is None:
        out = torch.empty(B, N, nch, device=coords.device, dtype=torch.float32)
    co = coords.contiguous()
    for v, pv in enumerate(pyrs):
        lv = pv.lv + [None] * (4 - len(pv.lv))
        call("craft_corr_lookup", lv[0], lv[1], lv[2], lv[3], pv.levels | (hip.PYR_TILED if getattr(pv, "tiled", False) else 0), pv.mu_rstd, co,
             B, pv.H8, pv.W8, radius, out, _ld(out), win2 * V, win2 * v)
    return out


def attn_probs(q: torch.Tensor, k: torch.Tensor, H8: int, W8: int, M: int, scale: float, pos_tab: Optional[torch.Tensor],
               pos_w: float, mask_radius: int, clamp_ord: Optional[torch.Tensor], prec: int,
               out: Optional[torch.Tensor] = None, defer: bool = False, relpos=None, tiled: Optional[bool] = None) -> torch.Tensor:
    """P [B, M, N, ldp] (ldp = N rounded up to 32; the tail columns are zero).  ``defer=True``: P is left un-normalised
    (exp(logit - rowmax)) and carries its row sums as ``P.craft_rowsum`` [B, M, N]; ``attn_apply`` divides by them.
    Only for P that goes straight to ``attn_apply`` -- anything handed to a caller is normalised.
    ``tiled`` (deferred one-launch path only; default on, ``CRAFT_P_ROWMAJOR=1`` turns it off): P is returned in 32-query x 64-key
    tiles (``hip.P_TILED``; shape [B, M, N rounded up to 32, N rounded up to 64], ``P.craft_tiled`` / ``P.craft_n`` set) -- opaque to
    everything but ``attn_apply`` / ``probs_slice`` / ``probs_rowmajor``.
    ``relpos = (Hs [B, M, N, 2*H8-1], Ws [B, M, N, 2*W8-1], weight)``: per-query relative-position scores (gma.RelPosEmb)."""
    B, N, C = q.shape
    ldp = round_up(N, 32)
    if pick(prec, "pv") not in PROB_DTYPE:
        raise hip.CraftHipError("inference stores the attention probabilities in the pv type: fp32, bf16 or fp16 (f16x3 is a "
                                "training-only pv mode)")
    given_out = out
    if out is None:
        out = torch.empty(B, M, N, ldp, device=q.device, dtype=PROB_DTYPE[pick(prec, "pv")])
    R = 0 if pos_tab is None else (pos_tab.shape[0] - 1) // 2
    pvp, sp = pick(prec, "pv"), pick(prec, "score")
    if (defer and relpos is None and C // M == 32 and sp in (hip.PREC_F16, hip.PREC_F16X3) and pvp in (hip.PREC_F16, hip.PREC_BF16)
            and N < 65536 and not os.environ.get("CRAFT_NO_FUSED_PROBS")):
        # one launch of independent waves (craft_attn_probs_fused): maxima + P' + row sums, keys pre-split in fragment order
        rowsum = torch.empty(B, M, N, device=q.device, dtype=torch.float32)
        ws = torch.empty(B * M * ((N + 127) // 128) * 16384, device=q.device, dtype=torch.uint8)
        if tiled is None:
            tiled = given_out is None and not os.environ.get("CRAFT_P_ROWMAJOR")
        if tiled:
            ldp = round_up(N, 64)
            if given_out is None:
                out = torch.empty(B, M, round_up(N, 32), ldp, device=q.device, dtype=PROB_DTYPE[pvp])
            elif tuple(out.shape) != (B, M, round_up(N, 32), ldp):
                raise hip.CraftHipError("tiled probabilities need an out of shape [B, M, round_up(N, 32), round_up(N, 64)]")
        call("craft_attn_probs_fused", q, _ld(q), k, _ld(k), B, H8, W8, M, C // M, scale,
             None if pos_tab is None else pos_tab.contiguous(), R, pos_w, mask_radius, clamp_ord, out, ldp, rowsum, ws,
             pvp | (hip.P_TILED if tiled else 0), sp)
        out.craft_rowsum = rowsum
        if tiled:
            out.craft_tiled, out.craft_n = True, N
        return out
    if tiled:
        raise hip.CraftHipError("tiled probabilities exist only on the deferred one-launch path (craft_attn_probs_fused)")
    # row sums | scratch: row maxima, per-key-chunk partial sums (CRAFT_ATTN_CHUNK_KEYS = 1024)
    rowsum = torch.empty(2 + (N + 1023) // 1024, B, M, N, device=q.device, dtype=torch.float32) if defer else None
    rph, rpw, rpwt = (None, None, 0.0) if relpos is None else (relpos[0].contiguous(), relpos[1].contiguous(), float(relpos[2]))
    if relpos is not None and (tuple(rph.shape) != (B, M, N, 2 * H8 - 1) or tuple(rpw.shape) != (B, M, N, 2 * W8 - 1)):
        raise hip.CraftHipError("relpos tables must be [B, M, N, 2*H8-1] and [B, M, N, 2*W8-1]")
    call("craft_attn_probs", q, _ld(q), k, _ld(k), B, H8, W8, M, C // M, scale,
         None if pos_tab is None else pos_tab.contiguous(), R, pos_w, mask_radius, clamp_ord, rph, rpw, rpwt, out, ldp, rowsum,
         pick(prec, "pv"), pick(prec, "score"))
    if defer:
        out.craft_rowsum = rowsum[0]
    return out


def flash_supported(N: int, W8: int, d: int, Dv: int, prec: int) -> bool:
    """Shapes / precisions craft_flash_attention implements (anything else: attn_probs + attn_apply)."""
    return (d == 64 and Dv == 256 and pick(prec, "pv") == hip.PREC_F16 and pick(prec, "score") in (hip.PREC_F16, hip.PREC_F16X3)
            and N < 65536 and W8 >= 2)


def flash_attention(q: torch.Tensor, k: torch.Tensor, vT: torch.Tensor, H8: int, W8: int, M: int, Dv: int, scale: float,
                    pos_tab: Optional[torch.Tensor], pos_w: float, mask_radius: int, clamp_ord: Optional[torch.Tensor], prec: int,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """O [B, M, N, Dv] = softmax(scores) @ V in one pass (the probabilities never reach memory).  ``vT`` from
    ``linear_t(..., Dv=Dv, acc_order=True)`` with ldt = N rounded up to 32."""
    B, N, C = q.shape
    d = C // M
    ldt = vT.shape[-1]
    sp = pick(prec, "score")
    planes = 2 if sp == hip.PREC_F16X3 else 1
    ws = torch.empty(B * M * (8 * ((N + 255) // 256) + (N + 31) // 32) * (d // 16) * planes * 1024, device=q.device, dtype=torch.uint8)
    if out is None:
        out = torch.empty(B, M, N, Dv, device=q.device, dtype=torch.float32)
    R = 0 if pos_tab is None else (pos_tab.shape[0] - 1) // 2
    call("craft_flash_attention", q, _ld(q), k, _ld(k), vT, ldt, B, H8, W8, M, d, Dv, scale,
         None if pos_tab is None else pos_tab.contiguous(), R, pos_w, mask_radius, clamp_ord, out, ws, sp, pick(prec, "pv"))
    return out


def vt_stride(P: torch.Tensor) -> int:
    """Row stride (key extent) of the V^T operand that ``attn_apply`` expects beside ``P``: P's own row stride for a row-major P, N
    rounded up to 32 for a tiled one (whose key extent is rounded to 64: include/craft_hip.h, craft_attn_apply)."""
    return round_up(P.craft_n, 32) if getattr(P, "craft_tiled", False) else P.shape[-1]


def probs_slice(P: torch.Tensor, b0: int, b1: int) -> torch.Tensor:
    """Batch slice of an ``attn_probs`` result that keeps the deferred row sums attached."""
    v = P[b0:b1]
    rs = getattr(P, "craft_rowsum", None)
    if rs is not None:
        v.craft_rowsum = rs[b0:b1]
    if getattr(P, "craft_tiled", False):
        v.craft_tiled, v.craft_n = True, P.craft_n
    return v


def probs_rowmajor(P: torch.Tensor) -> torch.Tensor:
    """Row-major [B, M, N, N] copy of an ``attn_probs`` result in either layout (tests / debugging; still un-normalised when
    the row sums are deferred)."""
    if not getattr(P, "craft_tiled", False):
        return P[..., :P.shape[2]]
    B, M, Nr, ldp = P.shape
    N = P.craft_n
    return P.view(B, M, Nr // 32, ldp // 64, 32, 64).permute(0, 1, 2, 4, 3, 5).reshape(B, M, Nr, ldp)[:, :, :N, :N]


def probs_tiled(P: torch.Tensor, fill: float = 0.0) -> torch.Tensor:
    """Tiled (``hip.P_TILED``) copy of a row-major P [B, M, N, >= N] (torch ops, not a kernel: tests and callers that build P
    themselves); the padding rows are filled with ``fill`` (the kernels never write them and the consumer must not depend on them)."""
    B, M, N = P.shape[:3]
    Nr, ldp = round_up(N, 32), round_up(N, 64)
    full = torch.full((B, M, Nr, ldp), fill, device=P.device, dtype=P.dtype)
    full[:, :, :N, :N] = P[..., :N]
    full[:, :, :N, N:] = 0
    out = full.view(B, M, Nr // 32, 32, ldp // 64, 64).permute(0, 1, 2, 4, 3, 5).contiguous().view(B, M, Nr, ldp)
    rs = getattr(P, "craft_rowsum", None)
    if rs is not None:
        out.craft_rowsum = rs
    out.craft_tiled, out.craft_n = True, N
    return out


def attn_apply(P: torch.Tensor, vT: torch.Tensor, Dv: int, prec: int, out: Optional[torch.Tensor] = None,
               rows32: int = 0) -> torch.Tensor:
    """O[b][m] = P[b][m] @ V_m with vT [B, M*Dv, ldp] (16-bit: fragment order, ``linear_t(..., Dv=Dv)``) -> O [B, M, N, Dv].
    ``rows32`` (0 = chosen from the grid size; 4..7: 32-row groups per block of the 4-wave 16-bit kernel; 8 / 10 / 12 / 14: the 8-wave kernel
    with rows32 / 2 groups per row half -- CRAFT_PV_ROWS)."""
    B, M, N, ldp = P.shape
    tiled = getattr(P, "craft_tiled", False)
    if tiled:
        N = P.craft_n
        if not P.is_contiguous():
            raise hip.CraftHipError("tiled probabilities must be contiguous")
    if out is None:
        out = torch.empty(B, M, N, Dv, device=P.device, dtype=torch.float32)
    pv = pick(prec, "pv")
    if PROB_DTYPE[pv] != P.dtype:
        raise hip.CraftHipError(f"attention probabilities are {P.dtype} but the pv precision expects {PROB_DTYPE[pv]}")
    if vT.dtype != P.dtype:
        raise hip.CraftHipError(f"V^T is {vT.dtype} but P is {P.dtype}")
    if vT.shape[-1] != vt_stride(P):
        raise hip.CraftHipError(f"V^T has key extent {vT.shape[-1]}, attn_apply expects {vt_stride(P)} beside this P (ops.vt_stride)")
    call("craft_attn_apply", P, ldp, getattr(P, "craft_rowsum", None), vT, B, N, M, Dv, out,
         pv | (rows32 << hip.PV_ROWS_SHIFT) | (hip.P_TILED if tiled else 0))
    return out


def mode_pool_ln(O: torch.Tensor, x: torch.Tensor, w_agg: torch.Tensor, skip_coeff: torch.Tensor,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    B, M, N, C = O.shape
    _check_rows(x)
    if out is None:
        out = torch.empty(B, N, C, device=O.device, dtype=torch.float32)
    _check_rows(out)
    call("craft_mode_pool_ln", O, x, _ld(x), w_agg.contiguous().view(-1), skip_coeff, B, N, M, C, out, _ld(out))
    return out


def gma_residual(mf: torch.Tensor, O: torch.Tensor, gamma: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    B, N, C = mf.shape
    if out is None:
        out = torch.empty(B, N, C, device=mf.device, dtype=torch.float32)
    call("craft_gma_residual", mf, _ld(mf), O, gamma, B, N, C, out, _ld(out))
    return out


def convex_upsample(mask: torch.Tensor, flow: torch.Tensor, H8: int, W8: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    B = mask.shape[0]
    up = torch.empty(B, 2, 8 * H8, 8 * W8, device=mask.device, dtype=torch.float32) if out is None else out
    call("craft_convex_upsample", mask, flow, B, H8, W8, up)
    return up


def coords_init(flow_init: Optional[torch.Tensor], B: int, H8: int, W8: int, device):
    N = H8 * W8
    if flow_init is not None:
        if not flow_init.is_cuda:
            raise ValueError("flow_init must be on the GPU")
        if tuple(flow_init.shape) == (1, 2, H8, W8) and B > 1:
            flow_init = flow_init.expand(B, -1, -1, -1)        # the reference's `coords1 + flow_init` broadcasts a batch of 1
        if tuple(flow_init.shape) != (B, 2, H8, W8):
            raise ValueError(f"flow_init must be [{B}, 2, {H8}, {W8}] (1/8 resolution), got {tuple(flow_init.shape)}")
    c0 = torch.empty(B, N, 2, device=device, dtype=torch.float32)
    c1 = torch.empty_like(c0)
    fl = torch.empty_like(c0)
    call("craft_coords_init", None if flow_init is None else flow_init.contiguous().float(), B, H8, W8, c0, c1, fl)
    return c0, c1, fl


def conv2d_tokens(x: torch.Tensor, hw, w_packed: torch.Tensor, bias: torch.Tensor, cout: int, KH: int, KW: int, act: int,
                  prec: int, packed: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Conv2d (stride 1, same padding) + bias (+ReLU) on tokens [B, N, Cin] -> [B, N, Cout]."""
    B, N, Cin = x.shape
    _check_rows(x)
    if out is None:
        out = torch.empty(B, N, cout, device=x.device, dtype=torch.float32)
    call("craft_conv2d_nhwc", x, _ld(x), Cin, w_packed, bias, cout, KH, KW, act, out, _ld(out), B, hw[0], hw[1],
         prec | (hip.W_PACKED if packed else 0))
    return out


# ---- weight packing (layout plumbing, cached by the modules) --------------------------------------
def pack_conv(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, KH, KW] -> [Cout, KH, KW, Cin] contiguous."""
    return w.detach().permute(0, 2, 3, 1).contiguous().float()


class WeightPackRegistry:
    """Every conv-weight operand a training step packs (``pack_conv_weights``), remembered so that the NEXT steps re-pack all of them in ONE
    launch right after the optimizer update (craft_pack_conv_weights_batch) instead of one launch per operand as the forward / backward
    reach them (71 at configs[3]).  Owned by a ``train.Trainer``: it is the active registry only inside that trainer's steps, its entries
    hold the weight tensors (views of the optimizer's flat buffer: addresses are stable), and ``repack()`` is called by the trainer after
    ``optimizer.step()``.  An operand asked for while the registry is not fresh (first step, after a checkpoint load: ``invalidate``) is
    packed on the spot, as before."""

    def __init__(self, storage_ptr: Optional[int] = None):
        self.storage_ptr = storage_ptr      # only operands of weights that live in THIS storage (the optimizer's flat buffer) are remembered:
                                            # a temporary (torch.cat of two weights, a folded BatchNorm) has a new address every step
        self.entries = {}           # key -> [out tensor, (w0, w1) kept alive, fill arguments]
        self.fresh = False          # the outputs hold the CURRENT weights
        self._tables = None         # (jobs device tensor, prefix device tensor, n, blocks) for the current entry set
        self._epoch = None

    def invalidate(self):
        self.fresh = False

    @staticmethod
    def _versions(w0, w1):
        return (w0._version, w1._version if w1 is not None else -1)

    def lookup(self, key):
        """The packed operand if it holds the CURRENT weights: packed after the last raw-pointer update (the optimizer's epoch) AND the
        source tensors untouched since (their autograd version counters: load_state_dict / load_checkpoint / any in-place edit between
        steps moves them -- ADVICE r5)."""
        if self.fresh and self._epoch == hip.weights_epoch():
            e = self.entries.get(key)
            if e is not None and e[3] == self._versions(*e[1]):
                return e[0]
        return None

    def remember(self, key, out, w0, w1, fill_args):
        if key not in self.entries:
            self._tables = None
        self.entries[key] = [out, (w0, w1), fill_args, self._versions(w0, w1)]

    def prepare(self):
        """Build (and validate) the batched job table for the current entry set on the host + one upload; raises before anything of the
        step's update is enqueued (Trainer calls it in front of optimizer.step())."""
        if not self.entries or self._tables is not None:
            return
        import ctypes
        lib = hip.load()
        nbytes = int(lib.craft_pack_conv_job_bytes())
        n = len(self.entries)
        buf = (ctypes.c_ubyte * (nbytes * n))()
        first, blocks = [0], 0
        for i, (out, (w0, w1), fa, _) in enumerate(self.entries.values()):
            cout0, cout1, Cin, KH, KW, a0, a1, b0, b1, transposed, prec = fa
            nb = int(lib.craft_pack_conv_job_fill(ctypes.byref(buf, i * nbytes), ctypes.c_void_p(w0.data_ptr()), cout0,
                                                  ctypes.c_void_p(w1.data_ptr() if w1 is not None else 0), cout1, Cin, KH, KW, a0, a1, b0, b1,
                                                  transposed, prec, ctypes.c_void_p(out.data_ptr())))
            if nb < 0:
                raise hip.CraftHipError(f"craft_pack_conv_job_fill failed with code {-nb}")
            blocks += nb
            first.append(blocks)
        dev = next(iter(self.entries.values()))[0].device
        jobs = torch.frombuffer(bytearray(buf), dtype=torch.uint8).to(dev)
        pref = torch.tensor(first, dtype=torch.int32).to(dev)
        self._tables = (jobs, pref, n, blocks)

    def repack(self):
        """All remembered operands from the current weights, one launch (enqueued on the current stream)."""
        if not self.entries:
            return
        self.prepare()
        jobs, pref, n, blocks = self._tables
        call("craft_pack_conv_weights_batch", jobs, pref, n, blocks)
        for e in self.entries.values():
            e[3] = self._versions(*e[1])
        self.fresh, self._epoch = True, hip.weights_epoch()


class _ActivePacks(threading.local):
    """The registry of the trainer whose step is running ON THIS THREAD (train.Trainer.step), else None: a second trainer or a validation
    forward on another thread sees its own slot (the backward modes of autograd.py are thread-local for the same reason)."""
    reg = None

    def __getitem__(self, i):
        return self.reg

    def __setitem__(self, i, v):
        self.reg = v


ACTIVE_WEIGHT_PACKS = _ActivePacks()


def pack_conv_weights(w0: torch.Tensor, prec: int, w1: Optional[torch.Tensor] = None, sel=None, transposed: bool = False) -> torch.Tensor:
    """nn.Conv2d weights [Cout, Cin, KH, KW] (w0 and optionally w1 concatenated along Cout) -> the MFMA fragment-order operand of the
    conv kernels in ONE launch (craft_pack_conv_weights): input channels ``sel`` = ((a0, a1), (b0, b1)) or None (all); ``transposed``:
    the operand of the input-gradient convolution (rows = selected input channels, taps flipped).  16-bit / f16x3 precisions."""
    w0 = w0.detach()
    w0 = w0 if w0.is_contiguous() else w0.contiguous()
    cout0, Cin, KH, KW = w0.shape
    cout1 = 0
    if w1 is not None:
        w1 = w1.detach()
        w1 = w1 if w1.is_contiguous() else w1.contiguous()
        cout1 = w1.shape[0]
        assert tuple(w1.shape[1:]) == (Cin, KH, KW)
    (a0, a1), (b0, b1) = sel if sel is not None else ((0, Cin), (0, 0))
    nsel, cout = (a1 - a0) + (b1 - b0), cout0 + cout1
    rows, Cp = (nsel, round_up(cout, 32)) if transposed else (cout, round_up(nsel, 32))
    planes = 2 if prec == hip.PREC_F16X3 else 1
    reg = ACTIVE_WEIGHT_PACKS[0]
    key = None
    if (reg is not None and w0.dtype == torch.float32 and (w1 is None or w1.dtype == torch.float32) and w0.is_cuda
            and (reg.storage_ptr is None or (w0.untyped_storage().data_ptr() == reg.storage_ptr
                                             and (w1 is None or w1.untyped_storage().data_ptr() == reg.storage_ptr)))):
        key = (w0.data_ptr(), w1.data_ptr() if w1 is not None else 0, cout0, cout1, Cin, KH, KW, a0, a1, b0, b1, bool(transposed), prec)
        hit = reg.lookup(key)
        if hit is not None:
            return hit                                       # packed by the trainer's batched launch after the last optimizer update
        e = reg.entries.get(key)
        out = e[0] if e is not None else None                # (same buffer every step: the batch job's output address is fixed)
    else:
        out = None
    if out is None:
        out = torch.empty(planes * round_up(rows, 32) * KH * KW * Cp, device=w0.device, dtype=torch.bfloat16 if prec == hip.PREC_BF16 else torch.float16)
    call("craft_pack_conv_weights", w0.float() if w0.dtype != torch.float32 else w0, cout0, w1, cout1, Cin, KH, KW, a0, a1, b0, b1, int(transposed), prec, out)
    if key is not None:
        reg.remember(key, out, w0, w1, (cout0, cout1, Cin, KH, KW, a0, a1, b0, b1, int(transposed), prec))
    return out


def pack_conv_prec(w: torch.Tensor, prec: int) -> torch.Tensor:
    """[Cout, Cin, KH, KW] -> the weight operand the KxK conv kernels take with W_PACKED (craft_pack_weights): fp32 stays
    [Cout, KH, KW, Cin]; bf16 / fp16 / f16x3 become MFMA fragment order [K/32][ceil(Cout/32)][planes][2][64][8] (flat)."""
    if prec != PREC_F32 and w.dtype == torch.float32:
        return pack_conv_weights(w, prec)
    wp = pack_conv(w)
    if prec == PREC_F32:
        return wp
    rows, K = wp.shape[0], wp[0].numel()
    planes = 2 if prec == hip.PREC_F16X3 else 1
    out = torch.empty(planes * round_up(rows, 32) * K, device=w.device, dtype=torch.bfloat16 if prec == hip.PREC_BF16 else torch.float16)
    call("craft_pack_weights", wp, rows, K, prec, out)
    return out


def pack_convf1(w: torch.Tensor) -> torch.Tensor:
    """[128, 2, 7, 7] -> [7*7*2, 128] (tap-major, output channel contiguous)."""
    return w.detach().permute(2, 3, 1, 0).reshape(-1, w.shape[0]).contiguous().float()


def pack_convf1_mfma(w: torch.Tensor, prec: int) -> torch.Tensor:
    """[128, 2, 7, 7] -> the matrix-core operand of convf1: rows = output channels, k = ky*16 + kx*2 + c (kx = 7 and k >= 112
    zero), K = 128, in craft_pack_weights' fragment order for ``prec`` (bf16 / fp16 / f16x3)."""
    co = w.shape[0]
    m = torch.zeros(co, 7, 8, 2, device=w.device, dtype=torch.float32)
    m[:, :, :7, :] = w.detach().float().permute(0, 2, 3, 1)
    m = torch.cat([m.reshape(co, 112), torch.zeros(co, 16, device=w.device)], dim=1).contiguous()
    planes = 2 if prec == hip.PREC_F16X3 else 1
    out = torch.empty(planes * round_up(co, 32) * 128, device=w.device, dtype=torch.bfloat16 if prec == hip.PREC_BF16 else torch.float16)
    call("craft_pack_weights", m, co, 128, prec, out)
    return out
