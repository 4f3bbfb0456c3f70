"""Training path: every hot-path operator as a ``torch.autograd.Function`` whose forward AND backward are HIP kernels.

The reference trains through PyTorch autograd over ATen ops (train.py:228-236, network.py:164-267).  Here autograd only
records the graph (plumbing: views, ``torch.cat`` of token columns, gradient accumulation); all arithmetic is
``libcraft_hip.so`` through the C ABI of ``include/craft_hip.h`` ("training" section):

* dense contractions: ``craft_gemm`` (one general strided batched GEMM on the MFMA engine: dX = dY W, dW = dY^T X with
  split-K, Q K^T, P V, dP = dO V^T, dV = P^T dO, dQ = dS K, dK = dS^T Q), ``craft_conv2d_wgrad``; convolution input
  gradients are the forward convolution with flipped / transposed weights;
* row-wise and element-wise operators and their gradients: LayerNorm tokens, softmax with positional bias / clamp / mask,
  mode pooling, correlation pooling + global LayerNorm + pyramid, bilinear lookup (scatter), GRU gates, convex upsampling,
  dropout (counter-based mask regenerated in the backward), the sequence loss.

Tensors are channels-last fp32 "tokens" ``[B, N, C]`` as in inference.  There is no fallback: a missing extension raises.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
from torch.autograd import Function

from . import hip, ops
from .hip import ACT_NONE, ACT_RELU, ACT_TANH, STATS_REPLICAS, call, pick, round_up


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _rows(t: torch.Tensor) -> torch.Tensor:
    """A tokens tensor usable as a flat [rows, C] matrix: unit channel stride, uniform row stride, 16-byte aligned rows."""
    if t.stride(-1) != 1 or (t.dim() == 3 and t.shape[0] > 1 and t.stride(0) != t.shape[1] * t.stride(1)) or (t.stride(-2) & 3) \
            or (t.data_ptr() & 15):
        return t.contiguous()
    return t


def gemm(A, a_sm, a_sk, a_bs0, a_bs1, B, b_sn, b_sk, b_bs0, b_bs1, C, ldc, c_bs0, c_bs1, zdiv, batch, M, N, K, alpha=1.0,
         accumulate=False, ksplit=1, prec=hip.PREC_F32):
    call("craft_gemm", A, a_sm, a_sk, a_bs0, a_bs1, B, b_sn, b_sk, b_bs0, b_bs1, C, ldc, c_bs0, c_bs1, zdiv, batch, M, N, K,
         float(alpha), int(accumulate), int(ksplit), prec)


# ------------------------------------------------------------------------------------------------------------------
# layout / normalisation / dropout
# ------------------------------------------------------------------------------------------------------------------
class NchwToTokens(Function):
    """NCHW [B, C, H, W] -> tokens [B, H*W, C] (the view/transpose of setrans.py:789); backward: craft_tokens_to_nchw."""

    @staticmethod
    def forward(ctx, x):
        ctx.hw = x.shape[-2:]
        return ops.tokens_from_nchw_wide(_c(x.float())).contiguous()

    @staticmethod
    def backward(ctx, dy):
        return ops.tokens_to_nchw(_rows(dy), *ctx.hw)


class TokensNorm(Function):
    """y = LayerNorm?(act(x)) per token (setrans.py:791-793 comb_norm_layer; network.py:209-212 tanh / relu of the context
    halves).  x may be a column slice of a wider tokens tensor."""

    @staticmethod
    def forward(ctx, x, act, ln):
        x = _rows(x)
        ctx.save_for_backward(x)
        ctx.act, ctx.ln = act, ln
        return ops.tokens_norm(x, act=act, ln=ln)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _rows(dy)
        B, N, C = x.shape
        dx = torch.empty(B, N, C, device=x.device, dtype=torch.float32)
        call("craft_tokens_bwd", x, x.stride(-2), dy, dy.stride(-2), dx, C, B * N, C, ctx.act, int(ctx.ln))
        return dx, None, None


class Dropout(Function):
    """nn.Dropout in training mode with a counter-based mask (craft_dropout): backward = the same call on the gradient."""

    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        x = _c(x)
        y = torch.empty_like(x)
        call("craft_dropout", x, y, x.numel(), float(p), int(seed))
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty_like(dy)
        call("craft_dropout", dy, dx, dy.numel(), float(ctx.p), int(ctx.seed))
        return dx, None, None


def dropout(x, p: float, seed: int):
    return x if p <= 0.0 else Dropout.apply(x, p, seed)


class Act(Function):
    """y = scale * act(x) on tokens (ReLU after the 1x1 convolutions; the 0.25 of the mask head, update.py:161)."""

    @staticmethod
    def forward(ctx, x, act, scale):
        if act not in (ACT_NONE, ACT_RELU) and scale != 1.0:
            raise ValueError("a scaled activation is only defined for none / relu here")
        x = _rows(x)
        y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
        rows, C = x.numel() // x.shape[-1], x.shape[-1]
        call("craft_act_fwd", x, x.stride(-2), y, C, rows, C, act, float(scale))
        ctx.save_for_backward(y if act != ACT_NONE else None)        # relu: sign(y) == sign(act(x)) for scale > 0
        ctx.act, ctx.scale, ctx.shape = act, scale, tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _rows(dy)
        C = ctx.shape[-1]
        rows = dy.numel() // C
        dx = torch.empty(ctx.shape, device=dy.device, dtype=torch.float32)
        call("craft_act_bwd", dy, dy.stride(-2), y, C, dx, C, rows, C, ctx.act, float(ctx.scale))
        return dx, None, None


# ------------------------------------------------------------------------------------------------------------------
# nn.Linear on tokens
# ------------------------------------------------------------------------------------------------------------------
# ------------------------------------------------------------------------------------------------------------------
# Weight gradients of layers that run many times per pass (the update block: 12 iterations).  Autograd would get one freshly
# zero-filled gradient tensor per call and add them up with one elementwise kernel each (~500 tiny kernels per step); instead the
# calls of one pass share ONE accumulation buffer per weight (the weight-gradient kernels add into their output anyway), the
# uses are counted in the forward, and only the last backward call hands the buffer to autograd (the others return None).
# ``cache`` is the per-pass dict the conv operands already live in.
# ------------------------------------------------------------------------------------------------------------------
def _count_use(cache, t):
    if cache is not None and t is not None:
        k = (id(t), "uses")
        cache[k] = cache.get(k, 0) + 1


def _acc_buffer(cache, t, kind, shape, device):
    """(buffer, is_last_use): the pass-wide accumulation buffer for the gradient of tensor t."""
    if cache is None:
        return hip.zeros(shape, device), True
    k = (id(t), kind)
    buf = cache.get(k)
    if buf is None:
        buf = cache[k] = hip.zeros(shape, device)
    return buf, None


def pending_uses(cache) -> int:
    """Forward uses whose backward has not run (0 after a complete backward pass).  A non-zero count after ``loss.backward()`` means
    autograd pruned a call of an accumulated layer -- its weight gradient would be missing -- so callers treat it as an error."""
    return sum(v for k, v in (cache or {}).items() if isinstance(k, tuple) and len(k) == 2 and k[1] == "uses")


def _release_use(cache, t) -> bool:
    """Count one backward call of t; True when it was the last one of this pass (the accumulated gradient is complete)."""
    if cache is None:
        return True
    k = (id(t), "uses")
    cache[k] -= 1
    return cache[k] == 0


class Linear(Function):
    @staticmethod
    def forward(ctx, x, w, b, prec, cache=None):
        x = _rows(x)
        w = _c(w)
        ctx.save_for_backward(x, w)
        ctx.has_bias, ctx.prec = b is not None, prec
        ctx.cache, ctx.wobj = cache, w          # (the Python object: saved_tensors may hand back a different wrapper, and ids key the cache)
        ctx.bw_modes = modes()
        _count_use(cache, w)
        return ops.linear(x, w, b, prec)

    @staticmethod
    def backward(ctx, dy):
        use_modes(ctx.bw_modes)
        x, w = ctx.saved_tensors
        dy = _rows(dy)
        B, N, Cin = x.shape
        Cout = w.shape[0]
        rows = B * N
        pp = pick(ctx.prec, "proj")
        cache, wobj = ctx.cache, ctx.wobj
        last = _release_use(cache, wobj)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(B, N, Cin, device=x.device, dtype=torch.float32)
            # dX = dY . W : A = dY rows (k = cout contiguous), B(n = ci, k = co) = W[co][ci] k-major
            gemm(dy, dy.stride(-2), 1, 0, 0, w, 1, Cin, 0, 0, dx, Cin, 0, 0, 1, 1, rows, Cin, Cout, prec=pp)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        accb = _acc_buffer(cache, wobj, "db", (Cout,), x.device)[0] if want_db else None
        pk = _use_pk(pp) and ctx.needs_input_grad[1] and Cin % 4 == 0 and Cout % 4 == 0
        if pk:
            co_p, ci_p = round_up(Cout, 32), round_up(Cin, 32)
            acc, _ = _acc_buffer(cache, wobj, "dw", (co_p, ci_p), x.device)
            _wgrad_deferred(cache, wobj, last, (Packed(dy, gprec(pp), colsum=accb), Packed(x, xprec(pp))), 1, 1, acc)
            dw = acc[:Cout, :Cin] if last else None
        elif ctx.needs_input_grad[1]:
            acc, _ = _acc_buffer(cache, wobj, "dw", (Cout, Cin), x.device)
            # dW = dY^T . X : both operands k-major over the rows, split-K
            gemm(dy, 1, dy.stride(-2), 0, 0, x, 1, x.stride(-2), 0, 0, acc, Cin, 0, 0, 1, 1, Cout, Cin, rows, accumulate=True, ksplit=0, prec=pp)
            dw = acc if last else None
        if want_db:
            if not pk:
                call("craft_colsum", dy, dy.stride(-2), rows, Cout, accb)
            db = accb if last else None
        return dx, dw, db, None, None


# ------------------------------------------------------------------------------------------------------------------
# attention: scores, softmax, apply, mode pooling
# ------------------------------------------------------------------------------------------------------------------
class Scores(Function):
    """S[b][m] = scale * Q_m K_m^T (setrans.py:507-515), materialised fp32 [B, M, N, ld] (ld = N rounded up to 32)."""

    @staticmethod
    def forward(ctx, q, k, M, scale, prec, link=None):
        """link (a ScoreLink shared with the AttnSoftmax that consumes S directly): the three products run on packed operands
        (craft_gemm_pk) and the softmax backward hands dS over as a pack instead of an fp32 tensor."""
        q, k = _rows(q), _rows(k)
        B, N, C = q.shape
        d = C // M
        ld = round_up(N, 32)
        S = torch.empty(B, M, N, ld, device=q.device, dtype=torch.float32)
        sp = pick(prec, "score")
        ctx.M, ctx.scale, ctx.prec, ctx.link = M, scale, sp, None
        bwd = getattr(prec, "sbw", None)
        if link is not None and sp == hip.PREC_F16X3 and bwd in (hip.PREC_F16, hip.PREC_BF16) and d % 32 == 0:
            # hybrid (policy role sbw): the scores themselves in f16x3 on the fp32-source engine (forward parity with inference), their
            # GRADIENT products dQ = dS K, dK = dS^T Q on single-plane packs -- dS leaves the softmax backward as a 16-bit pack (half the
            # bytes of the fp32 tensor) and the two products run on craft_gemm_pk
            gemm(q, q.stride(-2), 1, N * q.stride(-2), d, k, k.stride(-2), 1, N * k.stride(-2), d, S, ld, M * N * ld, N * ld, M, B * M,
                 N, N, d, alpha=scale, prec=sp)
            qpk = PkMat(B, N, C, bwd, q.device).fill(q)
            kpk = qpk if k is q else PkMat(B, N, C, bwd, q.device).fill(k)
            link.want, link.prec = True, bwd
            ctx.prec = bwd
            ctx.link, ctx.packs, ctx.dims = link, (qpk, kpk), (B, N, C)
            return S
        if link is not None and sp != hip.PREC_F32 and sp != hip.PREC_F16X3 and d % 32 == 0:
            qpk = PkMat(B, N, C, sp, q.device).fill(q)                           # rows (b, i), channels (m, d)
            kpk = qpk if k is q else PkMat(B, N, C, sp, q.device).fill(k)
            cg = d // 32
            gemm_pk(qpk, qpk.desc(PK_CH, 1, 0, 0, 0, cg), kpk, kpk.desc(PK_CH, 1, 0, 0, 0, cg), S, ld, M * N * ld, N * ld, M, B * M, N, N, d, alpha=scale)
            link.want, link.prec = True, sp
            ctx.link, ctx.packs, ctx.dims = link, (qpk, kpk), (B, N, C)
            return S
        gemm(q, q.stride(-2), 1, N * q.stride(-2), d, k, k.stride(-2), 1, N * k.stride(-2), d, S, ld, M * N * ld, N * ld, M, B * M,
             N, N, d, alpha=scale, prec=sp)
        ctx.save_for_backward(q, k)
        return S

    @staticmethod
    def backward(ctx, dS):
        if ctx.link is not None:
            link, (qpk, kpk), (B, N, C) = ctx.link, ctx.packs, ctx.dims
            M, d = ctx.M, C // ctx.M
            cg = d // 32
            dspk = link.dS
            if dspk is None:                                                     # the consumer did not hand a pack over: pack here
                dspk = PkMat(B * M, N, dS.shape[-1], ctx.prec, dS.device).fill(_c(dS))
            dq = torch.empty(B, N, C, device=dS.device, dtype=torch.float32)
            dk = torch.empty(B, N, C, device=dS.device, dtype=torch.float32)
            gemm_pk(dspk, dspk.desc(PK_CH, M, 1), kpk, kpk.desc(PK_ROWS, 1, 0, 0, 0, cg), dq, C, N * C, d, M, B * M, N, d, N, alpha=ctx.scale)
            gemm_pk(dspk, dspk.desc(PK_ROWS, M, 1), qpk, qpk.desc(PK_ROWS, 1, 0, 0, 0, cg), dk, C, N * C, d, M, B * M, N, d, N, alpha=ctx.scale)
            link.dS = ctx.packs = None
            return dq, dk, None, None, None, None
        q, k = ctx.saved_tensors
        dS = _c(dS)
        B, N, C = q.shape
        M, d = ctx.M, C // ctx.M
        ld = dS.shape[-1]
        dq = torch.empty(B, N, C, device=q.device, dtype=torch.float32)
        dk = torch.empty(B, N, C, device=q.device, dtype=torch.float32)
        # dQ_m = scale * dS_m . K_m : A = dS rows (k = key j), B(n = c, k = j) = K[b][j][m*d + c] k-major
        gemm(dS, ld, 1, M * N * ld, N * ld, k, 1, k.stride(-2), N * k.stride(-2), d, dq, C, N * C, d, M, B * M, N, d, N, alpha=ctx.scale,
             prec=ctx.prec)
        # dK_m = scale * dS_m^T . Q_m : A(m = j, k = i) = dS[i][j] k-major, B(n = c, k = i) = Q[b][i][m*d + c] k-major
        gemm(dS, 1, ld, M * N * ld, N * ld, q, 1, q.stride(-2), N * q.stride(-2), d, dk, C, N * C, d, M, B * M, N, d, N, alpha=ctx.scale,
             prec=ctx.prec)
        return dq, dk, None, None, None, None


class ScoreLink:
    """Side channel between a Scores node and the AttnSoftmax that consumes its output directly: when the scores were formed on packed
    operands (``want``), the softmax backward writes dS as a pack of mode ``prec`` into ``dS`` and returns a zero-stride placeholder."""
    __slots__ = ("want", "prec", "dS")

    def __init__(self):
        self.want, self.prec, self.dS = False, 0, None


# probability-gradient buffers that a backward of THIS module just allocated for AttnSoftmax.backward (AttnApply / ProbsToken) are TAGGED
# as tensor objects (an attribute on the returned tensor; PyTorch keeps a tensor's Python object alive across the autograd engine): only a
# tagged gradient may be overwritten in place with dS.  A gradient that arrives from anywhere else -- a caller-held tensor, a new sum the
# engine formed, a tensor that merely reuses a freed address -- carries no tag and is copied first.
def _fresh_dp(t):
    t._craft_fresh_dp = True
    return t


def _take_fresh_dp(t):
    """True exactly once for a tensor tagged by _fresh_dp (the tag is consumed)."""
    if t.is_contiguous() and getattr(t, "_craft_fresh_dp", False):
        t._craft_fresh_dp = False
        return True
    return False


class AttnSoftmax(Function):
    """P = softmax_j(clamp?(S) + pos_w pb + mask) in place on S (setrans.py:520-551); with drop_p > 0 the dropout of the probabilities
    (setrans.py:553-557) in the same two kernels: the forward returns the dropped copy (P itself stays in S for the backward), the
    backward applies the mask to the incoming gradient as it reads it -- no separate pass over the [B, M, N, N] tensor either way."""

    @staticmethod
    def forward(ctx, S, pos_tab, pos_w, mask_radius, clamp_ord, hw, drop_p=0.0, seed=0, pk=None, link=None):
        """pk (a PkMat of B*M batches x N rows x ld channels): the (dropped) probabilities are written as that packed operand INSTEAD of a
        dropped fp32 tensor, and the return value is only the autograd handle of P (its data: the undropped P) -- the consumers
        (AttnApply, ProbsToken / train_update) multiply with the pack."""
        B, M, N, ld = S.shape
        R = (pos_tab.shape[0] - 1) // 2 if pos_tab is not None else 0          # None: plain softmax (gma.Attention, gma.py:96-98)
        bits = torch.empty(B * M * N * (ld // 32), device=S.device, dtype=torch.int32)
        tab = _c(pos_tab.detach()) if pos_tab is not None else None
        out = torch.empty_like(S) if (drop_p > 0.0 and pk is None) else None
        call("craft_attn_softmax_fwd", S, ld, B, M, hw[0], hw[1], tab, R, float(pos_w), int(mask_radius), clamp_ord, bits,
             out, float(drop_p), int(seed), pk.buf if pk is not None else None, pk.rows_total if pk is not None else 0,
             pk.np_ if pk is not None else 0, pk.prec if pk is not None else 0)
        ctx.hw, ctx.R, ctx.pos_w, ctx.has_tab, ctx.drop = hw, R, pos_w, pos_tab is not None, (float(drop_p), int(seed))
        ctx.link = link if (link is not None and link.want) else None
        ctx.save_for_backward(S, bits, clamp_ord)
        if out is None:
            ctx.mark_dirty(S)
            return S
        return out                                     # S (this function's scratch by contract) keeps P for the backward

    @staticmethod
    def backward(ctx, dP):
        P, bits, clamp_ord = ctx.saved_tensors
        B, M, N, ld = P.shape
        # dS over the incoming gradient when it is the fresh output of the one consumer of P (AttnApply / ProbsToken: nothing else
        # holds it, 1 GB saved at configs[3]); any other gradient tensor is copied
        if _take_fresh_dp(dP):
            dS = dP
        else:
            dS = dP.clone(memory_format=torch.contiguous_format)
        T = 2 * ctx.R + 1
        rep = hip.zeros((STATS_REPLICAS, T * T,), P.device) if ctx.has_tab else None
        dspk = PkMat(B * M, N, ld, ctx.link.prec, P.device) if ctx.link is not None else None
        call("craft_attn_softmax_bwd", P, dS, ld, B, M, ctx.hw[0], ctx.hw[1], ctx.R, float(ctx.pos_w), clamp_ord, bits, rep,
             ctx.drop[0], ctx.drop[1], dspk.buf if dspk is not None else None, dspk.rows_total if dspk is not None else 0,
             dspk.np_ if dspk is not None else 0, dspk.prec if dspk is not None else 0)
        if dspk is not None:
            ctx.link.dS = dspk
            dS = dS.new_zeros(1).expand(B, M, N, ld)            # (a placeholder of the right shape: Scores.backward takes the pack)
        dtab = None
        if ctx.has_tab:
            dtab = hip.zeros((T, T,), P.device)
            call("craft_reduce_replicas", rep, STATS_REPLICAS, T * T, dtab)
        return dS, dtab, None, None, None, None, None, None, None, None


class RelPosAdd(Function):
    """S += w (Hs (+) Ws): the relative-position scores of gma.Attention (RelPosEmb, gma.py:21-50, :84-98) added to the materialised
    scores in place.  Hs [B, heads, N, nh >= 2 H8 - 1] / Ws [.., nw >= 2 W8 - 1]: q . E_h / q . E_w of the offsets that can occur."""

    @staticmethod
    def forward(ctx, S, Hs, Ws, w, hw):
        B, Mh, N, ld = S.shape
        Hs, Ws = _c(Hs), _c(Ws)
        call("craft_relpos_add", S, ld, B * Mh, hw[0], hw[1], Hs, Hs.shape[-1], Ws, Ws.shape[-1], float(w))
        ctx.mark_dirty(S)
        ctx.w, ctx.hw, ctx.nh, ctx.nw = float(w), hw, Hs.shape[-1], Ws.shape[-1]
        return S

    @staticmethod
    def backward(ctx, dS):
        dS = _c(dS)
        B, Mh, N, ld = dS.shape
        dHs = torch.empty(B, Mh, N, ctx.nh, device=dS.device, dtype=torch.float32)
        dWs = torch.empty(B, Mh, N, ctx.nw, device=dS.device, dtype=torch.float32)
        call("craft_relpos_bwd", dS, ld, B * Mh, ctx.hw[0], ctx.hw[1], dHs, ctx.nh, ctx.nh, dWs, ctx.nw, ctx.nw, ctx.w)
        return dS, dHs, dWs, None, None


class AttnApply(Function):
    """O[b][m] = P[b][m] V_m with V = first_linear(x) [B, N, M*C] in its natural layout (setrans.py:373-384)."""

    @staticmethod
    def forward(ctx, P, v, prec, pk=None):
        """pk: P as a packed operand (AttnSoftmax(pk=...)): the three products run on craft_gemm_pk and P's own data is not read."""
        v = _rows(v)
        B, M, N, ld = P.shape
        C = v.shape[-1] // M
        O = torch.empty(B, M, N, C, device=P.device, dtype=torch.float32)
        pv = pick(prec, "pv")
        ctx.pk = pk
        if pk is not None:
            cg = C // 32
            vpk = PkMat(B, N, M * C, pk.prec, P.device).fill(v)                    # rows j, channels (m, c)
            gemm_pk(pk, pk.desc(PK_CH, M, 1), vpk, vpk.desc(PK_ROWS, 1, 0, 0, 0, cg), O, C, M * N * C, N * C, M, B * M, N, C, ld)
            ctx.vpk, ctx.dims = vpk, (B, M, N, ld, C)
            return O
        ldv = v.stride(-2)
        gemm(P, ld, 1, M * N * ld, N * ld, v, 1, ldv, N * ldv, C, O, C, M * N * C, N * C, M, B * M, N, C, N, prec=pv)
        ctx.save_for_backward(P, v)
        ctx.prec = pv
        return O

    @staticmethod
    def backward(ctx, dO):
        dO = _c(dO)
        if ctx.pk is not None:
            pk, vpk = ctx.pk, ctx.vpk
            B, M, N, ld, C = ctx.dims
            cg = C // 32
            dopk = PkMat(B * M, N, C, pk.prec, dO.device).fill(dO)                  # rows i, channels c
            dP = dv = None
            if ctx.needs_input_grad[0]:
                dP = torch.empty(B, M, N, ld, device=dO.device, dtype=torch.float32)   # (columns >= N: never read by the softmax backward)
                gemm_pk(dopk, dopk.desc(PK_CH, M, 1), vpk, vpk.desc(PK_CH, 1, 0, 0, 0, cg), dP, ld, M * N * ld, N * ld, M, B * M, N, N, C)
            if ctx.needs_input_grad[1]:
                dv = torch.empty(B, N, M * C, device=dO.device, dtype=torch.float32)
                gemm_pk(pk, pk.desc(PK_ROWS, M, 1), dopk, dopk.desc(PK_ROWS, M, 1), dv, M * C, N * M * C, C, M, B * M, N, C, N)
            ctx.pk = ctx.vpk = None
            return (_fresh_dp(dP) if dP is not None else None), dv, None, None
        P, v = ctx.saved_tensors
        B, M, N, ld = P.shape
        C = v.shape[-1] // M
        ldv = v.stride(-2)
        dP = dv = None
        if ctx.needs_input_grad[0]:
            dP = torch.zeros(B, M, N, ld, device=P.device, dtype=torch.float32) if ld != N else torch.empty(B, M, N, ld, device=P.device)
            # dP_m = dO_m . V_m^T : A = dO rows (k = c), B(n = j, k = c) = V[b][j][m*C + c] rows
            gemm(dO, C, 1, M * N * C, N * C, v, ldv, 1, N * ldv, C, dP, ld, M * N * ld, N * ld, M, B * M, N, N, C, prec=ctx.prec)
        if ctx.needs_input_grad[1]:
            dv = torch.empty(B, N, M * C, device=P.device, dtype=torch.float32)
            # dV_m = P_m^T . dO_m : A(m = j, k = i) = P[i][j] k-major, B(n = c, k = i) = dO[i][c] k-major
            gemm(P, 1, ld, M * N * ld, N * ld, dO, 1, C, M * N * C, N * C, dv, M * C, N * M * C, C, M, B * M, N, C, N, prec=ctx.prec)
        return (_fresh_dp(dP) if dP is not None else None), dv, None, None


class SharedProbs:
    """Attention probabilities that several AttnApplyShared calls consume (the intra-frame attention: computed once, applied in
    every refinement iteration, network.py:214 / update.py:143-149) and what their backward passes leave behind for the ONE
    gradient product of P."""

    def __init__(self, P: torch.Tensor, pk=None):
        self.P = P.detach()
        self.pk = pk                       # P as a packed operand (then P's own data is the UNdropped P: only the pack may be multiplied)
        self.pending: List = []            # (dO_t [B, M, N, C], v_t [B, N, M*C]) of every use, in backward order
        self.cat = None                    # or: the concatenations ([B, M, N, T*C] dO, V) already built by train_update._phase2 -- tensors, or
                                           # (with pk) the packs (dO: rows (b, m, i) x channels (t, c); V: rows (b, j) x channels (m, t, c))


class ProbsToken(Function):
    """P -> a one-element handle.  Every AttnApplyShared takes the handle as an input, so this backward runs after all of them.
    dP = sum_t dO_t V_t^T is then ONE product with the T uses concatenated along K ([dO_1 .. dO_T] . [V_1 .. V_T]^T, K = T*C):
    the 1 GB gradient of P is written once instead of being produced, zero-filled and accumulated T times."""

    @staticmethod
    def forward(ctx, P, box, prec, pk=None):
        holder = SharedProbs(P, pk)
        box.append(holder)
        ctx.holder, ctx.prec = holder, pick(prec, "pv")
        return hip.zeros((1,), P.device)

    @staticmethod
    def backward(ctx, _dtoken):
        holder = ctx.holder
        P = holder.P
        B, M, N, ld = P.shape
        if holder.pk is not None:
            if holder.cat is None:
                raise RuntimeError("ProbsToken: packed probabilities are consumed by train_update only (it leaves the packed dO / V behind)")
            dO, V = holder.cat
            dP = torch.empty(B, M, N, ld, device=P.device, dtype=torch.float32)     # (columns >= N: never read by the softmax backward)
            K = dO.ncg * 32
            gemm_pk(dO, dO.desc(PK_CH, M, 1), V, V.desc(PK_CH, 1, 0, 0, 0, K // 32), dP, ld, M * N * ld, N * ld, M, B * M, N, N, K)
            holder.cat = holder.pk = None
            return _fresh_dp(dP), None, None, None
        dP = torch.zeros(B, M, N, ld, device=P.device, dtype=torch.float32) if ld != N else torch.empty_like(P)
        if holder.cat is not None:
            dO, V = holder.cat
            K = dO.shape[-1]
            gemm(dO, K, 1, M * N * K, N * K, V, K, 1, M * N * K, N * K, dP, ld, M * N * ld, N * ld, M, B * M, N, N, K, prec=ctx.prec)
            holder.cat = None
        elif holder.pending:
            T = len(holder.pending)
            C = holder.pending[0][0].shape[-1]
            dO = torch.cat([d for d, _ in holder.pending], dim=-1)                                   # [B, M, N, T*C]
            V = torch.cat([v.view(B, N, M, C).permute(0, 2, 1, 3) for _, v in holder.pending], dim=-1)   # [B, M, N, T*C]
            K = T * C
            gemm(dO, K, 1, M * N * K, N * K, V, K, 1, M * N * K, N * K, dP, ld, M * N * ld, N * ld, M, B * M, N, N, K, prec=ctx.prec)
            holder.pending = []
        elif ld == N:
            dP.zero_()
        return _fresh_dp(dP), None, None, None


class AttnApplyShared(Function):
    """O[b][m] = P[b][m] V_m for a shared P (see ProbsToken): its dO / V pair is queued for the deferred dP product."""

    @staticmethod
    def forward(ctx, token, v, holder, prec):
        v = _rows(v)
        P = holder.P
        B, M, N, ld = P.shape
        C = v.shape[-1] // M
        O = torch.empty(B, M, N, C, device=P.device, dtype=torch.float32)
        pv = pick(prec, "pv")
        ldv = v.stride(-2)
        gemm(P, ld, 1, M * N * ld, N * ld, v, 1, ldv, N * ldv, C, O, C, M * N * C, N * C, M, B * M, N, C, N, prec=pv)
        ctx.save_for_backward(v)
        ctx.holder, ctx.prec = holder, pv
        return O

    @staticmethod
    def backward(ctx, dO):
        (v,) = ctx.saved_tensors
        holder = ctx.holder
        P = holder.P
        dO = _c(dO)
        B, M, N, ld = P.shape
        C = v.shape[-1] // M
        holder.pending.append((dO, v if v.is_contiguous() else v.contiguous()))
        dv = torch.empty(B, N, M * C, device=P.device, dtype=torch.float32)
        gemm(P, 1, ld, M * N * ld, N * ld, dO, 1, C, M * N * C, N * C, dv, M * C, N * M * C, C, M, B * M, N, C, N, prec=ctx.prec)
        return hip.zeros((1,), P.device), dv, None, None


class GmaResidual(Function):
    """gma.Aggregate.forward's tail (gma.py:138): out = mf + gamma * O."""

    @staticmethod
    def forward(ctx, mf, O, gamma):
        mf, O = _rows(mf), _c(O)
        ctx.save_for_backward(O, gamma)
        return ops.gma_residual(mf, O, gamma)

    @staticmethod
    def backward(ctx, dy):
        O, gamma = ctx.saved_tensors
        dy = _c(dy)
        B, N, C = dy.shape
        dO = dg = None
        if ctx.needs_input_grad[1]:                      # gamma * dy = dy + (gamma - 1) * dy: the forward kernel, no host read of gamma
            dO = ops.gma_residual(dy, dy, (gamma.detach() - 1.0).contiguous())
        if ctx.needs_input_grad[2]:                      # <dy, O>: a 1 x 1 product over K = all elements, split-K
            K = B * N * C
            acc = hip.zeros((1, 1,), dy.device)
            gemm(dy, K, 1, 0, 0, O, K, 1, 0, 0, acc, 1, 0, 0, 1, 1, 1, 1, K, accumulate=True, ksplit=0, prec=hip.PREC_F16X3)
            dg = acc.view(gamma.shape)
        return dy, dO, dg


class ModePoolLN(Function):
    """y = LayerNorm(skip * x + sum_m softmax_m(<O_m, w>) O_m)  (setrans.py:395-407)."""

    @staticmethod
    def forward(ctx, O, x, w_agg, skip):
        O, x = _c(O), _rows(x)
        ctx.save_for_backward(O, x, w_agg, skip)
        return ops.mode_pool_ln(O, x, w_agg.detach(), skip.detach())

    @staticmethod
    def backward(ctx, dy):
        O, x, w_agg, skip = ctx.saved_tensors
        dy = _rows(dy)
        B, M, N, C = O.shape
        dO = torch.empty_like(O)
        dx = torch.empty(B, N, C, device=O.device, dtype=torch.float32)
        rep = hip.zeros((STATS_REPLICAS, C + 1,), O.device)
        call("craft_mode_pool_ln_bwd", O, x, x.stride(-2), _c(w_agg.detach()).view(-1), skip.detach(), dy, dy.stride(-2), B, N, M, C,
             dO, dx, C, rep)
        red = hip.zeros((C + 1,), O.device)
        call("craft_reduce_replicas", rep, STATS_REPLICAS, C + 1, red)
        return dO, dx, red[:C].reshape(w_agg.shape), red[C:].reshape(skip.shape)


# ------------------------------------------------------------------------------------------------------------------
# correlation volume, pyramid, lookup
# ------------------------------------------------------------------------------------------------------------------
class TrainPyramid:
    """The pyramid of one forward pass plus the shared gradient buffers its lookups scatter into."""

    def __init__(self, pyr: ops.CorrPyramid):
        self.pyr = pyr
        self.G: Optional[List[torch.Tensor]] = None

    def grads(self):
        if self.G is None:
            self.G = [torch.zeros_like(t) for t in self.pyr.lv]
        return self.G


class CorrVolume(Function):
    """Scores [B, M, N, ld] -> pooled, globally normalised 4-level pyramid (corr.py:186-204; setrans.py:520-550).  Returns the
    (mean, rstd) tensor as the autograd handle of the pyramid: every lookup takes it as an input, so this backward runs
    after all of them and finds their scattered gradients in ``holder.grads()``."""

    @staticmethod
    def forward(ctx, S, pos_tab, w_aggr, pos_w, clamp_ord, hw, holder_box, do_norm):
        B, M, N, ld = S.shape
        H8, W8 = hw
        R = (pos_tab.shape[0] - 1) // 2 if pos_tab is not None else 0          # None: no positional bias (CorrBlock.corr, corr.py:73-81)
        pyr = ops.CorrPyramid(B, H8, W8, 4, S.device)
        tab = _c(pos_tab.detach()) if pos_tab is not None else None
        wv = _c(w_aggr.detach()).view(-1)
        call("craft_corr_pool_fwd", S, ld, B, M, H8, W8, tab, R, float(pos_w), wv, clamp_ord, pyr.lv[0], pyr.sums)
        call("craft_corr_finish", pyr.lv[0], pyr.lv[1], pyr.lv[2], pyr.lv[3], pyr.sums, pyr.mu_rstd, B, H8, W8, int(do_norm))
        holder = TrainPyramid(pyr)
        holder_box.append(holder)
        ctx.holder, ctx.hw, ctx.R, ctx.pos_w, ctx.do_norm, ctx.has_tab = holder, hw, R, pos_w, do_norm, pos_tab is not None
        ctx.save_for_backward(S, tab, wv, clamp_ord)
        ctx.w_shape = w_aggr.shape
        return pyr.mu_rstd

    @staticmethod
    def backward(ctx, _dtoken):
        S, tab, wv, clamp_ord = ctx.saved_tensors
        B, M, N, ld = S.shape
        H8, W8 = ctx.hw
        pyr = ctx.holder.pyr
        G = ctx.holder.grads()
        gstats = hip.zeros((B, 2,), S.device, torch.float64)
        call("craft_corr_pyramid_bwd", G[0], G[1], G[2], G[3], pyr.lv[0], pyr.mu_rstd, B, H8, W8, gstats)
        T = 2 * ctx.R + 1
        rep = hip.zeros((STATS_REPLICAS, T * T,), S.device) if ctx.has_tab else None
        dw = hip.zeros((1,), S.device, torch.float64)
        dS = S                                                    # the scores are dead after this: overwritten with dS
        call("craft_corr_pool_bwd", dS, ld, B, M, H8, W8, tab, ctx.R, float(ctx.pos_w), wv, clamp_ord, pyr.lv[0], G[0], pyr.mu_rstd, gstats,
             int(ctx.do_norm), rep, dw)
        dtab = None
        if ctx.has_tab:
            dtab = hip.zeros((T, T,), S.device)
            call("craft_reduce_replicas", rep, STATS_REPLICAS, T * T, dtab)
        # the pyramid's (mean, rstd) tensor is this node's OUTPUT: node -> holder -> pyramid -> output -> node is a reference cycle that
        # kept 350 MB per step alive until the cyclic collector ran; the volume is finished here
        ctx.holder.G = None
        ctx.holder.pyr = None
        ctx.holder = None
        return dS, dtab, (dw.float().reshape(ctx.w_shape) if ctx.needs_input_grad[2] else None), None, None, None, None, None


class CorrLookup(Function):
    """corr tokens [B, N, V*4*81] = bilinear samples of the normalised pyramid(s) around coords (corr.py:47-71).  coords carry no
    gradient (network.py:232 detaches them); the volume's gradient is scattered into the holder's buffers.  ``holder``: one
    TrainPyramid, or the list of the V = 2 volumes of the two-way correlation of --f1 (corr.py:164-171: every level holds
    [volume 0 window | volume 1 window]); ``token`` is then the sum of their tokens, so that every volume's backward runs after this."""

    @staticmethod
    def forward(ctx, token, coords, holder, radius):
        holders = list(holder) if isinstance(holder, (list, tuple)) else [holder]
        ctx.holders, ctx.radius = holders, radius
        coords = _c(coords.detach())
        ctx.save_for_backward(coords)
        return ops.corr_lookup([h.pyr for h in holders], coords, radius)

    @staticmethod
    def backward(ctx, dout):
        (coords,) = ctx.saved_tensors
        dout = _rows(dout)
        lookup_bwd(ctx.holders, dout, coords, ctx.radius)
        return hip.zeros((ctx.holders[0].pyr.B, 2,), dout.device), None, None, None


def lookup_bwd(holders, dout, coords, radius):
    """Gradient of the lookup output into the shared gradient buffers of the (normalised) pyramids of every volume."""
    V, win2 = len(holders), (2 * radius + 1) ** 2
    for v, h in enumerate(holders):
        pyr, G = h.pyr, h.grads()
        call("craft_corr_lookup_bwd", dout, dout.stride(-2), coords, G[0], G[1], G[2], G[3], pyr.levels, pyr.B, pyr.H8, pyr.W8, radius,
             win2 * V if V > 1 else 0, win2 * v)


# ------------------------------------------------------------------------------------------------------------------
# convolutions (stride 1, same padding) on tokens
# ------------------------------------------------------------------------------------------------------------------
def _pad_cols(x: torch.Tensor, c: int) -> torch.Tensor:
    """tokens [B, N, C] -> [B, N, c] with zero columns appended (channel counts of the conv engine are multiples of 32)."""
    if x.shape[-1] == c:
        return _rows(x)
    y = torch.zeros(*x.shape[:-1], c, device=x.device, dtype=torch.float32)
    y[..., : x.shape[-1]] = x
    return y


_ZERO_ROWS = {}


def zero_row(n: int, device) -> torch.Tensor:
    """A READ-ONLY row of >= n zeros that lives as long as the process (the absent bias of the input-gradient convolutions: a
    torch.zeros per layer and step was 30 fill kernels of a training step)."""
    key = (str(device), )
    z = _ZERO_ROWS.get(key)
    if z is None or z.numel() < n:
        z = _ZERO_ROWS[key] = torch.zeros(max(4096, n), device=device, dtype=torch.float32)
    return z[:n]


def _conv_weights(w, b, cp, cache, transposed: bool):
    """Operands of the conv kernels for one weight tensor, built once per forward pass (``cache``: a dict that lives as long as
    the pass' graph -- the refinement loop calls every layer 12 times with the same weights, forward and backward):
    raw [cout_p][KH][KW][cin_p] (channel counts padded to multiples of 32), the padded bias, and for 16-bit / f16x3 precisions
    and kernels up to 5x5 the MFMA fragment-order packing the halo kernel streams (craft_pack_weights).
    ``transposed``: the operand of the INPUT-gradient convolution, W'[ci][KH-1-ky][KW-1-kx][co] = W[co][ky][kx][ci]."""
    key = (id(w), cp, transposed)
    hit = cache.get(key) if cache is not None else None
    if hit is not None:
        return hit
    Cout, Cin, KH, KW = w.shape
    cin_p, cout_p = round_up(Cin, 32), round_up(Cout, 32)
    if cp != hip.PREC_F32 and KH * KW > 1 and KH <= 5 and KW <= 5:
        # fragment-order operand in one launch, straight from the nn.Conv2d layout (padding, flip and transposition included)
        wt = ops.pack_conv_weights(w, cp, transposed=transposed)
        rows = cin_p if transposed else cout_p
        if b is not None and not transposed:
            bias = torch.zeros(rows, device=w.device, dtype=torch.float32)
            bias[:Cout] = b.detach()
        else:
            bias = zero_row(rows, w.device)
        out = (wt, bias, hip.W_PACKED, None)
        if cache is not None:
            cache[key] = out
        return out
    raw = cache.get((id(w), "raw")) if cache is not None else None
    if raw is None:
        raw = torch.zeros(cout_p, KH, KW, cin_p, device=w.device, dtype=torch.float32)
        raw[:Cout, :, :, :Cin] = w.detach().permute(0, 2, 3, 1)
        if cache is not None:
            cache[(id(w), "raw")] = raw
    if transposed:
        wt = raw.flip(1, 2).permute(3, 1, 2, 0).contiguous()
        rows, bias = cin_p, torch.zeros(cin_p, device=w.device, dtype=torch.float32)
    else:
        wt = raw
        rows, bias = cout_p, torch.zeros(cout_p, device=w.device, dtype=torch.float32)
        if b is not None:
            bias[:Cout] = b.detach()
    flag = 0
    if cp != hip.PREC_F32 and KH * KW > 1 and KH <= 5 and KW <= 5:
        planes = 2 if cp == hip.PREC_F16X3 else 1
        K = wt[0].numel()
        packed = torch.empty(planes * rows * K, device=w.device, dtype=torch.bfloat16 if cp == hip.PREC_BF16 else torch.float16)
        call("craft_pack_weights", wt, rows, K, cp, packed)
        wt, flag = packed, hip.W_PACKED
    out = (wt, bias, flag, raw)
    if cache is not None:
        cache[key] = out
    return out


class Packed:
    """A tokens tensor as a packed MFMA operand (craft_pack_operand): 16-bit planes [plane][C/32][rows_p][32] over a zero-padded
    pixel grid, so that a convolution tap is one row shift and the weight-gradient K loop is a pure copy (kernels_gemm_pk.hip)."""
    __slots__ = ("buf", "rows_p", "guard", "K", "C_p", "prec", "Wp", "rows", "C")

    def __init__(self, x, prec: int, spatial=None, colsum=None, batch=None):
        """x: tokens [.., C] (unit channel stride, uniform row stride) or a LIST of such tensors over the same rows -- their channel
        concatenation (every source but the last with C % 32 == 0), packed without a torch.cat; spatial = (B, H, W, padH, padW) or
        None (plain rows); colsum: [C] buffer that receives += the column sums (single source only).  ``batch`` (a PackBatch): the
        pack is queued instead of launched -- ``batch.flush()`` packs everything queued in ONE launch (craft_pack_operands)."""
        srcs = list(x) if isinstance(x, (list, tuple)) else [x]
        C = sum(t.shape[-1] for t in srcs)
        rows = srcs[0].numel() // srcs[0].shape[-1]
        self.rows, self.C = rows, C
        tail = 0
        if spatial is not None:
            B, H, W, ph, pw = spatial
            grid = B * (H + 2 * ph) * (W + 2 * pw)
            self.Wp = W + 2 * pw
            self.guard = ph * self.Wp + pw
        else:
            B = H = W = ph = pw = 0
            grid, self.Wp, self.guard = rows, 1, 0
        self.K = round_up(grid, 32)
        self.rows_p = round_up(2 * self.guard + self.K, 64)
        self.C_p, self.prec = sum(round_up(t.shape[-1], 32) for t in srcs), prec
        planes = 2 if prec == hip.PREC_F16X3 else 1
        ncg = self.C_p // 32
        self.buf = torch.empty(planes * ncg * self.rows_p * 32, device=srcs[0].device, dtype=torch.int16)
        off = 0
        for t in srcs:
            c = t.shape[-1]
            if t is not srcs[-1] and c % 32:
                raise ValueError("Packed: only the last source of a concatenation may have a channel count that is not a multiple of 32")
            cs = colsum if len(srcs) == 1 else None
            if batch is not None:
                if not t.is_cuda:
                    raise hip.CraftHipError("craft_amd HIP ops need device tensors (no CPU fallback)")
                batch.add([t.data_ptr(), t.stride(-2), c, rows, B, H, W, ph, pw, self.guard, self.rows_p, prec, self.buf.data_ptr(), off, ncg,
                           cs.data_ptr() if cs is not None else 0, tail], (t, self.buf, cs))
            else:
                call("craft_pack_operand", t, t.stride(-2), c, rows, B, H, W, ph, pw, self.guard, self.rows_p, prec, self.buf, off, ncg, cs, tail)
            off += round_up(c, 32) // 32


class PkMat:
    """A batched packed operand of craft_gemm_pk: ``nb`` batches of ``rows`` rows, each padded to np_ = round_up(rows, 32) rows, over
    ``ncg`` 32-channel groups -- [plane][ncg][nb * np_][32] 16-bit.  Filled by ``fill`` (craft_pack_operand(s)) or by a producer kernel
    (craft_attn_softmax_fwd's Ppk)."""
    __slots__ = ("buf", "nb", "rows", "np_", "ncg", "prec", "rows_total")

    def __init__(self, nb: int, rows: int, C_p: int, prec: int, device):
        self.nb, self.rows, self.np_, self.ncg, self.prec = nb, rows, round_up(rows, 32), C_p // 32, prec
        self.rows_total = round_up(nb * self.np_, 64)
        planes = 2 if prec == hip.PREC_F16X3 else 1
        self.buf = torch.empty(planes * self.ncg * self.rows_total * 32, device=device, dtype=torch.int16)

    def fill(self, x, cg_off: int = 0, batch=None, boff: int = 0):
        """x [.., C] whose leading dims flatten to (batches, rows) with ONE row stride (a channel slice of a contiguous tensor is fine):
        its batches go to batches boff, boff + 1, .. of the pack, channel groups [cg_off, cg_off + ceil(C / 32)); rows beyond ``rows`` and
        channels beyond C of those groups are zeroed (every group of the pack must be filled by exactly one call)."""
        C = x.shape[-1]
        n = x.numel() // C
        nbx = n // self.rows
        if nbx * self.rows != n or nbx + boff > self.nb:
            raise ValueError("PkMat.fill: the source is not a whole number of batches of the pack")
        args = [x.data_ptr(), x.stride(-2), C, n, nbx, 1, self.rows, 0, 0, boff * self.np_, self.rows_total, self.prec, self.buf.data_ptr(), cg_off,
                self.ncg, 0, self.np_ - self.rows]
        if batch is not None:
            if not x.is_cuda:
                raise hip.CraftHipError("craft_amd HIP ops need device tensors (no CPU fallback)")
            batch.add(args, (x, self.buf))
        else:
            call("craft_pack_operand", x, *args[1:12], self.buf, cg_off, self.ncg, None, args[16])
        return self

    def desc(self, kind: int, row_outer: int, row_inner: int, cg0: int = 0, cg_outer: int = 0, cg_inner: int = 0):
        """The 9 longs of craft_gemm_pk for this pack (row strides in BATCHES of the pack)."""
        return [kind, self.rows_total, self.ncg, 0, row_outer * self.np_, row_inner * self.np_, cg0, cg_outer, cg_inner]


PK_ROWS, PK_CH = 0, 1


def gemm_pk(A: PkMat, a_desc, B: PkMat, b_desc, C: torch.Tensor, ldc: int, c_outer: int, c_inner: int, inner: int, nbatch: int, M: int, N: int, K: int,
            alpha: float = 1.0, c_blk_shift: int = 0):
    """C[z][m][n] = alpha sum_k A_z[m, k] B_z[n, k] over packed operands (craft_gemm_pk; K padded to a multiple of 32 with zeros in both packs).
    ``c_blk_shift`` s > 0 (CRAFT_PK_CBLK, c_inner == 2^s): output columns in blocks of 2^s, consecutive blocks inner * 2^s elements apart."""
    import ctypes
    assert A.prec == B.prec
    call("craft_gemm_pk", A.buf, hip.carray(ctypes.c_long, a_desc), B.buf, hip.carray(ctypes.c_long, b_desc), C, ldc, c_outer, c_inner, inner, nbatch, M, N,
         round_up(K, 32), float(alpha), A.prec | (c_blk_shift << 8))


# Operand modes of the backward products of f16x3 layers, set per training pass by train_forward.forward_train from the policy's roles
# (hip.Precision): PREC_F16 -> one fp16 plane, None -> the layer's mode.
#   wgx: the activation operand X of dW = dY^T X        wgy: its gradient operand dY (weight gradients are leaves of the backward graph:
#   dxw: the weight operand of dX = dY W^T              their rounding error does not propagate; dY of the dX chain always keeps both planes)
# The modes belong to a PASS, not to the process: every Function whose backward reads them captures the triple at forward time
# (``ctx.bw_modes = modes()``) and re-installs it on entry (``use_modes``) -- two models or policies may interleave forward A, forward B,
# backward A -- and the installed triple is thread-local (nn.DataParallel runs one backward thread per device).
import threading as _threading


class _Modes(_threading.local):
    wgx = wgy = dxw = None


_M = _Modes()


def set_backward_modes(prec):
    _M.wgx, _M.wgy, _M.dxw = getattr(prec, "wgx", None), getattr(prec, "wgy", None), getattr(prec, "dxw", None)
    if _M.wgy == hip.PREC_F16:
        _M.wgx = hip.PREC_F16               # (a one-plane dY beside a two-plane X is not an instantiation of k_gemm_pk)


def modes():
    """The triple a forward captures for its backward."""
    return (_M.wgx, _M.wgy, _M.dxw)


def use_modes(m):
    """Entry of a backward: the modes of the pass that built this node."""
    if m is not None:
        _M.wgx, _M.wgy, _M.dxw = m


def xprec(prec: int) -> int:
    """Pack mode of a weight gradient's X operand for a layer whose mode is ``prec``."""
    return hip.PREC_F16 if (prec == hip.PREC_F16X3 and _M.wgx == hip.PREC_F16) else prec


def gprec(prec: int) -> int:
    """Pack mode of a weight gradient's dY operand for a layer whose mode is ``prec``."""
    return hip.PREC_F16 if (prec == hip.PREC_F16X3 and _M.wgy == hip.PREC_F16) else prec


def dxflag(prec: int) -> int:
    """CRAFT_CONV_W16 for the input-gradient convolution of a layer in ``prec`` when the policy's role dxw asks for one fp16 plane of the
    weights: two MFMAs per product, dY keeps both planes."""
    return hip.CONV_W16 if (prec == hip.PREC_F16X3 and _M.dxw == hip.PREC_F16) else 0


class PackBatch:
    """Queued craft_pack_operand calls, launched together by ``flush`` (craft_pack_operands)."""

    def __init__(self):
        self.descs, self.keep = [], []

    def add(self, desc, keep):
        self.descs.append(desc)
        self.keep.append(keep)                   # the sources stay alive until the launch is enqueued

    def flush(self):
        import ctypes
        if self.descs:
            flat = [int(v) for d in self.descs for v in d]
            call("craft_pack_operands", hip.carray(ctypes.c_long, flat), len(self.descs))
        self.descs, self.keep = [], []


def wgrad_pk(pairs, KH: int, KW: int, acc: torch.Tensor):
    """acc[cout_p][KH][KW][cin_p] += sum over the (dY pack, X pack) pairs of dY^T X (craft_wgrad_pk: ONE launch over the concatenated K).
    An X operand may be a pair of packs (attributes .a / .b: train_update._CatPack) read as their channel concatenation."""
    import ctypes
    if isinstance(pairs, tuple):
        pairs = [pairs]
    gp, xp = pairs[0]
    two = hasattr(xp, "a")
    for g2, x2 in pairs:
        assert (g2.K, g2.guard, g2.prec, g2.rows_p, g2.C_p, g2.Wp) == (gp.K, gp.guard, gp.prec, gp.rows_p, gp.C_p, gp.Wp)
        assert (x2.K, x2.guard, x2.prec, x2.rows_p, x2.C_p, hasattr(x2, "a")) == (gp.K, gp.guard, xp.prec, xp.rows_p, xp.C_p, two)
        assert not two or x2.a.C_p == xp.a.C_p
    if xp.prec != gp.prec and (gp.prec, xp.prec) != (hip.PREC_F16X3, hip.PREC_F16):
        raise hip.CraftHipError("craft_wgrad_pk: the X packs must be in dY's mode, or one fp16 plane beside f16x3 dY packs")
    flag = hip.WGRAD_X_PREC(xp.prec) if xp.prec != gp.prec else 0
    n = len(pairs)
    ga = hip.carray(ctypes.c_void_p, [g2.buf.data_ptr() for g2, _ in pairs])
    xa = hip.carray(ctypes.c_void_p, [(x2.a if two else x2).buf.data_ptr() for _, x2 in pairs])
    xb = hip.carray(ctypes.c_void_p, [x2.b.buf.data_ptr() for _, x2 in pairs]) if two else None
    call("craft_wgrad_pk", ga, xa, xb, xp.a.C_p if two else xp.C_p, n, gp.rows_p, gp.C_p, xp.rows_p, xp.C_p, gp.guard, gp.K, KH, KW, gp.Wp, acc, gp.prec | flag)


def _wgrad_deferred(cache, wobj, last: bool, pair, KH: int, KW: int, acc: torch.Tensor):
    """Weight gradient of a layer that runs several times per pass (the update block: once per refinement iteration): the packed
    (dY, X) pairs of its calls are queued and the LAST backward call launches one product over all of them -- K = calls x pixels,
    one split-K atomic epilogue and one launch per pass instead of one per call (the epilogue was a third of a per-call launch)."""
    if cache is None:
        wgrad_pk([pair], KH, KW, acc)
        return
    k = (id(wobj), "pk_pending")
    cache.setdefault(k, []).append(pair)
    if last:
        wgrad_pk(cache.pop(k), KH, KW, acc)


def _use_pk(prec: int) -> bool:
    return prec != hip.PREC_F32 and not _NO_PK


import os as _os
_NO_PK = bool(_os.environ.get("CRAFT_NO_PK"))      # developer A/B: the round-2 weight-gradient kernels


def _conv_wgrad(xp, g, B, H8, W8, cin_p, cout_p, KH, KW, prec, db=None) -> torch.Tensor:
    """dW[co][ky][kx][ci] += sum_pix dY[pix][co] X[pix + tap][ci] (craft_conv2d_wgrad; split-K partial sums added with fp32 atomics:
    measured equal to the scratch + reduction form, one kernel less)."""
    dw = hip.zeros((cout_p, KH, KW, cin_p,), xp.device)
    if _use_pk(prec) and cin_p % 32 == 0 and cout_p % 32 == 0:
        geom = (B, H8, W8, KH // 2, KW // 2)
        wgrad_pk([(Packed(g, gprec(prec), geom, colsum=db), Packed(xp, xprec(prec), geom))], KH, KW, dw)
    else:
        call("craft_conv2d_wgrad", xp, xp.stride(-2), cin_p, g, g.stride(-2), cout_p, KH, KW, B, H8, W8, dw, db, None, 0, prec)
    return dw


class Conv(Function):
    """nn.Conv2d (stride 1, padding K//2) + bias (+ReLU) on tokens; w in PyTorch layout [Cout, Cin, KH, KW].
    Backward: input gradient = the same forward kernel with flipped / transposed weights, weight gradient =
    craft_conv2d_wgrad, bias gradient = column sums (both accumulated over the calls of one pass, see _acc_buffer)."""

    @staticmethod
    def forward(ctx, x, w, b, hw, act, prec, cache=None):
        B, N, Cin = x.shape
        Cout, _, KH, KW = w.shape
        cp = pick(prec, "conv")
        cin_p, cout_p = round_up(Cin, 32), round_up(Cout, 32)
        xp = _pad_cols(x, cin_p)
        wt, bp, flag, _ = _conv_weights(w, b, cp, cache, False)
        y = torch.empty(B, N, cout_p, device=x.device, dtype=torch.float32)
        call("craft_conv2d_nhwc", xp, xp.stride(-2), cin_p, wt, bp, cout_p, KH, KW, act, y, cout_p, B, hw[0], hw[1], cp | flag)
        ctx.save_for_backward(xp, y if act != ACT_NONE else None)
        ctx.w, ctx.cache = w, cache
        ctx.bw_modes = modes()
        ctx.dims = (B, N, Cin, Cout, KH, KW, cin_p, cout_p)
        ctx.hw, ctx.act, ctx.prec, ctx.has_bias = hw, act, cp, b is not None
        _count_use(cache, w)
        return y[..., :Cout] if cout_p != Cout else y

    @staticmethod
    def backward(ctx, dy):
        use_modes(ctx.bw_modes)
        xp, y = ctx.saved_tensors
        B, N, Cin, Cout, KH, KW, cin_p, cout_p = ctx.dims
        H8, W8 = ctx.hw
        dev = xp.device
        cache = ctx.cache
        last = _release_use(cache, ctx.w)
        g = _pad_cols(dy, cout_p)
        if ctx.act != ACT_NONE:
            ga = torch.empty(B, N, cout_p, device=dev, dtype=torch.float32)
            call("craft_act_bwd", g, g.stride(-2), y, cout_p, ga, cout_p, B * N, cout_p, ctx.act, 1.0)
            g = ga
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt, zb, flag, _ = _conv_weights(ctx.w, None, ctx.prec, cache, True)
            dxp = torch.empty(B, N, cin_p, device=dev, dtype=torch.float32)
            call("craft_conv2d_nhwc", g, g.stride(-2), cout_p, wt, zb, cin_p, KH, KW, ACT_NONE, dxp, cin_p, B, H8, W8, ctx.prec | flag | dxflag(ctx.prec))
            dx = dxp[..., :Cin] if cin_p != Cin else dxp
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        accb = _acc_buffer(cache, ctx.w, "db", (cout_p,), dev)[0] if want_db else None
        if ctx.needs_input_grad[1]:
            acc, _ = _acc_buffer(cache, ctx.w, "dw", (cout_p, KH, KW, cin_p), dev)
            if _use_pk(ctx.prec):
                # packed operands (the bias gradient rides on the pack of dY)
                geom = (B, H8, W8, KH // 2, KW // 2)
                _wgrad_deferred(cache, ctx.w, last, (Packed(g, gprec(ctx.prec), geom, colsum=accb), Packed(xp, xprec(ctx.prec), geom)), KH, KW, acc)
            else:
                # (the bias gradient rides on the same launch: the blocks of tap 0 add the column sums of dY)
                call("craft_conv2d_wgrad", xp, xp.stride(-2), cin_p, g, g.stride(-2), cout_p, KH, KW, B, H8, W8, acc, accb, None, 0, ctx.prec)
            if last:
                dw = acc[:Cout, :, :, :Cin].permute(0, 3, 1, 2)
        elif want_db:
            call("craft_colsum", g, g.stride(-2), B * N, cout_p, accb)
        if want_db and last:
            db = accb[:Cout]
        return dx, dw, db, None, None, None, None


# ------------------------------------------------------------------------------------------------------------------
# SepConvGRU gates, convex upsampling, loss
# ------------------------------------------------------------------------------------------------------------------
class GruZR(Function):
    """(zr_pre [B,N,2C], h [B,N,C]) -> (z, r*h)  (update.py:55-57 / :60-62)."""

    @staticmethod
    def forward(ctx, zr_pre, h):
        zr_pre, h = _rows(zr_pre), _rows(h)
        B, N, C = h.shape
        z, r, rh = (torch.empty(B, N, C, device=h.device, dtype=torch.float32) for _ in range(3))
        call("craft_gru_zr_fwd", zr_pre, zr_pre.stride(-2), h, h.stride(-2), z, r, rh, B * N, C)
        ctx.save_for_backward(z, r, h)
        return z, rh

    @staticmethod
    def backward(ctx, dz, drh):
        z, r, h = ctx.saved_tensors
        B, N, C = h.shape
        dz = _c(dz) if dz is not None else torch.zeros_like(z)
        drh = _rows(drh) if drh is not None else torch.zeros_like(z)
        dzr = torch.empty(B, N, 2 * C, device=h.device, dtype=torch.float32)
        dh = torch.zeros(B, N, C, device=h.device, dtype=torch.float32)
        call("craft_gru_zr_bwd", dz, drh, drh.stride(-2), z, r, h, h.stride(-2), dzr, dh, B * N, C, None, None, 0)
        return dzr, dh


class GruOut(Function):
    """(q_pre, z, h) -> h' = (1 - z) h + z tanh(q_pre)  (update.py:57-58 / :62-63)."""

    @staticmethod
    def forward(ctx, q_pre, z, h):
        q_pre, z, h = _rows(q_pre), _c(z), _rows(h)
        B, N, C = h.shape
        q = torch.empty(B, N, C, device=h.device, dtype=torch.float32)
        hn = torch.empty(B, N, C, device=h.device, dtype=torch.float32)
        call("craft_gru_out_fwd", q_pre, q_pre.stride(-2), z, h, h.stride(-2), q, hn, C, B * N, C)
        ctx.save_for_backward(z, q, h)
        return hn

    @staticmethod
    def backward(ctx, dhn):
        z, q, h = ctx.saved_tensors
        dhn = _rows(dhn)
        B, N, C = h.shape
        dqp, dz, dh = (torch.empty(B, N, C, device=h.device, dtype=torch.float32) for _ in range(3))
        call("craft_gru_out_bwd", dhn, dhn.stride(-2), z, q, h, h.stride(-2), dqp, dz, dh, B * N, C, None)
        return dqp, dz, dh


class ZeroGradEdge(Function):
    """y = x (same storage), plus an autograd edge to ``params`` that carries an exactly-zero gradient (train_forward.
    _touch_cancelling_biases)."""

    @staticmethod
    def forward(ctx, x, *params):
        ctx.shapes = [(p.shape, p.dtype) for p in params]
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        return (dy,) + tuple(hip.zeros(tuple(sh), dy.device, dt) for sh, dt in ctx.shapes)


class ConvexUpsample(Function):
    """CRAFT.upsample_flow (network.py:151-162): mask tokens [B,N,576], flow tokens [B,N,2] -> [B,2,8*H8,8*W8]."""

    @staticmethod
    def forward(ctx, mask, flow, hw):
        mask, flow = _rows(mask), _c(flow)
        ctx.save_for_backward(mask, flow)
        ctx.hw = hw
        if mask.stride(-2) != 576:
            mask = mask.contiguous()
        return ops.convex_upsample(mask, flow, hw[0], hw[1])

    @staticmethod
    def backward(ctx, dup):
        mask, flow = ctx.saved_tensors
        B, N, _ = flow.shape
        dup = _c(dup)
        dmask = torch.empty(B, N, 576, device=flow.device, dtype=torch.float32)
        dflow = hip.zeros((B, N, 2,), flow.device)
        call("craft_convex_upsample_bwd", mask, mask.stride(-2), flow, dup, B, ctx.hw[0], ctx.hw[1], dmask, 576, dflow, 2)
        return dmask, dflow, None


class SequenceLoss(Function):
    """train.py:44-61: sum_i gamma^(T-1-i) mean(valid |pred_i - gt|) with its gradient from craft_flow_l1_loss."""

    @staticmethod
    def forward(ctx, gt, valid, gamma, max_flow, *preds):
        T = len(preds)
        B, _, H, W = gt.shape
        dev = preds[0].device
        gt, va = _c(gt.to(dev).float()), _c(valid.to(dev).float())
        loss = torch.zeros((), device=dev, dtype=torch.float64)
        grads = []
        for i, p in enumerate(preds):
            g = torch.empty(B, 2, H, W, device=dev, dtype=torch.float32)
            call("craft_flow_l1_loss", _c(p.float()), gt, va, B, H, W, float(gamma ** (T - i - 1)), float(max_flow), loss, g)
            grads.append(g)
        ctx.grads = grads
        return loss.float()

    @staticmethod
    def backward(ctx, dl):
        grads, ctx.grads = ctx.grads, None
        if grads is None:          # (the gradient buffers are scaled in place and handed on: this node runs once)
            raise RuntimeError("SequenceLoss: this graph was already consumed by a backward pass (its gradient buffers are scaled in place "
                               "and released; call sequence_loss again instead of backward(retain_graph=True) twice)")
        torch._foreach_mul_(grads, dl.reshape(()).to(grads[0].dtype))        # (one multi-tensor launch for the T predictions, in place: the buffers are this node's own)
        return (None, None, None, None) + tuple(grads)


class DeferredMetrics:
    """The metrics of train.py:63-71 accumulated on the device; ``resolve()`` reads them back (a host synchronisation)."""

    def __init__(self, m):
        self.m = m

    def resolve(self):
        r = self.m.result()
        return {"epe": r["epe"], "1px": r["px1"], "3px": r["px3"], "5px": r["px5"]}


def sequence_loss(flow_preds, flow_gt, valid, gamma: float = 0.8, max_flow: float = 400.0, defer_metrics: bool = False):
    """Differentiable loss (0-dim tensor) and the metrics dict of train.py:63-71 (computed on the device).  ``defer_metrics``: return a
    DeferredMetrics instead of the dict -- reading the metrics back right here drains the device queue between the forward and the
    backward pass (train.Trainer.step resolves them at the end of the step, together with the loss)."""
    from .evaluate import FlowMetrics
    loss = SequenceLoss.apply(flow_gt, valid, gamma, max_flow, *flow_preds)
    m = FlowMetrics(flow_preds[-1].device, max_mag=max_flow)
    m.update(flow_preds[-1].detach(), flow_gt, valid)
    if defer_metrics:
        return loss, DeferredMetrics(m)
    return loss, DeferredMetrics(m).resolve()
