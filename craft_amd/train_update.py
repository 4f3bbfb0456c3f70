"""One refinement iteration of ``CRAFT.forward`` in training mode as ONE autograd node with a hand-written backward
(network.py:230-260, update.py:137-162: motion encoder -> motion aggregator -> SepConvGRU -> flow head / mask head -> convex
upsampling).

The first training path (round 2, craft_amd/train_forward.py with ``args.hip_fused_update = False``) composed the iteration from ~40
small autograd.Functions: correct, but every ``torch.cat`` / zero fill / gradient add between them was a PyTorch kernel (~1 000 per
step, 8 % of the kernel time) and the forward could not use the fused inference kernels.  Here the iteration works on ONE 640-wide
token buffer per iteration

    HX_t = [ h1 (128) | h0 (128) | mf (128) | mfg (128) ]          h0 = net_t, h1 = hidden state after the horizontal GRU pass,
                                                                  the vertical pass writes net_{t+1} into HX_{t+1}'s h0 slot

The context features ``inp`` are the same in every iteration, so their share of the six gate convolutions is hoisted out of the loop in
BOTH directions (as the inference path hoists the forward): forward = per-pixel bias fields computed once (``craft_sepconv_gru_context``),
the per-iteration convolutions run over [h | mf | mfg] only (K 2560 -> 1920); backward = the gate gradients dY are summed over the
iterations and ONE input-gradient convolution and ONE weight-gradient product per gate convolution handle the inp channels at the end.

(every ``torch.cat`` of update.py is a column range, the q gate's cat([r*h, x]) a two-segment convolution input), the forward runs
the inference entry points where they keep what the backward needs (``craft_motion_encoder``, ``craft_flow_head``, ``craft_mask_head``:
their workspaces ARE the saved activations), the convolution inputs are packed for the weight gradients while they are at hand
(``autograd.Packed``: the fp32 copies need not survive), and the backward is one straight sequence of kernel calls: input gradients =
the forward convolution with flipped weights, weight gradients queued per layer and launched once per pass over all 12 iterations
(``craft_wgrad_pk``), gate / activation / pooling gradients by their kernels.  Results equal the unfused path's up to summation order
(tests/test_train_update.py).
"""
from __future__ import annotations

import os

import torch
from torch.autograd import Function

from . import autograd as AG
from . import hip, ops
from .hip import ACT_NONE, ACT_RELU, PREC_F32, STATS_REPLICAS, W1X1_PACKED, W_PACKED, call, pick, round_up

_C = 512          # columns of HX
H1, H0, MF, MFG = 0, 128, 256, 384


class UpdatePass:
    """Everything the 12 iterations of one forward pass share: buffers, packed weights, gradient accumulators."""

    def __init__(self, model, prec, hw, B, iters, net, inp, holder, pholder, radius):
        self.model, self.prec, self.hw, self.B, self.iters = model, prec, hw, B, iters
        self.N = hw[0] * hw[1]
        self.holders = list(holder) if isinstance(holder, (list, tuple)) else [holder]     # one volume, or the two of --f1
        self.pholder, self.radius = pholder, radius
        self.vcat = None           # packed [V_1 .. V_T] of the aggregator (craft_gemm_pk; built iteration by iteration in the forward)
        dev = net.device
        self.dev = dev
        self.cp = pick(prec, "conv")
        if self.cp == hip.PREC_F32:
            raise NotImplementedError("the fused update iteration runs the 16-bit / f16x3 operand modes; fp32 takes the unfused path")
        ub = model.update_block
        self.setrans = bool(model.args.use_setrans)
        # one buffer for all iterations (+ 1: the last iteration's net_{T} slot); inp is written once for all of them
        self.HX = torch.empty(iters + 1, B, self.N, _C, device=dev, dtype=torch.float32)
        self.HX[0, :, :, H0:H0 + 128] = net.detach()
        self.inp = AG._rows(inp.detach())
        self.cache = {}            # conv operand cache of the pass (transposed weights), gradient accumulators, queued packs
        gru = ub.gru
        # forward weights: the inference modules' packed copies (re-made when the optimizer bumps the weights epoch); one stream here,
        # so no device synchronisation behind the re-packing
        from . import update as _U
        _sync, _U.PACK_SYNC[0] = _U.PACK_SYNC[0], False
        try:
            self.w_enc = ub.encoder.packed(self.cp)
            # gate convolutions over [h | mf | mfg] (the inp channels 128..255 of the 512-channel input are hoisted): (zr1, q1, zr2, q2)
            self.w_gru, _ = gru.packed_split(self.cp, 128, 256)
            self.fields = gru.context_tokens(self.inp, hw, prec)        # [B, N, 768]: inp's share + bias of zr1 | q1 | zr2 | q2
            self.w_fh = ub.flow_head.packed(self.cp)
            self.w_mask = ub.packed_mask(self.cp)
        finally:                   # (an OOM / unsupported shape in here must not leave the inference path's re-packs unsynchronised)
            _U.PACK_SYNC[0] = _sync
        # input-gradient operands of the gate convolutions (flipped / transposed, one launch each): the varying channels [h | mf | mfg]
        # per iteration, the hoisted inp channels once per pass
        VAR, INP_ = ((0, 128), (256, 512)), ((128, 256), (0, 0))
        zr = [(gru.convz1.weight, gru.convr1.weight), (gru.convz2.weight, gru.convr2.weight)]
        q = [gru.convq1.weight, gru.convq2.weight]
        self.wzrT = [ops.pack_conv_weights(a, self.cp, b, sel=VAR, transposed=True) for a, b in zr]
        self.wqT = [ops.pack_conv_weights(a, self.cp, None, sel=VAR, transposed=True) for a in q]
        self.wzrT_inp = [ops.pack_conv_weights(a, self.cp, b, sel=INP_, transposed=True) for a, b in zr]
        self.wqT_inp = [ops.pack_conv_weights(a, self.cp, None, sel=INP_, transposed=True) for a in q]
        self.zero_bias = torch.zeros(1024, device=dev, dtype=torch.float32)
        self.dysum = {}            # (gate, pass) -> sum over the iterations of that convolution's output gradient (for the inp channels)
        self.rep_agg = None        # replicated (dw_agg, dskip) table of the aggregator's pooling, reduced at the end of the pass
        self.dgamma = None
        self.saved = [None] * iters
        self.ready = []
        self.dv = [None] * iters
        self.zero1 = torch.zeros(1, device=dev, dtype=torch.float32)
        self.zero_tok = None

    # ---- gradient accumulators ------------------------------------------------------------------------------------
    def acc(self, key, shape):
        k = ("acc", key)
        b = self.cache.get(k)
        if b is None:
            b = self.cache[k] = hip.zeros(shape, self.dev)
        return b

    def wgrad(self, key, pair, KH, KW, acc, last):
        """Queue the packed (dY, X) pair of this iteration; after the last one (iteration 0 in phase 1) ``launch_ready`` runs ONE product
        over all of them -- once the pack batch that holds this iteration's dY packs has been flushed."""
        k = ("pk_pending", key)
        self.cache.setdefault(k, []).append(pair)
        if last:
            self.ready.append((k, KH, KW, acc))

    def launch_ready(self):
        for k, KH, KW, acc in self.ready:
            AG.wgrad_pk(self.cache.pop(k), KH, KW, acc)
        self.ready = []


_NO_MASK_EPI = bool(os.environ.get("CRAFT_NO_MASK_EPI"))
_NO_FIELD_COL0 = bool(os.environ.get("CRAFT_NO_FIELD_COL0"))          # developer A/B: the per-iteration torch.add of the two passes' d[mf | mfg]


def _conv_dx(ps: UpdatePass, w, g, cout_p, KH, KW, out=None, cin_p=None, field=None, relu_y=None, field_col0=0):
    """Input gradient of a stride-1 'same' convolution: the forward kernel with flipped / transposed weights.  g [B, N, cout_p]
    (row stride may exceed cout_p) -> [B, N, cin_p].  w: the nn.Conv2d weight, or (with cin_p) an operand already packed by
    ops.pack_conv_weights(transposed=True).  field [B, N, >= cin_p]: out = conv + field (a gradient that is already there: the
    convolution's per-pixel bias field, `out` may be `field` itself) instead of a separate add pass.  relu_y [B, N, >= cin_p]: the saved
    output of the ReLU layer BELOW this convolution -- its backward (out = relu_y > 0 ? out : 0) runs in the epilogue
    (craft_conv2d_nhwc2_mask) instead of as a craft_act_bwd pass over the result.  field_col0 (a multiple of 32, plain field form only):
    the field is added to columns >= field_col0 (CRAFT_CONV_FIELD_COL0)."""
    if cin_p is None:
        wt, zb, flag, _ = AG._conv_weights(w, None, ps.cp, ps.cache, True)
        cin_p = round_up(w.shape[1], 32)
    else:
        wt, zb, flag = w, ps.zero_bias, W_PACKED
    if out is None:
        out = torch.empty(ps.B, ps.N, cin_p, device=ps.dev, dtype=torch.float32)
    if relu_y is not None and _NO_MASK_EPI:          # developer A/B: the separate craft_act_bwd pass
        out = _conv_dx(ps, w if cin_p is None else wt, g, cout_p, KH, KW, out=out, cin_p=cin_p, field=field)
        return _act_bwd(out, relu_y, out.shape[-1], out=out)
    if relu_y is not None:
        call("craft_conv2d_nhwc2_mask", g, g.stride(-2), cout_p, None, 0, 0, wt, zb if field is None else None, field,
             field.stride(-2) if field is not None else 0, cin_p, KH, KW, relu_y, relu_y.stride(-2), out, out.stride(-2), ps.B, ps.hw[0], ps.hw[1],
             ps.cp | flag | AG.dxflag(ps.cp))
    elif field is not None:
        call("craft_conv2d_nhwc2", g, g.stride(-2), cout_p, None, 0, 0, wt, None, field, field.stride(-2), cin_p, KH, KW, ACT_NONE, out, out.stride(-2),
             ps.B, ps.hw[0], ps.hw[1], ps.cp | flag | AG.dxflag(ps.cp) | ((field_col0 // 32) << 16))
    else:
        call("craft_conv2d_nhwc", g, g.stride(-2), cout_p, wt, zb, cin_p, KH, KW, ACT_NONE, out, out.stride(-2), ps.B, ps.hw[0], ps.hw[1],
             ps.cp | flag | AG.dxflag(ps.cp))
    return out


def _act_bwd(dy, y, C, out=None, act=ACT_RELU, scale=1.0):
    rows = dy.shape[0] * dy.shape[1]
    if out is None:
        out = torch.empty(dy.shape[0], dy.shape[1], C, device=dy.device, dtype=torch.float32)
    call("craft_act_bwd", dy, dy.stride(-2), y, y.stride(-2), out, out.stride(-2), rows, C, act, float(scale))
    return out


class UpdateIter(Function):
    """(net_t, correlation token, P token, inp, *parameters) -> (net_{t+1}, flow prediction t, coords1_{t+1}).

    Backward in two phases.  The motion features of iteration t depend on the (detached, network.py:232) coordinates only -- not on
    net_t -- so their gradient never re-enters the recurrence: phase 1 (this node's backward, iterations in reverse) runs the heads
    and the SepConvGRU and keeps the gradient of [mf | mfg]; phase 2 (once, inside the backward of iteration 0) runs aggregator,
    motion encoder and correlation-lookup gradients of ALL iterations, with the 12 products dV_t = P^T dO_t as ONE product over the
    concatenated dO (the 1 GB of attention probabilities is streamed once instead of twelve times; the same concatenation serves the
    deferred dP = [dO_1 .. dO_T] [V_1 .. V_T]^T of autograd.ProbsToken).  The correlation pyramid's and P's gradients leave through the
    two token inputs (autograd.CorrVolume / ProbsToken run after every node that took the token)."""

    @staticmethod
    def forward(ctx, net, token, ptoken, inp, ps: UpdatePass, t: int, coords1, coords0, *params):
        ctx.set_materialize_grads(False)          # (coords1 is non-differentiable, the last net unused: backward tests for None)
        m = ps.model
        ub = m.update_block
        B, N, (H8, W8) = ps.B, ps.N, ps.hw
        rows, cp, prec, dev = B * N, ps.cp, ps.prec, ps.dev
        hx, hxn = ps.HX[t], ps.HX[t + 1]
        coords1 = AG._c(coords1.detach())
        corr = ops.corr_lookup([h.pyr for h in ps.holders], coords1, ps.radius)       # network.py:235 / corr.py:47-71
        S = {"coords": coords1}
        pb = AG.PackBatch()                                                          # the iteration's conv inputs: ONE pack launch at its end
        if ps.zero_tok is None:
            ps.zero_tok = torch.zeros_like(token)
        flow = torch.empty(B, N, 2, device=dev, dtype=torch.float32)                 # coords1 - coords0, its padded copy and coords1's
        flow32 = torch.empty(B, N, 32, device=dev, dtype=torch.float32)              # copy for the flow head: one launch
        c1n = torch.empty(B, N, 2, device=dev, dtype=torch.float32)
        call("craft_flow_tokens", coords1, AG._c(coords0), rows, flow, flow32, c1n)
        # ---- BasicMotionEncoder (update.py:79-87): one fused call; its workspace keeps cor1 | [cor2 | flo2] | flo1
        me = torch.empty(rows * 640, device=dev, dtype=torch.float32)
        call("craft_motion_encoder", corr, corr.stride(-2), ub.encoder.cor_planes, flow, *ps.w_enc, B, H8, W8, hx[..., MF:MF + 128], _C, me,
             cp | W_PACKED | (W1X1_PACKED if (cp != PREC_F32 and not os.environ.get("CRAFT_NO_LINEAR_PACK")) else 0), None, None)
        S["cor1"] = me[: rows * 256].view(B, N, 256)
        S["cf"] = me[rows * 256: rows * 512].view(B, N, 256)
        S["flo1"] = me[rows * 512:].view(B, N, 128)
        mf = hx[..., MF:MF + 128]
        # packs of the motion encoder's conv inputs (for the weight gradients)
        g3, g7 = (B, H8, W8, 1, 1), (B, H8, W8, 3, 3)
        xcp = AG.xprec(cp)                                                       # mode of the weight gradients' X operands (policy role wgx)
        S["pk_corr"] = AG.Packed(corr, xcp, batch=pb)
        S["pk_cor1"] = AG.Packed(S["cor1"], xcp, g3, batch=pb)
        S["pk_flow"] = AG.Packed(flow32, xcp, g7, batch=pb)
        S["pk_flo1"] = AG.Packed(S["flo1"], xcp, g3, batch=pb)
        S["pk_cf"] = AG.Packed(S["cf"], xcp, g3, batch=pb)
        # ---- motion aggregator (update.py:143-149)
        P = ps.pholder.P
        Bp, M, _, ld = P.shape
        agg = ub.aggregator
        pv = pick(prec, "pv")
        if ps.setrans:
            va = ops.linear(mf, agg.first_linear.weight.detach(), None, prec)             # [B, N, M*128]
        else:
            va = ops.linear(mf, agg.to_v.weight.detach().view(agg.heads * agg.dim_head, -1), None, prec)
        Cv = va.shape[-1] // M
        Oa = torch.empty(B, M, N, Cv, device=dev, dtype=torch.float32)
        ppk = ps.pholder.pk
        if ppk is not None:
            # packed operands (craft_gemm_pk): V_t goes straight into the pack the deferred dP = [dO_1..dO_T] [V_1..V_T]^T reads in the
            # backward -- rows (b, j), channels (m, t, c) -- and O_t = P V_t reads its (t) channel groups from there
            T, cgv = ps.iters, Cv // 32
            if ps.vcat is None:
                ps.vcat = AG.PkMat(B, N, M * T * Cv, ppk.prec, dev)
            vb = AG.PackBatch()
            for m_ in range(M):
                ps.vcat.fill(va[..., m_ * Cv:(m_ + 1) * Cv], cg_off=(m_ * T + t) * cgv, batch=vb)
            vb.flush()
            AG.gemm_pk(ppk, ppk.desc(AG.PK_CH, M, 1), ps.vcat, ps.vcat.desc(AG.PK_ROWS, 1, 0, t * cgv, 0, T * cgv), Oa, Cv, M * N * Cv, N * Cv, M, B * M,
                       N, Cv, ld)
        else:
            AG.gemm(P, ld, 1, M * N * ld, N * ld, va, 1, va.stride(-2), N * va.stride(-2), Cv, Oa, Cv, M * N * Cv, N * Cv, M, B * M, N, Cv, N, prec=pv)
        if ps.setrans:
            ops.mode_pool_ln(Oa, mf, agg.feat_softaggr.feat2score.weight.detach(), agg.input_skip_coeff.detach(), out=hx[..., MFG:MFG + 128])
        else:
            ops.gma_residual(mf, Oa.view(B, N, Cv), agg.gamma.detach(), out=hx[..., MFG:MFG + 128])
        S["va"], S["Oa"] = (va if ppk is None else None), Oa
        S["pk_mf"] = AG.Packed(mf, AG.xprec(pick(prec, "proj")), batch=pb)
        # ---- SepConvGRU (update.py:49-64): horizontal pass h0 -> h1 (in HX_t), vertical pass h1 -> net_{t+1} (into HX_{t+1});
        # convolutions over [h | v], v = [mf | mfg]; inp's share and the biases arrive as per-pixel fields
        wzr1, wq1, wzr2, wq2 = ps.w_gru
        v = hx[..., MF:MF + 256]
        F_ = ps.fields
        for p_, (KH, KW, wzr, wq, fz, fq) in enumerate(((1, 5, wzr1, wq1, 0, 256), (5, 1, wzr2, wq2, 384, 640))):
            h = hx[..., H0:H0 + 128] if p_ == 0 else hx[..., H1:H1 + 128]
            hn = hx[..., H1:H1 + 128] if p_ == 0 else hxn[..., H0:H0 + 128]
            geom = (B, H8, W8, KH // 2, KW // 2)
            zr_pre = torch.empty(B, N, 256, device=dev, dtype=torch.float32)
            call("craft_conv2d_nhwc2", h, _C, 128, v, _C, 256, wzr, None, F_[..., fz:], 768, 256, KH, KW, ACT_NONE, zr_pre, 256, B, H8, W8, cp | W_PACKED)
            z, r, rh = (torch.empty(B, N, 128, device=dev, dtype=torch.float32) for _ in range(3))
            call("craft_gru_zr_fwd", zr_pre, 256, h, _C, z, r, rh, rows, 128)
            q_pre = zr_pre[..., :128]                                                # (zr_pre is dead: reuse its first half)
            call("craft_conv2d_nhwc2", rh, 128, 128, v, _C, 256, wq, None, F_[..., fq:], 768, 128, KH, KW, ACT_NONE, q_pre, 256, B, H8, W8, cp | W_PACKED)
            q = torch.empty(B, N, 128, device=dev, dtype=torch.float32)
            call("craft_gru_out_fwd", q_pre, 256, z, h, _C, q, hn, _C, rows, 128)
            S[f"pk_h{p_}"] = AG.Packed(h, xcp, geom, batch=pb)
            S[f"pk_rh{p_}"] = AG.Packed(rh, xcp, geom, batch=pb)
            S[f"pk_v{p_}"] = AG.Packed(v, xcp, geom, batch=pb)                                 # shared by the z|r and the q convolution of this pass
            S[f"z{p_}"], S[f"r{p_}"], S[f"q{p_}"] = z, r, q
        h2 = hxn[..., H0:H0 + 128]
        S["pk_h2"] = AG.Packed(h2, xcp, g3, batch=pb)
        # ---- heads (update.py:15-16, :124-127, :161) + coords1 += delta (network.py:247) + convex upsampling (:258)
        fh1 = torch.empty(B, N, 256, device=dev, dtype=torch.float32)
        flow_new = torch.empty(B, N, 2, device=dev, dtype=torch.float32)
        call("craft_flow_head", h2, _C, *ps.w_fh, B, H8, W8, c1n, coords0, flow_new, None, fh1, cp | W_PACKED)
        mh = torch.empty(B, N, 256, device=dev, dtype=torch.float32)
        mask = torch.empty(B, N, 576, device=dev, dtype=torch.float32)
        call("craft_mask_head", h2, _C, *ps.w_mask, B, H8, W8, mask, mh, cp | W_PACKED)
        up = ops.convex_upsample(mask, flow_new, H8, W8)
        S["fh1"], S["mh"], S["mask"], S["flow_new"] = fh1, mh, mask, flow_new
        S["pk_fh1"] = AG.Packed(fh1, xcp, g3, batch=pb)
        S["pk_mh"] = AG.Packed(mh, xcp, batch=pb)
        pb.flush()
        ps.saved[t] = S
        ctx.ps, ctx.t = ps, t
        ctx.bw_modes = AG.modes()
        ctx.nparams = len(params)
        ctx.mark_non_differentiable(c1n)
        return hxn[..., H0:H0 + 128], up, c1n

    @staticmethod
    def backward(ctx, d_hn, d_up, _dc):
        AG.use_modes(ctx.bw_modes)
        ps, t = ctx.ps, ctx.t
        S = ps.saved[t]
        m = ps.model
        ub = m.update_block
        enc, gru, fh, agg = ub.encoder, ub.gru, ub.flow_head, ub.aggregator
        B, N, (H8, W8) = ps.B, ps.N, ps.hw
        rows, cp, prec, dev = B * N, ps.cp, ps.prec, ps.dev
        hx, hxn = ps.HX[t], ps.HX[t + 1]
        last = t == 0                                # backward runs the iterations in reverse: t = 0 completes every accumulator
        g3, g7 = (B, H8, W8, 1, 1), (B, H8, W8, 3, 3)
        E = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)      # noqa: E731
        pb = AG.PackBatch()                                                  # this call's dY packs: one launch at its end
        h2 = hxn[..., H0:H0 + 128]

        # ---- convex upsampling, mask head
        dh2 = None
        if d_up is not None:
            dmask = E(B, N, 576)
            dflow = hip.zeros((B, N, 32), dev)           # (2 live columns; the flow head's padded output: accumulated in place, stride 32)
            call("craft_convex_upsample_bwd", S["mask"], 576, S["flow_new"], AG._c(d_up), B, H8, W8, dmask, 576, dflow, 32)
            # mask = 0.25 * (W2 mh + b2): the gradient w.r.t. the pre-scale output
            dm = _act_bwd(dmask, dmask, 576, out=dmask, act=ACT_NONE, scale=0.25)
            w2 = ub.mask[2].weight
            w2m = w2.detach().view(576, 256)
            d_mh = E(B, N, 256)
            AG.gemm(dm, 576, 1, 0, 0, w2m, 1, 256, 0, 0, d_mh, 256, 0, 0, 1, 1, rows, 256, 576, prec=cp)
            ps.wgrad(("mask2",), (AG.Packed(dm, AG.gprec(cp), colsum=ps.acc(("mask2", "db"), (576,)), batch=pb), S["pk_mh"]), 1, 1, ps.acc(("mask2", "dw"), (576, 256)), last)
            g_mh = _act_bwd(d_mh, S["mh"], 256, out=d_mh)
            dh2 = _conv_dx(ps, ub.mask[0].weight, g_mh, 256, 3, 3, field=AG._rows(d_hn) if d_hn is not None else None)   # (+ the recurrence's share)
            ps.wgrad(("mask0",), (AG.Packed(g_mh, AG.gprec(cp), g3, colsum=ps.acc(("mask0", "db"), (256,)), batch=pb), S["pk_h2"]), 3, 3, ps.acc(("mask0", "dw"), (256, 3, 3, 128)), last)
            # ---- flow head: delta = conv2(relu(conv1(h2)))
            g_fh1 = _conv_dx(ps, fh.conv2.weight, dflow, 32, 3, 3, relu_y=S["fh1"])          # (ReLU backward of conv1's output in the epilogue)
            ps.wgrad(("fh2",), (AG.Packed(dflow, AG.gprec(cp), g3, colsum=ps.acc(("fh2", "db"), (32,)), batch=pb), S["pk_fh1"]), 3, 3, ps.acc(("fh2", "dw"), (32, 3, 3, 256)), last)
            _conv_dx(ps, fh.conv1.weight, g_fh1, 256, 3, 3, out=dh2, field=dh2)
            ps.wgrad(("fh1",), (AG.Packed(g_fh1, AG.gprec(cp), g3, colsum=ps.acc(("fh1", "db"), (256,)), batch=pb), S["pk_h2"]), 3, 3, ps.acc(("fh1", "dw"), (256, 3, 3, 128)), last)
        elif d_hn is not None:
            dh2 = AG._c(d_hn).clone()
        for k in ("pk_h2", "pk_fh1", "pk_mh", "fh1", "mh", "mask", "flow_new"):
            S.pop(k)
        if dh2 is None:
            raise RuntimeError("UpdateIter.backward without any output gradient")

        # ---- SepConvGRU, vertical pass then horizontal pass
        dv = tq_prev = None
        dh = dh2
        for p_, (KH, KW) in ((1, (5, 1)), (0, (1, 5))):
            h = hx[..., H0:H0 + 128] if p_ == 0 else hx[..., H1:H1 + 128]
            z, r, q = S[f"z{p_}"], S[f"r{p_}"], S[f"q{p_}"]
            geom = (B, H8, W8, KH // 2, KW // 2)
            dqp, dz, dhp = E(B, N, 128), E(B, N, 128), E(B, N, 128)
            # the inp channels see the SUM of the gate gradients over the iterations (one input-gradient convolution and one weight-gradient
            # product per gate convolution at the end of the pass): the gate kernels keep the running sums
            if ("q", p_) not in ps.dysum:
                ps.dysum[("q", p_)], ps.dysum[("zr", p_)] = hip.zeros((B, N, 128), dev), hip.zeros((B, N, 256), dev)
            call("craft_gru_out_bwd", dh, dh.stride(-2), z, q, h, _C, dqp, dz, dhp, rows, 128, ps.dysum[("q", p_)])
            # d[rh | mf | mfg]; the second pass adds the first pass's d[mf | mfg] here (columns 128.. of its 384-wide result) -- it was a
            # torch.add over [B, N, 256] per iteration
            if tq_prev is not None and not _NO_FIELD_COL0:
                tq = _conv_dx(ps, ps.wqT[p_], dqp, 128, KH, KW, cin_p=384, field=tq_prev, field_col0=128)
                dv = None
            else:
                tq = _conv_dx(ps, ps.wqT[p_], dqp, 128, KH, KW, cin_p=384)
            ps.wgrad(("q", p_), (AG.Packed(dqp, AG.gprec(cp), geom, colsum=ps.acc(("q", p_, "db"), (128,)), batch=pb), _cat_pack(S[f"pk_rh{p_}"], S[f"pk_v{p_}"])), KH, KW,
                     ps.acc(("q", p_, "dw"), (128, KH, KW, 384)), last)
            dzr = E(B, N, 256)
            # tq[:, :128] <- dhp + d(rh) r: tq is now [dh so far | d(mf, mfg) so far], the field the z|r convolution adds its own to
            call("craft_gru_zr_bwd", dz, tq, 384, z, r, h, _C, dzr, dhp, rows, 128, ps.dysum[("zr", p_)], tq, 384)
            _conv_dx(ps, ps.wzrT[p_], dzr, 256, KH, KW, cin_p=384, out=tq, field=tq)          # += d[h | mf | mfg]
            ps.wgrad(("zr", p_), (AG.Packed(dzr, AG.gprec(cp), geom, colsum=ps.acc(("zr", p_, "db"), (256,)), batch=pb), _cat_pack(S[f"pk_h{p_}"], S[f"pk_v{p_}"])), KH, KW,
                     ps.acc(("zr", p_, "dw"), (256, KH, KW, 384)), last)
            dv = tq[..., 128:] if dv is None else torch.add(dv, tq[..., 128:])
            dh = tq[..., :128]
            tq_prev = tq
            for k in (f"pk_h{p_}", f"pk_rh{p_}", f"pk_v{p_}", f"z{p_}", f"r{p_}", f"q{p_}"):
                S.pop(k)
        d_net = dh                                                                            # gradient of net_t
        ps.dv[t] = dv                                                                         # gradient of [mf | mfg]: phase 2
        ps.saved[t] = S
        pb.flush()
        ps.launch_ready()
        d_inp = None
        if last:
            # ---- the hoisted inp channels: d_inp = sum over the four gate convolutions of conv^T(W_inp, sum_t dY_t), dW_inp = (sum_t dY_t)^T inp
            for p_, (KH, KW) in ((0, (1, 5)), (1, (5, 1))):
                geom = (B, H8, W8, KH // 2, KW // 2)
                pk_inp = AG.Packed(ps.inp, AG.xprec(cp), geom)
                for kind, w_inp, co in (("zr", ps.wzrT_inp[p_], 256), ("q", ps.wqT_inp[p_], 128)):
                    g = ps.dysum.pop((kind, p_))
                    d_inp = _conv_dx(ps, w_inp, g, co, KH, KW, cin_p=128, out=d_inp, field=d_inp)     # (+= through the convolution's bias field)
                    AG.wgrad_pk([(AG.Packed(g, AG.gprec(cp), geom), pk_inp)], KH, KW, ps.acc((kind, p_, "dw_inp"), (co, KH, KW, 128)))
        if last:
            _phase2(ps)
        grads = _param_grads(ps) if last else (None,) * ctx.nparams
        # the two token inputs only order CorrVolume / ProbsToken behind every iteration (the engine counts graph edges, defined or not):
        # one zero gradient, from the iteration that runs last, is enough -- twelve of them were 22 additions of zeros by the engine
        return (d_net, ps.zero_tok if last else None, ps.zero1 if last else None, d_inp, None, None, None, None) + tuple(grads)


def _phase2(ps: UpdatePass):
    """Aggregator, motion-encoder and correlation-lookup gradients of all iterations (see UpdateIter)."""
    m = ps.model
    ub = m.update_block
    enc, agg = ub.encoder, ub.aggregator
    B, N, (H8, W8), T = ps.B, ps.N, ps.hw, ps.iters
    rows, cp, prec, dev = B * N, ps.cp, ps.prec, ps.dev
    g3, g7 = (B, H8, W8, 1, 1), (B, H8, W8, 3, 3)
    E = lambda *s_: torch.empty(*s_, device=dev, dtype=torch.float32)      # noqa: E731
    P = ps.pholder.P
    _, M, _, ld = P.shape
    pv, pp = pick(prec, "pv"), pick(prec, "proj")
    Cv = ps.saved[0]["Oa"].shape[-1]
    # ---- A: gradient of the aggregator's pooling per iteration -> dO_t, the direct part of d mf_t
    dOs, d_mfs = [], []
    ppk = ps.pholder.pk
    docat = AG.PkMat(B * M, N, T * Cv, ppk.prec, dev) if ppk is not None else None       # rows (b, m, i), channels (t, c)
    dob = AG.PackBatch()
    missing = [t for t in range(T) if ps.dv[t] is None or ps.saved[t] is None]
    if missing:
        raise RuntimeError(f"the fused update block's backward needs every refinement iteration in the graph; iteration(s) {missing} of {T} "
                           "received no gradient (a loss over a prefix / subset of the predictions?) -- use every prediction, or "
                           "run fewer iterations")
    for t in range(T):
        S, dv = ps.saved[t], ps.dv[t]
        mf = ps.HX[t][..., MF:MF + 128]
        d_mfg = dv[..., 128:256]
        d_mf = E(B, N, 128)
        if ps.setrans:
            if ps.rep_agg is None:
                ps.rep_agg = hip.zeros((STATS_REPLICAS, Cv + 1), dev)
            dOa = torch.empty_like(S["Oa"])
            w_agg, skip = agg.feat_softaggr.feat2score.weight, agg.input_skip_coeff
            call("craft_mode_pool_ln_bwd", S["Oa"], mf, _C, AG._c(w_agg.detach()).view(-1), skip.detach(), d_mfg, d_mfg.stride(-2), B, N, M, Cv, dOa, d_mf,
                 128, ps.rep_agg)
        else:
            gamma = agg.gamma
            dmc = d_mfg.contiguous()
            dOa = ops.gma_residual(dmc, dmc, (gamma.detach() - 1.0).contiguous()).view(B, 1, N, Cv)          # gamma * d_mfg
            if ps.dgamma is None:
                ps.dgamma = hip.zeros((1, 1), dev)
            K = B * N * Cv
            AG.gemm(dmc, K, 1, 0, 0, S["Oa"], K, 1, 0, 0, ps.dgamma, 1, 0, 0, 1, 1, 1, 1, K, accumulate=True, ksplit=0, prec=hip.PREC_F16X3)
            d_mf.copy_(d_mfg)
        if docat is not None:
            docat.fill(dOa, cg_off=t * (Cv // 32), batch=dob)
            if len(dob.descs) == 4 or t == T - 1:
                dob.flush()                                          # (4 x 47 MB of dO alive at a time)
        else:
            dOs.append(dOa)
        d_mfs.append(d_mf)
        S.pop("Oa")
    # ---- B: dV of all iterations in one product over the concatenated dO (P is read once), and the operands of the deferred dP
    TC = T * Cv
    dva_cat = E(B, N, M, TC)
    cshift = Cv.bit_length() - 1 if (ppk is not None and Cv >= 32 and Cv & (Cv - 1) == 0) else 0
    if cshift:
        # dV of all iterations as [B, N, T, M, Cv] (CRAFT_PK_CBLK: the modes of a batch entry interleave inside every iteration's column
        # block), so that iteration t's [B*N, M*Cv] operand below is a strided view -- not 12 contiguous copies of 47 MB
        AG.gemm_pk(ppk, ppk.desc(AG.PK_ROWS, M, 1), docat, docat.desc(AG.PK_ROWS, M, 1), dva_cat, M * TC, N * M * TC, Cv, M, B * M, N, TC, N, c_blk_shift=cshift)
        ps.pholder.cat = (docat, ps.vcat)
        ps.vcat = None
    elif ppk is not None:
        AG.gemm_pk(ppk, ppk.desc(AG.PK_ROWS, M, 1), docat, docat.desc(AG.PK_ROWS, M, 1), dva_cat, M * TC, N * M * TC, TC, M, B * M, N, TC, N)
        ps.pholder.cat = (docat, ps.vcat)
        ps.vcat = None
    else:
        dO_cat = torch.cat(dOs, dim=-1) if T > 1 else dOs[0]                                     # [B, M, N, T*Cv]
        del dOs
        AG.gemm(P, 1, ld, M * N * ld, N * ld, dO_cat, 1, TC, M * N * TC, N * TC, dva_cat, M * TC, N * M * TC, TC, M, B * M, N, TC, N, prec=pv)
        V_cat = torch.cat([ps.saved[t]["va"].view(B, N, M, Cv).permute(0, 2, 1, 3) for t in range(T)], dim=-1)    # [B, M, N, T*Cv]
        ps.pholder.cat = (dO_cat, V_cat)
    w_v = agg.first_linear.weight if ps.setrans else agg.to_v.weight
    wv2 = w_v.detach().view(M * Cv, 128)
    dva5 = dva_cat.view(B, N, M, T, Cv)
    for t in range(T):
        S = ps.saved[t]
        pb = AG.PackBatch()
        last = t == T - 1                                    # (the order of phase 2 is free: the accumulators complete with its last iteration)
        mf = ps.HX[t][..., MF:MF + 128]
        d_mf = d_mfs[t]
        if cshift:
            dva = dva_cat.view(B, N, T, M * Cv)[:, :, t, :]             # strided view, row stride T*M*Cv
        else:
            dva = dva5[:, :, :, t, :].reshape(B, N, M * Cv)          # M > 1: a contiguous copy; one mode: a strided view (row stride T*Cv)
        AG.gemm(dva, dva.stride(-2), 1, 0, 0, wv2, 1, 128, 0, 0, d_mf, 128, 0, 0, 1, 1, rows, 128, M * Cv, accumulate=True, prec=pp)      # d_mf += dva W_v
        ps.wgrad(("agg_v",), (AG.Packed(dva, AG.gprec(pp), batch=pb), S["pk_mf"]), 1, 1, ps.acc(("agg_v", "dw"), (M * Cv, 128)), last)
        # ---- BasicMotionEncoder: g_out = ReLU'(mf) (d_mf + the recurrence's share), the two pass-through flow channels carry no gradient
        dv_t = ps.dv[t]
        if os.environ.get("CRAFT_NO_ACT_BWD2"):                 # (A/B: the three launches craft_act_bwd2 replaces)
            d_mf.add_(dv_t[..., 0:128])
            g_out = _act_bwd(d_mf, mf, 128, out=d_mf)
            g_out[..., 126:128] = 0.0
        else:
            g_out = d_mf
            call("craft_act_bwd2", d_mf, 128, dv_t, dv_t.stride(-2), mf, mf.stride(-2), g_out, 128, rows, 128, ACT_RELU, 1.0, 2)
        ps.dv[t] = dv_t = None
        g_cf = _conv_dx(ps, enc.conv.weight, g_out, 128, 3, 3, relu_y=S["cf"])
        ps.wgrad(("menc",), (AG.Packed(g_out, AG.gprec(cp), g3, colsum=ps.acc(("menc", "db"), (128,)), batch=pb), S["pk_cf"]), 3, 3, ps.acc(("menc", "dw"), (128, 3, 3, 256)), last)
        g_c2, g_f2 = g_cf[..., :192], g_cf[..., 192:256]
        g_cor1 = _conv_dx(ps, enc.convc2.weight, g_c2, 192, 3, 3, relu_y=S["cor1"])
        ps.wgrad(("c2",), (AG.Packed(g_c2, AG.gprec(cp), g3, colsum=ps.acc(("c2", "db"), (192,)), batch=pb), S["pk_cor1"]), 3, 3, ps.acc(("c2", "dw"), (192, 3, 3, 256)), last)
        wc1 = enc.convc1.weight.detach().view(256, -1)
        cpl = wc1.shape[1]
        d_corr = E(B, N, cpl)
        AG.gemm(g_cor1, 256, 1, 0, 0, wc1, 1, cpl, 0, 0, d_corr, cpl, 0, 0, 1, 1, rows, cpl, 256, prec=cp)
        ps.wgrad(("c1",), (AG.Packed(g_cor1, AG.gprec(cp), colsum=ps.acc(("c1", "db"), (256,)), batch=pb), S["pk_corr"]), 1, 1, ps.acc(("c1", "dw"), (256, round_up(cpl, 32))), last)
        g_flo1 = _conv_dx(ps, enc.convf2.weight, g_f2, 64, 3, 3, relu_y=S["flo1"])
        ps.wgrad(("f2",), (AG.Packed(g_f2, AG.gprec(cp), g3, colsum=ps.acc(("f2", "db"), (64,)), batch=pb), S["pk_flo1"]), 3, 3, ps.acc(("f2", "dw"), (64, 3, 3, 128)), last)
        ps.wgrad(("f1",), (AG.Packed(g_flo1, AG.gprec(cp), g7, colsum=ps.acc(("f1", "db"), (128,)), batch=pb), S["pk_flow"]), 7, 7, ps.acc(("f1", "dw"), (128, 7, 7, 32)), last)
        # ---- correlation lookup (corr.py:47-71): the gradient goes into the shared buffers of the normalised pyramid; autograd.CorrVolume
        # (every iteration took its token) folds them into the volume's gradient after this node
        AG.lookup_bwd(ps.holders, d_corr, S["coords"], ps.radius)
        pb.flush()
        ps.launch_ready()
        S.clear()
        ps.saved[t] = None


class _CatPack:
    """Two packs over the same rows read as one operand: the channel concatenation [a | b] (craft_wgrad_pk takes the B operand as up
    to two packs).  Used for cat([h, x]) / cat([r*h, x]): x is packed once per GRU pass and shared by both gates."""
    __slots__ = ("a", "b", "rows", "C", "K", "guard", "prec", "rows_p", "C_p", "Wp")

    def __init__(self, a, b):
        assert (a.K, a.guard, a.prec, a.rows_p, a.Wp) == (b.K, b.guard, b.prec, b.rows_p, b.Wp) and a.C_p % 32 == 0
        self.a, self.b = a, b
        self.rows, self.C, self.K, self.guard, self.prec, self.rows_p, self.Wp = a.rows, a.C + b.C, a.K, a.guard, a.prec, a.rows_p, a.Wp
        self.C_p = a.C_p + b.C_p


def _cat_pack(a, b):
    return _CatPack(a, b)


def update_params(model):
    """The parameters of the update block in the order UpdateIter takes (and returns gradients for) them."""
    ub = model.update_block
    enc, gru, fh, agg = ub.encoder, ub.gru, ub.flow_head, ub.aggregator
    ps = [enc.convc1.weight, enc.convc1.bias, enc.convc2.weight, enc.convc2.bias, enc.convf1.weight, enc.convf1.bias, enc.convf2.weight,
          enc.convf2.bias, enc.conv.weight, enc.conv.bias]
    for c in (gru.convz1, gru.convr1, gru.convq1, gru.convz2, gru.convr2, gru.convq2):
        ps += [c.weight, c.bias]
    ps += [fh.conv1.weight, fh.conv1.bias, fh.conv2.weight, fh.conv2.bias, ub.mask[0].weight, ub.mask[0].bias, ub.mask[2].weight, ub.mask[2].bias]
    if model.args.use_setrans:
        ps += [agg.first_linear.weight, agg.feat_softaggr.feat2score.weight, agg.input_skip_coeff]
    else:
        ps += [agg.to_v.weight, agg.gamma]
    return ps


def _param_grads(ps: UpdatePass):
    """The accumulated gradients in the layout of update_params (PyTorch's [Cout, Cin, KH, KW])."""
    m = ps.model
    ub = m.update_block
    enc, gru, fh, agg = ub.encoder, ub.gru, ub.flow_head, ub.aggregator
    A = lambda *k: ps.cache[("acc", k)]             # noqa: E731

    def conv(key, w, has_b=True):
        co, ci = w.shape[0], w.shape[1]
        return [A(*key, "dw")[:co, :, :, :ci].permute(0, 3, 1, 2), A(*key, "db")[:co] if has_b else None]
    out = [A("c1", "dw")[:, : enc.convc1.weight.shape[1]].reshape(enc.convc1.weight.shape), A("c1", "db")]
    out += conv(("c2",), enc.convc2.weight) + conv(("f1",), enc.convf1.weight) + conv(("f2",), enc.convf2.weight) + conv(("menc",), enc.conv.weight)
    for p_ in (0, 1):
        def full(kind):       # input channels [h | inp | mf | mfg]: h, mf, mfg from the per-iteration products, inp from the hoisted one
            var, hoisted = A(kind, p_, "dw"), A(kind, p_, "dw_inp")
            return torch.cat([var[..., :128], hoisted, var[..., 128:]], dim=-1).permute(0, 3, 1, 2)
        zr, zrb = full("zr"), A("zr", p_, "db")
        qw, qb = full("q"), A("q", p_, "db")
        out += [zr[:128], zrb[:128], zr[128:], zrb[128:], qw, qb]
    out += conv(("fh1",), fh.conv1.weight) + conv(("fh2",), fh.conv2.weight) + conv(("mask0",), ub.mask[0].weight)
    out += [A("mask2", "dw").reshape(ub.mask[2].weight.shape), A("mask2", "db")]
    if ps.setrans:
        Cv = agg.first_linear.weight.shape[0] // ps.pholder.P.shape[1]
        red = hip.zeros((Cv + 1,), ps.dev)
        call("craft_reduce_replicas", ps.rep_agg, STATS_REPLICAS, Cv + 1, red)
        out += [A("agg_v", "dw"), red[:Cv].reshape(agg.feat_softaggr.feat2score.weight.shape), red[Cv:].reshape(agg.input_skip_coeff.shape)]
    else:
        out += [A("agg_v", "dw").reshape(agg.to_v.weight.shape), ps.dgamma.view(agg.gamma.shape)]
    return out
