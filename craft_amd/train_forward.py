"""``CRAFT.forward`` under ``model.train()`` (network.py:164-267 with autograd on): the same algorithm as the inference path,
composed from the differentiable HIP operators of ``craft_amd.autograd`` -- materialised scores instead of fused attention,
GRU gates as separate stages -- so that ``loss.backward()`` runs the hand-written backward kernels.

Training-mode semantics of the reference that are reproduced: BatchNorm batch statistics in ``cnet`` unless ``freeze_bn()``
(extractor.py norm_fn='batch'; network.py:136-140), dropout 0.1 on the LayerNorm-ed tokens of every vispos encoder
(setrans.py:791-795) and 0.2 on the attention probabilities of the F2 transformer and the intra-frame attention
(setrans.py:553-557; not on the inter-frame scores, :544-550), ``coords1.detach()`` at the top of every iteration
(network.py:232), all T upsampled predictions returned (test_mode=0).  The two CNN encoders run on the HIP kernels too
(craft_amd/train_encoder.py); ``args.hip_encoders=False`` keeps them as PyTorch-ROCm modules under torch autograd (BASELINE.json
north_star: "Host code stays Python on PyTorch-ROCm for the CNN feature/context extractors").  Trains what the reference's shipped scripts train: ``--craft --f2 full`` with either attention (``--setrans`` or GMA's)
and the plain-correlation GMA model; ``--f1`` and GMA's relative-position scores run in inference only.
"""
from __future__ import annotations

import math
import os

import torch

from . import autograd as AG
from . import hip as hip_mod
from . import ops
from . import train_encoder as TE
from .hip import ACT_NONE, ACT_RELU, ACT_TANH, PREC_BF16, PREC_F16, PREC_F32


_warned_promote = []


def training_precision(prec, loss_scaled: bool = False):
    """The operand modes a training pass runs in.  Plain fp16 MFMA operands need a loss scale: the reference's fp16 autocast training
    relies on GradScaler (train.py:215,231-238) -- unscaled gradients (~3e-7 per pixel at 368x496, batch 8) sit in fp16's subnormal /
    flush range and degrade silently in dP = dO V^T, dV = P^T dO and the convolution gradients.  ``loss_scaled`` (set by train.Trainer,
    whose step multiplies the loss gradient by a power of two and un-scales inside the optimizer: ``args.hip_loss_scaled``): fp16 roles
    run as asked.  Otherwise (a bare ``loss.backward()``) every role that asks for fp16 (e.g. ``pv`` of the inference default "mixed",
    which args.mixed_precision=True selects) runs in f16x3 instead: fp32-class results, same fp16 MFMA pipe (f16x3's planes are fp16 too:
    it needs the loss scale just as much at full image sizes, tests/test_train_backward.py).  bf16 roles keep bf16 (fp32's exponent
    range)."""
    from .hip import PREC_F16X3, Precision
    roles = [r for r in Precision.__slots__ if getattr(prec, r) == PREC_F16]
    if not roles or loss_scaled:
        return prec
    if not _warned_promote:
        _warned_promote.append(1)
        import warnings
        warnings.warn(f"craft_amd training: fp16 operand mode of role(s) {roles} promoted to f16x3 (no loss scale announced: train.Trainer "
                      "sets args.hip_loss_scaled; a bare loss.backward() should use a bf16 policy such as train_amp_bf16 for 16-bit operands)")
    out = Precision(prec.proj, prec.score, prec.pv, prec.conv, prec.enc, prec.wgx, prec.wgy, prec.dxw, prec.sbw)
    for r in roles:
        setattr(out, r, None if r in Precision.BACKWARD_ROLES else PREC_F16X3)
    return out


def use_pk_attention(prec) -> bool:
    """The attention products on packed operands (craft_gemm_pk): every 16-bit / f16x3 `pv` mode; CRAFT_NO_PK=1 keeps the fp32-source engine."""
    return AG.pick(prec, "pv") != PREC_F32 and not os.environ.get("CRAFT_NO_PK") and not os.environ.get("CRAFT_NO_PK_ATTN")


def _attention_probs(module, x_ln, hw, prec, p_attn: float, seed: int, pk_box=None):
    """CrossAttFeatTrans up to (and including) the dropout of the probabilities (setrans.py:507-557) -> P [B, M, N, ld].
    pk_box (a list): the dropped probabilities are produced as a packed operand appended to it, the returned tensor is P's autograd handle."""
    st = module.setrans
    q = AG.Linear.apply(x_ln, st.query.weight, st.query.bias, prec)
    k = AG.Linear.apply(x_ln, st.key.weight, st.key.bias, prec)
    M = st.num_modes
    scale = 1.0 / math.sqrt(st.attention_mode_dim)
    mx = ops.score_max(q.detach(), k.detach(), hw[0], hw[1], M, scale, prec)
    # scores and their gradients on packed operands too -- for the single-plane modes (bf16 / fp16: dS leaves the softmax backward in half
    # the bytes; configs[4] -1 ms).  In f16x3 the three products gain 165 us per attention and the packed dS store costs them again
    # (measured +2 ms per step at configs[3] with it on: CRAFT_PK_SCORES=1 forces it for A/B runs)
    sp_ = AG.pick(prec, "score")
    link = None
    if use_pk_attention(prec) and not os.environ.get("CRAFT_NO_PK_SCORES") and (sp_ in (PREC_BF16, PREC_F16) or getattr(prec, "sbw", None) is not None):
        link = AG.ScoreLink()
    S = AG.Scores.apply(q, k, M, scale, prec, link)
    pk = None
    if pk_box is not None:
        pk = AG.PkMat(S.shape[0] * S.shape[1], S.shape[2], S.shape[3], AG.pick(prec, "pv"), S.device)
        pk_box.append(pk)
    return AG.AttnSoftmax.apply(S, module.vispos_encoder.pos_table, float(st.pos_code_weight), int(module.attn_mask_radius), mx, hw,
                                float(p_attn), int(seed), pk, link)


def _conv(x, conv, hw, act, prec, cache):
    return AG.Conv.apply(x, conv.weight, conv.bias, hw, act, prec, cache)


def _touch_cancelling_biases(model, preds):
    """The bias of a softmax-over-modes score (``*.feat2score.bias``, setrans.py:448-458, 495-498) cancels in the softmax: the kernels
    never read it, its gradient is mathematically zero -- and torch autograd hands the reference's optimizer exactly that, a zero
    TENSOR (weight decay still applies), not None.  Tie those parameters to the last prediction with a zero-gradient edge so that any
    optimizer wrapped around this class (torch.optim.AdamW under GradScaler / DDP, tests/test_reference_wrappers.py) treats them like
    the reference does.  (The intra-frame attention's pooling never runs at all -- the reference leaves its gradients None too:
    ``train.unused_parameters``.)"""
    ps = model.__dict__.get("_cancelling_biases")
    if ps is None:
        from .train import unused_parameters
        skip = {id(p) for p in unused_parameters(model)}
        ps = model.__dict__["_cancelling_biases"] = [p for n, p in model.named_parameters() if n.endswith("feat2score.bias") and id(p) not in skip]
    live = [p for p in ps if p.requires_grad]
    if live:
        preds = list(preds)
        preds[-1] = AG.ZeroGradEdge.apply(preds[-1], *live)
    return preds


def forward_train(model, image1, image2, iters=12, flow_init=None):
    args = model.args
    # the reference's four shipped training scripts: --craft --f2 full --setrans (train-craft-f2full.sh), --craft --f2 full with GMA's
    # attention / aggregator (train-craft-f2full-gma.sh), plain correlation + GMA (train-gma.sh), and any mix of those three switches
    if args.f2trans == "none":
        raise NotImplementedError("--f2 none: the reference's own constructor fails without the F2 transformer (network.py:93-106)")
    prec = training_precision(model.hip_prec(), bool(getattr(args, "hip_loss_scaled", False)))
    AG.set_backward_modes(prec)                      # operand modes of the backward products for this pass (roles wgx / wgy / dxw)
    B, _, H, W = image1.shape
    if H % 8 or W % 8:
        raise ValueError("image height and width must be multiples of 8")
    H8, W8 = H // 8, W // 8
    N = H8 * W8
    if N % 4:
        raise ValueError(f"training needs (H/8)*(W/8) to be a multiple of 4 (got {H8}x{W8}): the backward GEMMs read 16-byte vectors")
    hw = (H8, W8)
    dev = image1.device
    step = model.__dict__.setdefault("_train_calls", 0)
    model.__dict__["_train_calls"] = step + 1
    base_seed = (torch.initial_seed() * 1000003 + step * 64) & 0x7FFFFFFFFFFFFFF

    def p_hidden(cfg):
        return float(getattr(args, "dropout_prob", -1)) if getattr(args, "dropout_prob", -1) >= 0 else float(cfg.hidden_dropout_prob)

    def p_attn(cfg):
        return float(getattr(args, "dropout_prob", -1)) if getattr(args, "dropout_prob", -1) >= 0 else float(cfg.attention_probs_dropout_prob)

    # ---- CNN encoders (network.py:169-183, :203) -----------------------------------------------------------------------
    # default: on the HIP kernels in the policy's `enc` operand mode (craft_amd/train_encoder.py).  args.hip_encoders=False keeps the
    # PyTorch-ROCm modules under torch autograd -- with enc = bf16 under autocast, like the reference's mixed-precision training
    # (network.py:179,199); fp16 roles were promoted to f16x3 above (training_precision): no loss scaling is built
    # the step's small zero-initialised buffers come from one allocation (hip.ZeroPool; valid through this pass's backward)
    pool = model.__dict__.get("_zero_pool")
    if pool is None:
        pool = model.__dict__["_zero_pool"] = hip_mod.ZeroPool()
    pool.begin(dev)
    hip_mod.set_zero_pool(None if os.environ.get("CRAFT_NO_ZERO_POOL") else pool)
    H, W = image1.shape[-2:]
    use_henc = getattr(args, "hip_encoders", True) and TE.supported(model.fnet, H, W) and TE.supported(model.cnet, H, W)
    if use_henc:
        B = image1.shape[0]
        f12 = TE.encoder_forward_train(model.fnet, torch.cat([image1, image2], dim=0).float(), prec)
        f1_tok, f2_tok = f12[:B], f12[B:]
        cn_tok = TE.encoder_forward_train(model.cnet, image1.float(), prec)
    else:
        im1 = (2 * (image1.float() / 255.0) - 1.0).contiguous()
        im2 = (2 * (image2.float() / 255.0) - 1.0).contiguous()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=prec.enc == PREC_BF16):
            fmap1, fmap2 = model.fnet([im1, im2])
            cnet_feat = model.cnet(im1)
        f1_tok = AG.NchwToTokens.apply(fmap1.float())
        f2_tok = AG.NchwToTokens.apply(fmap2.float())
        cn_tok = AG.NchwToTokens.apply(cnet_feat.float())
    net = AG.TokensNorm.apply(cn_tok[..., 0:128], ACT_TANH, False)              # network.py:209-211
    inp = AG.TokensNorm.apply(cn_tok[..., 128:256], ACT_RELU, False)

    # ---- F2 transformer (network.py:185-187; setrans.py:578-619, 364-410); with --f1 shared | private the same block (the same
    # module or a private one, network.py:94-103) transforms frame 1 too (:180-183)
    def ln_tok(enc, tok, positions=None):
        """SETransInputFeatEncoder (setrans.py:763-800): LayerNorm of the tokens ('bias'), or of tokens + pos_code_weight * E ('lsinu': a
        small differentiable torch branch -- pos_fc's gradients come from torch autograd, the LayerNorm backward is the kernel's)."""
        e = enc.embedding(hw, tok.device, positions)
        return AG.TokensNorm.apply(tok if e is None else tok + e, ACT_NONE, True)

    def feature_transformer(mod, tok, seed):
        c = mod.config
        x = AG.dropout(ln_tok(mod.vispos_encoder, tok), p_hidden(c), seed)
        ot = mod.setrans.out_trans
        pkb = [] if (use_pk_attention(prec) and (ot.first_linear.weight.shape[0] // mod.setrans.num_modes) % 32 == 0) else None
        Pm = _attention_probs(mod, x, hw, prec, p_attn(c), seed + 1, pkb)
        v = AG.Linear.apply(x, ot.first_linear.weight, None, prec)
        return AG.ModePoolLN.apply(AG.AttnApply.apply(Pm, v, prec, pkb[0] if pkb else None), x, ot.feat_softaggr.feat2score.weight, ot.input_skip_coeff)

    fmap2_t = feature_transformer(model.f2_trans, f2_tok, base_seed + 1)
    f1t = getattr(model, "f1_trans", None)
    fmap1_t = feature_transformer(f1t, f1_tok, base_seed + 9) if (f1t is not None and args.craft) else None

    # ---- inter-frame correlation volume + pyramid (network.py:225-228; corr.py:148-207) -------------------------------
    box = []
    if args.craft:
        cf = model.corr_fn
        cc = cf.config
        st = cf.setrans
        scale = 1.0 / math.sqrt(st.attention_mode_dim)
        w_aggr = st.attn_softaggr.feat2score.weight if st.num_modes > 1 else torch.ones(1, 1, device=dev)

        pos1 = None                               # frame 1 is encoded at coords1 = grid + flow_init (corr.py:153); training recomputes the
        if flow_init is not None and cf.vispos_encoder.pos_code_type != "bias":        # code per call, so frame 2 stays on the grid
            ys, xs = torch.meshgrid(torch.arange(H8, device=dev), torch.arange(W8, device=dev), indexing="ij")
            pos1 = torch.stack([ys, xs], dim=-1).reshape(1, N, 2).float() + ops.tokens_from_nchw(flow_init.detach().float()).flip(-1)

        def vispos(tok, seed, positions=None):    # the correlation block's own input encoder: (+ embedding) LayerNorm + dropout (setrans.py:791-795)
            return AG.dropout(ln_tok(cf.vispos_encoder, tok, positions), p_hidden(cc), seed)

        def volume(xq, xk):
            q = AG.Linear.apply(xq, st.query.weight, st.query.bias, prec)
            k = AG.Linear.apply(xk, st.key.weight, st.key.bias, prec)
            mx = ops.score_max(q.detach(), k.detach(), H8, W8, st.num_modes, scale, prec)
            Sc = AG.Scores.apply(q, k, st.num_modes, scale, prec)
            return AG.CorrVolume.apply(Sc, cf.vispos_encoder.pos_table, w_aggr, float(st.pos_code_weight), mx, hw, box,
                                       bool(cf.do_corr_global_norm))

        if fmap1_t is not None:
            # two-way correlation (corr.py:164-171): (transformed 1, conv 2) and (conv 1, transformed 2), concatenated per level
            token = (volume(vispos(fmap1_t, base_seed + 3, pos1), vispos(f2_tok, base_seed + 11))
                     + volume(vispos(f1_tok, base_seed + 12, pos1), vispos(fmap2_t, base_seed + 4)))
        else:
            token = volume(vispos(f1_tok, base_seed + 3, pos1), vispos(fmap2_t, base_seed + 4))
        radius = cf.radius
    else:
        # CorrBlock (corr.py:17-45, :73-81): <fmap1, fmap2> / sqrt(256), no positional bias, no global LayerNorm, avg-pool pyramid
        Sc = AG.Scores.apply(f1_tok, fmap2_t, 1, 1.0 / math.sqrt(256.0), prec)
        token = AG.CorrVolume.apply(Sc, None, torch.ones(1, 1, device=dev), 0.0, None, hw, box, False)
        radius = int(args.corr_radius)
    holder = box if len(box) > 1 else box[0]          # (two volumes with --f1)

    # ---- intra-frame attention (network.py:214): computed once, used by every iteration ------------------------------
    att = model.att
    fused = getattr(args, "hip_fused_update", True) and prec.conv != PREC_F32 and not os.environ.get("CRAFT_TRAIN_UNFUSED")
    agg_ = model.update_block.aggregator
    if getattr(agg_, "project", None) is not None:
        fused = False          # gma.Aggregate with --num_heads > 1 (head merge + project, gma.py:133-137): the operator-level path below
    cv_agg = (agg_.first_linear.weight.shape[0] // att.setrans.num_modes) if args.use_setrans else agg_.dim_head
    apk = [] if (fused and use_pk_attention(prec) and cv_agg % 32 == 0) else None       # (the packed P is consumed by train_update only)
    if args.use_setrans:
        ca = att.config
        xc = AG.dropout(ln_tok(att.vispos_encoder, inp), p_hidden(ca), base_seed + 5)
        Patt = _attention_probs(att, xc, hw, prec, p_attn(ca), base_seed + 6, apk)
    else:
        # gma.Attention (gma.py:53-102): softmax(scale * q k^T) of the 1x1-conv projections of the context features, no dropout
        inner = att.heads * att.dim_head
        qk = AG.Linear.apply(inp, att.to_qk.weight.view(2 * inner, -1), None, prec)
        pos_only = bool(getattr(args, "position_only", False))
        q_ = qk[..., :inner]
        Sg = AG.Scores.apply(q_, qk[..., inner:], att.heads, 0.0 if pos_only else float(att.scale), prec)
        if pos_only or getattr(args, "position_and_content", False):
            # RelPosEmb (gma.py:21-50): (scale q)(x, y) . E_h[u - x] + (scale q)(x, y) . E_w[v - y] -- per query a row of 2 H8 - 1 and one
            # of 2 W8 - 1 scores (two small products against the embedding rows of the offsets that can occur), added to the scores
            if max(H8, W8) > att.max_pos_size:
                raise ValueError(f"feature map {H8}x{W8} exceeds RelPosEmb max_pos_size {att.max_pos_size}")
            P0 = att.max_pos_size - 1
            pad4 = lambda e: torch.nn.functional.pad(e, (0, 0, 0, (-e.shape[0]) % 4))          # noqa: E731  (row counts % 4: vector loads)
            Eh = pad4(att.pos_emb.rel_height.weight[P0 - (H8 - 1): P0 + H8])
            Ew = pad4(att.pos_emb.rel_width.weight[P0 - (W8 - 1): P0 + W8])
            qh = q_.reshape(B, N, att.heads, att.dim_head).permute(0, 2, 1, 3).reshape(B * att.heads, N, att.dim_head)
            Hs = AG.Linear.apply(qh, Eh, None, prec).view(B, att.heads, N, -1)
            Ws = AG.Linear.apply(qh, Ew, None, prec).view(B, att.heads, N, -1)
            Sg = AG.RelPosAdd.apply(Sg, Hs, Ws, float(att.scale) * (1.0 if pos_only else float(att.pos_embed_weight)), hw)
        if apk is not None:
            apk.append(AG.PkMat(Sg.shape[0] * Sg.shape[1], Sg.shape[2], Sg.shape[3], AG.pick(prec, "pv"), Sg.device))
        Patt = AG.AttnSoftmax.apply(Sg, None, 0.0, -1, None, hw, 0.0, 0, apk[0] if apk else None)
    pbox = []
    ptoken = AG.ProbsToken.apply(Patt, pbox, prec, apk[0] if apk else None)        # the 12 uses of Patt share ONE gradient product
    pholder = pbox[0]

    # ---- iterative refinement (network.py:230-260; update.py:137-162) -------------------------------------------------
    radius_ = radius
    if fused:
        # one autograd node per iteration with a hand-written backward (craft_amd/train_update.py)
        from . import train_update as TU
        coords0, coords1, _ = ops.coords_init(flow_init, B, H8, W8, dev)
        ups = TU.UpdatePass(model, prec, hw, B, iters, net, inp, holder, pholder, radius_)
        model.__dict__["_train_pass_cache"] = ups.cache
        params = TU.update_params(model)
        preds = []
        for t in range(iters):       # (the correlation lookup of network.py:235 is part of the node: coords1 carries no gradient, :232)
            net, up, coords1 = TU.UpdateIter.apply(net, token, ptoken, inp, ups, t, coords1, coords0, *params)
            preds.append(up)
        return _touch_cancelling_biases(model, preds)
    ub = model.update_block
    enc, gru, fh, agg = ub.encoder, ub.gru, ub.flow_head, ub.aggregator
    coords0, coords1, _ = ops.coords_init(flow_init, B, H8, W8, dev)
    preds = []
    wcache = {}          # conv operands (padded / packed / transposed weights) built once for this pass and its backward
    model.__dict__["_train_pass_cache"] = wcache            # (autograd.pending_uses: Trainer.step checks it after backward)
    wzr = [torch.cat([gru.convz1.weight, gru.convr1.weight], 0), torch.cat([gru.convz2.weight, gru.convr2.weight], 0)]
    bzr = [torch.cat([gru.convz1.bias, gru.convr1.bias], 0), torch.cat([gru.convz2.bias, gru.convr2.bias], 0)]
    convq = [gru.convq1, gru.convq2]
    wc1 = enc.convc1.weight.view(256, -1)
    wm2 = ub.mask[2].weight.view(576, -1)
    for _ in range(iters):
        coords1 = coords1.detach()                                              # network.py:232
        corr = AG.CorrLookup.apply(token, coords1, holder, radius)              # :235
        flow = coords1 - coords0
        # BasicMotionEncoder (update.py:79-87)
        cor = AG.Act.apply(AG.Linear.apply(corr, wc1, enc.convc1.bias, prec.conv, wcache), ACT_RELU, 1.0)
        cor = _conv(cor, enc.convc2, hw, ACT_RELU, prec, wcache)
        flo = _conv(flow, enc.convf1, hw, ACT_RELU, prec, wcache)
        flo = _conv(flo, enc.convf2, hw, ACT_RELU, prec, wcache)
        out = _conv(torch.cat([cor, flo], dim=-1), enc.conv, hw, ACT_RELU, prec, wcache)
        mf = torch.cat([out, flow], dim=-1)                                     # [B, N, 128]
        # motion aggregator (update.py:143-149): ExpandedFeatTrans on the raw motion features
        if args.use_setrans:
            va = AG.Linear.apply(mf, agg.first_linear.weight, None, prec, wcache)
            Oa = AG.AttnApplyShared.apply(ptoken, va, pholder, prec)
            mfg = AG.ModePoolLN.apply(Oa, mf, agg.feat_softaggr.feat2score.weight, agg.input_skip_coeff)
        else:                                                                   # gma.Aggregate (gma.py:128-140), one head
            va = AG.Linear.apply(mf, agg.to_v.weight.view(agg.heads * agg.dim_head, -1), None, prec, wcache)
            Oa = AG.AttnApplyShared.apply(ptoken, va, pholder, prec)            # [B, heads, N, 128]
            if agg.project is not None:                                         # 'b h (x y) d -> b (h d) x y' + 1x1 projection (gma.py:135-138)
                Oa = AG.Linear.apply(Oa.permute(0, 2, 1, 3).reshape(B, N, -1), agg.project.weight.view(agg.project.weight.shape[0], -1), None, prec,
                                     wcache)
            mfg = AG.GmaResidual.apply(mf, Oa.reshape(B, N, -1), agg.gamma)
        # SepConvGRU (update.py:49-64)
        x = torch.cat([inp, mf, mfg], dim=-1)                                   # [B, N, 384]
        h = net
        for ps, (kh, kw) in enumerate(((1, 5), (5, 1))):
            zr_pre = AG.Conv.apply(torch.cat([h, x], dim=-1), wzr[ps], bzr[ps], hw, ACT_NONE, prec, wcache)
            z, rh = AG.GruZR.apply(zr_pre, h)
            q_pre = _conv(torch.cat([rh, x], dim=-1), convq[ps], hw, ACT_NONE, prec, wcache)
            h = AG.GruOut.apply(q_pre, z, h)
        net = h
        # heads (update.py:15-16, :124-127, :161)
        delta = _conv(_conv(net, fh.conv1, hw, ACT_RELU, prec, wcache), fh.conv2, hw, ACT_NONE, prec, wcache)
        mh = _conv(net, ub.mask[0], hw, ACT_RELU, prec, wcache)
        mask = AG.Act.apply(AG.Linear.apply(mh, wm2, ub.mask[2].bias, prec.conv, wcache), ACT_NONE, 0.25)
        coords1 = coords1 + delta                                               # network.py:247
        preds.append(AG.ConvexUpsample.apply(mask, coords1 - coords0, hw))      # :258
    return _touch_cancelling_biases(model, preds)
