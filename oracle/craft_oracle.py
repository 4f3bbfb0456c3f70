"""CPU oracle for CRAFT's hot path — TEST INFRASTRUCTURE, NOT THE PRODUCT.

A functional fp32 restatement (torch-CPU / numpy, no nn.Module; differentiable by torch autograd, which is how the
gradient tests use it: ``craft_train_forward``) of the reference's inner-loop algorithm, written from SURVEY.md Appendix A and the cited reference lines.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the
product path (``craft_amd``) never does and fails loudly when ``libcraft_hip.so`` is missing.

Pinning: every function here is checked in ``tests/test_oracle_golden.py`` against tensors captured
from the imported reference (``tools/make_golden.py`` -> ``tests/golden/*.npz``), so parity claims
made against this oracle are pinned to the reference itself (the reference has no tests of its
own).

All citations ``file:line`` are relative to the reference root (``/root/reference``).
Weights are addressed by the reference's ``state_dict`` key names.

Formulation differences from the reference (all verified equal to rounding by the goldens):
  * the sliding positional bias is computed from (dh, dw) instead of a materialised index_put;
  * softmax-over-modes pooling of scores uses the closed form c = sum_m s_m softmax_m(w s_m);
  * the correlation pyramid is stored un-normalised and the global LayerNorm is applied lazily
    inside the lookup (exactly equivalent incl. zero padding, see ``corr_lookup``);
  * the lookup is an explicit 4-tap bilinear gather, not ``F.grid_sample``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
LN_EPS = 1e-12          # setrans.py:715 (comb_norm_layer), :362 (skip_layer_norm), corr.py:203
ATTN_CLIP = 100.0       # setrans.py:98  (attn_clip)

# The reference's model.train() forward applies nn.Dropout at six places of the canonical configuration: to the LayerNorm-ed tokens of
# every SETransInputFeatEncoder (setrans.py:791-795, hidden_dropout_prob 0.1) and to the attention probabilities of the two self-attentions
# (setrans.py:553-557, attention_probs_dropout_prob 0.2).  Its masks come from torch's generator and cannot be replayed; the functions below
# take them as DATA instead: a dict {site: mask already scaled by 1 / (1 - p), broadcastable to the site's tensor}, or None (no dropout).
# Sites: "<prefix>.hidden" / "<prefix>.attn" for prefix in f2_trans, f1_trans, att; "corr_fn.x1" / "corr_fn.x2" for the correlation
# block's encoder applied to frame 1 / frame 2.  Set by the tests around a call (tests/test_train_dropout_parity.py).
DROPOUT_MASKS: Optional[Dict[str, "torch.Tensor"]] = None


def _drop(t: "torch.Tensor", site: Optional[str]) -> "torch.Tensor":
    if DROPOUT_MASKS is None or site is None or site not in DROPOUT_MASKS:
        return t
    return t * DROPOUT_MASKS[site]


@dataclass
class OracleConfig:
    """The argparse fields that reach the model (network.py:27-134, train.py:311-406)."""
    craft: bool = True
    use_setrans: bool = True
    f1trans: str = "none"
    f2trans: str = "full"
    corr_radius: int = 4
    corr_levels: int = 4
    pos_bias_radius: int = 7
    inter_num_modes: int = 4
    intra_num_modes: int = 4
    f2_num_modes: int = 4
    inter_pos_code_weight: float = 0.5
    intra_pos_code_weight: float = 1.0
    f2_pos_code_weight: float = 0.5
    f2_attn_mask_radius: int = -1
    num_heads: int = 1
    position_only: bool = False
    position_and_content: bool = False
    inter_pos_code_type: str = "bias"       # 'bias' | 'lsinu' (train.py --interpos / --intrapos); the functions below read the type off the
    intra_pos_code_type: str = "bias"       # state dict's keys (vispos_tokens), these two only let a test pass its overrides through
    extra: dict = field(default_factory=dict)


# ------------------------------------------------------------------------------------------------
# tokens / LayerNorm / positional bias
# ------------------------------------------------------------------------------------------------
def tokens_layernorm(x_nchw: Tensor) -> Tensor:
    """SETransInputFeatEncoder.forward, 'bias' pos-code type (setrans.py:763-800):
    NCHW -> [B, N, C], LayerNorm over C without affine (eps 1e-12); pos_embed is 0 (:776)."""
    B, C, H, W = x_nchw.shape
    t = x_nchw.reshape(B, C, H * W).transpose(1, 2)
    return layernorm_lastdim(t)


def vispos_tokens(x_nchw: Tensor, sd: Dict[str, Tensor], prefix: str, pos_w: float, positions: Optional[Tensor] = None,
                  drop_site: Optional[str] = None) -> Tensor:
    """SETransInputFeatEncoder.forward (setrans.py:763-800) for either positional-code type, chosen by the state dict's keys:
    'bias' (``<prefix>.vispos_encoder.pos_coder.biases``): LayerNorm of the tokens, the table goes to the scores.
    'lsinu' (``...pos_coder.pos_fc.weight``): tokens + pos_w * E(p / max p) before the LayerNorm, E = LearnedSinuPosEmbedder
    (setrans.py:624-646: LayerNorm(interlace(sin(fc(p)[0::2]), cos(fc(p)[1::2]))), no affine, omega = 1); p = (y, x) grid indices
    (gen_all_indices, setrans.py:32-39) or the given positions [B, N, 2]."""
    B, C, H, W = x_nchw.shape
    t = x_nchw.reshape(B, C, H * W).transpose(1, 2)
    wk = f"{prefix}.vispos_encoder.pos_coder.pos_fc.weight"
    if wk in sd:
        if positions is None:
            ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
            positions = torch.stack([ys, xs], dim=-1).reshape(1, H * W, 2).to(t.dtype).expand(B, -1, -1)
        pn = positions / positions.max()                          # setrans.py:772
        e0 = F.linear(pn, sd[wk], sd[f"{prefix}.vispos_encoder.pos_coder.pos_fc.bias"])
        mix = torch.stack((torch.sin(e0[..., 0::2]), torch.cos(e0[..., 1::2])), dim=-1).reshape(e0.shape)
        t = t + pos_w * layernorm_lastdim(mix)
    return _drop(layernorm_lastdim(t), drop_site)                # (training: self.dropout(feat_normed), setrans.py:791-795)


def pos_table_matrix(sd: Dict[str, Tensor], prefix: str, H8: int, W8: int):
    """pos_w-free positional bias [N, N] of a 'bias' encoder; 0.0 for 'lsinu' (setrans.py:781: pos_biases = None)."""
    k = f"{prefix}.vispos_encoder.pos_coder.biases"
    return pos_bias_matrix(sd[k], H8, W8) if k in sd else 0.0


def layernorm_lastdim(t: Tensor) -> Tensor:
    mu = t.mean(dim=-1, keepdim=True)
    var = ((t - mu) ** 2).mean(dim=-1, keepdim=True)        # biased variance
    return (t - mu) / torch.sqrt(var + LN_EPS)


def pos_bias_matrix(biases: Tensor, H8: int, W8: int) -> Tensor:
    """SlidingPosBiases2D.forward (setrans.py:690-708) as a closed form:
    pb[(h1,w1),(h2,w2)] = biases[h2-h1+R, w2-w1+R] if |dh|<=R and |dw|<=R else 0.  -> [N, N]"""
    R = (biases.shape[0] - 1) // 2
    hs = torch.arange(H8)
    ws = torch.arange(W8)
    dh = hs[None, :] - hs[:, None]                           # [h1, h2]
    dw = ws[None, :] - ws[:, None]                           # [w1, w2]
    okh = dh.abs() <= R
    okw = dw.abs() <= R
    ih = (dh + R).clamp(0, 2 * R)
    iw = (dw + R).clamp(0, 2 * R)
    pb = biases[ih[:, None, :, None], iw[None, :, None, :]]  # [h1, w1, h2, w2]
    pb = pb * (okh[:, None, :, None] & okw[None, :, None, :]).to(pb.dtype)
    return pb.reshape(H8 * W8, H8 * W8)


def chebyshev_mask(H8: int, W8: int, radius: int) -> Optional[Tensor]:
    """SelfAttVisPosTrans.forward attention mask (setrans.py:580-584): -1e9 where the Chebyshev
    distance between the two grid positions exceeds ``radius``; None when radius <= 0."""
    if radius <= 0:
        return None
    hs = torch.arange(H8)
    ws = torch.arange(W8)
    dh = (hs[None, :] - hs[:, None]).abs()
    dw = (ws[None, :] - ws[:, None]).abs()
    d = torch.maximum(dh[:, None, :, None].expand(H8, W8, H8, W8), dw[None, :, None, :].expand(H8, W8, H8, W8))
    return ((d > radius).float() * -1e9).reshape(H8 * W8, H8 * W8)


# ------------------------------------------------------------------------------------------------
# multi-mode attention scores (CrossAttFeatTrans, setrans.py:501-566)
# ------------------------------------------------------------------------------------------------
def mm_scores(xq: Tensor, xk: Tensor, Wq: Tensor, bq: Optional[Tensor], Wk: Tensor, bk: Optional[Tensor],
              M: int) -> Tensor:
    """Q,K Linear -> split M modes -> Q K^T / sqrt(d)   (setrans.py:507-515)  -> [B, M, Nq, Nk]"""
    B, Nq, C = xq.shape
    q = F.linear(xq, Wq, bq)
    k = F.linear(xk, Wk, bk)
    d = q.shape[-1] // M
    q = q.reshape(B, Nq, M, d).permute(0, 2, 1, 3)
    k = k.reshape(B, xk.shape[1], M, d).permute(0, 2, 1, 3)
    return torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d)


def clamp_rule(S: Tensor) -> Tensor:
    """setrans.py:520-529: if the global max (over batch, modes, i, j) exceeds attn_clip, clamp the
    whole tensor to [-100, 100]; only the positive max is tested."""
    if float(S.detach().max()) > ATTN_CLIP:
        return S.clamp(-ATTN_CLIP, ATTN_CLIP)
    return S


def softaggr_scores(S: Tensor, w: Tensor) -> Tensor:
    """LearnedSoftAggregate(num_feat=1, group_dim=1) on scores (setrans.py:279-300, :545-550):
    out = sum_m s_m * softmax_m(w*s_m + b); the bias b cancels in the softmax.  [B,M,N,N]->[B,N,N]"""
    p = torch.softmax(S * w.reshape(()), dim=1)
    return (S * p).sum(dim=1)


# ------------------------------------------------------------------------------------------------
# A2. inter-frame correlation volume + pyramid (corr.py:148-207)
# ------------------------------------------------------------------------------------------------
def inter_corr_raw(fmap1: Tensor, fmap2: Tensor, sd: Dict[str, Tensor], cfg: OracleConfig, positions1: Optional[Tensor] = None,
                   positions2: Optional[Tensor] = None) -> Tensor:
    """TransCorrBlock.corr before the global LayerNorm: [B, N, N] un-normalised c(i,j).  positions1 / positions2: the two frames' (y, x)
    positions when they are not the grid (only the 'lsinu' positional code reads them: coords1 = grid + flow_init, corr.py:153)."""
    B, C, H8, W8 = fmap1.shape
    x1 = vispos_tokens(fmap1, sd, "corr_fn", cfg.inter_pos_code_weight, positions1, drop_site="corr_fn.x1")
    x2 = vispos_tokens(fmap2, sd, "corr_fn", cfg.inter_pos_code_weight, positions2, drop_site="corr_fn.x2")
    W = sd["corr_fn.setrans.query.weight"]
    b = sd.get("corr_fn.setrans.query.bias")
    S = mm_scores(x1, x2, W, b, W, b, cfg.inter_num_modes)           # tied projection (:475-478)
    S = clamp_rule(S)
    pb = pos_table_matrix(sd, "corr_fn", H8, W8)
    S = S + cfg.inter_pos_code_weight * pb                            # setrans.py:538-540
    if cfg.inter_num_modes > 1:
        return softaggr_scores(S, sd["corr_fn.setrans.attn_softaggr.feat2score.weight"])
    return S[:, 0]


def plain_corr_raw(fmap1: Tensor, fmap2: Tensor) -> Tensor:
    """CorrBlock.corr (corr.py:73-81): <fmap1(:,i), fmap2(:,j)> / sqrt(C), no norm.  [B,N,N]"""
    B, C, H8, W8 = fmap1.shape
    a = fmap1.reshape(B, C, H8 * W8)
    b = fmap2.reshape(B, C, H8 * W8)
    return torch.matmul(a.transpose(1, 2), b) / math.sqrt(C)


def global_stats(c: Tensor) -> Tuple[Tensor, Tensor]:
    """mean and 1/sqrt(var+eps) over all N*N entries per sample (corr.py:200-204)."""
    B = c.shape[0]
    flat = c.reshape(B, -1).double()
    mu = flat.mean(dim=1)
    var = ((flat - mu[:, None]) ** 2).mean(dim=1)
    return mu.to(c.dtype), (1.0 / torch.sqrt(var + LN_EPS)).to(c.dtype)     # (c.dtype: float32 everywhere except float64 conditioning studies)


def build_pyramid(c: Tensor, H8: int, W8: int, levels: int = 4) -> List[Tensor]:
    """corr.py:178,186-189: [B,N,N] -> [B*N,1,H8,W8] and 3x avg_pool2d(2,2) (floor sizes)."""
    B, N, _ = c.shape
    lvl = c.reshape(B * N, 1, H8, W8)
    pyr = [lvl]
    for _ in range(levels - 1):
        h, w = lvl.shape[-2] // 2, lvl.shape[-1] // 2
        lvl = lvl[..., : 2 * h, : 2 * w].reshape(B * N, 1, h, 2, w, 2).mean(dim=(3, 5))
        pyr.append(lvl)
    return pyr


def corr_lookup(pyr: List[Tensor], coords: Tensor, radius: int,
                mu: Optional[Tensor] = None, rstd: Optional[Tensor] = None) -> Tensor:
    """CorrBlock.__call__ + bilinear_sampler (corr.py:47-71, utils.py:65-79), explicit form.

    channel k = l*(2r+1)^2 + a*(2r+1) + b samples level l at
        X = coords_x / 2^l + (a - r),   Y = coords_y / 2^l + (b - r)
    (the x offset varies along the FIRST window axis: corr.py:57 builds delta=(dy,dx) meshgrid and adds
    it to (x,y) centroids), bilinear with zero padding, align_corners=True in pixel coordinates.

    ``pyr`` holds the UN-normalised pyramid; with (mu, rstd) the global LayerNorm is applied lazily:
    sample(norm(c)) = (sample(c) - mu * sum_of_in_bounds_weights) * rstd, because zero padding
    contributes 0 to both and avg-pooling commutes with the affine map.   -> [B, L*(2r+1)^2, H8, W8]
    """
    B, _, H8, W8 = coords.shape
    N = H8 * W8
    r = radius
    win = 2 * r + 1
    cx = coords[:, 0].reshape(B * N)
    cy = coords[:, 1].reshape(B * N)
    offs = torch.arange(-r, r + 1, dtype=coords.dtype)
    out = []
    rows = torch.arange(B * N)
    for l, lvl in enumerate(pyr):
        h, w = lvl.shape[-2:]
        img = lvl.reshape(B * N, h * w)
        X = (cx / (2 ** l))[:, None, None] + offs[None, :, None]          # varies along a (first axis)
        Y = (cy / (2 ** l))[:, None, None] + offs[None, None, :]          # varies along b
        X = X.expand(B * N, win, win)
        Y = Y.expand(B * N, win, win)
        x0 = torch.floor(X)
        y0 = torch.floor(Y)
        fx = X - x0
        fy = Y - y0
        acc = torch.zeros(B * N, win, win, dtype=lvl.dtype)
        wsum = torch.zeros(B * N, win, win, dtype=lvl.dtype)
        for dy, wy in ((0, 1 - fy), (1, fy)):
            for dx, wx in ((0, 1 - fx), (1, fx)):
                xi = (x0 + dx).long()
                yi = (y0 + dy).long()
                ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
                idx = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1)).reshape(B * N, win * win)
                v = torch.gather(img, 1, idx).reshape(B * N, win, win)
                wgt = wy * wx * ok.to(lvl.dtype)
                acc = acc + wgt * v
                wsum = wsum + wgt
        if mu is not None:
            m = mu.repeat_interleave(N)[:, None, None]
            s = rstd.repeat_interleave(N)[:, None, None]
            acc = (acc - m * wsum) * s
        out.append(acc.reshape(B, H8, W8, win * win))
    return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous()


def corr_lookup2(pyrs: List[List[Tensor]], coords: Tensor, radius: int, mus, rstds) -> Tensor:
    """Lookup on the two-way volume (corr.py:164-171: the two correlations are concatenated on the channel axis of every
    pyramid level, so each level contributes [volume 0 window | volume 1 window]).  -> [B, L*2*(2r+1)^2, H8, W8]"""
    B, _, H8, W8 = coords.shape
    win2 = (2 * radius + 1) ** 2
    parts = [corr_lookup(p, coords, radius, m, s).reshape(B, len(p), win2, H8, W8) for p, m, s in zip(pyrs, mus, rstds)]
    return torch.stack(parts, dim=2).reshape(B, -1, H8, W8)


# ------------------------------------------------------------------------------------------------
# ExpandedFeatTrans (setrans.py:364-410) and the two self-attentions built on it
# ------------------------------------------------------------------------------------------------
def expanded_feat_trans(x: Tensor, P: Tensor, Wv: Tensor, w_agg: Tensor, skip_coeff: Tensor) -> Tensor:
    """x [B,N,C] (the skip input), P [B,M,N,N] -> [B,N,C]
    V = x Wv^T split into M modes of C channels (:373-378); O_m = P_m V_m (:384);
    a_m = softmax_m(<O_m, w_agg> + b) (:395-396, b cancels); y = LN(c_skip*x + sum_m a_m O_m) (:405-407)."""
    B, N, C = x.shape
    M = P.shape[1]
    V = F.linear(x, Wv).reshape(B, N, M, C).permute(0, 2, 1, 3)          # [B,M,N,C]
    O = torch.matmul(P, V)                                                # [B,M,N,C]
    a = torch.softmax((O * w_agg.reshape(1, 1, 1, C)).sum(-1), dim=1)     # [B,M,N]
    y = (O * a[..., None]).sum(dim=1)
    return layernorm_lastdim(skip_coeff.reshape(()) * x + y)


def self_attn_probs(x_tokens: Tensor, Wq: Tensor, Wk: Tensor, biases: Tensor, pos_w: float, M: int,
                    H8: int, W8: int, mask_radius: int = -1, drop_site: Optional[str] = None) -> Tensor:
    """CrossAttFeatTrans with key_feat=query_feat up to the softmax (setrans.py:507-557). [B,M,N,N]"""
    S = mm_scores(x_tokens, x_tokens, Wq, None, Wk, None, M)
    S = clamp_rule(S)
    if biases is not None:
        S = S + pos_w * pos_bias_matrix(biases, H8, W8)
    m = chebyshev_mask(H8, W8, mask_radius)
    if m is not None:
        S = S + m
    return _drop(torch.softmax(S, dim=-1), drop_site)            # (training: self.att_dropout(attention_probs), setrans.py:553-557)


def f2_transform(fmap2: Tensor, sd: Dict[str, Tensor], cfg: OracleConfig, prefix: str = "f2_trans") -> Tensor:
    """A1: SelfAttVisPosTrans 'F2 transformer' (network.py:185-187; setrans.py:578-619).  NCHW->NCHW"""
    B, C, H8, W8 = fmap2.shape
    x = vispos_tokens(fmap2, sd, prefix, cfg.f2_pos_code_weight, drop_site=f"{prefix}.hidden")
    P = self_attn_probs(x, sd[f"{prefix}.setrans.query.weight"], sd[f"{prefix}.setrans.key.weight"],
                        sd.get(f"{prefix}.vispos_encoder.pos_coder.biases"), cfg.f2_pos_code_weight,
                        cfg.f2_num_modes, H8, W8, cfg.f2_attn_mask_radius, drop_site=f"{prefix}.attn")
    y = expanded_feat_trans(x, P,
                            sd[f"{prefix}.setrans.out_trans.first_linear.weight"],
                            sd[f"{prefix}.setrans.out_trans.feat_softaggr.feat2score.weight"],
                            sd[f"{prefix}.setrans.out_trans.input_skip_coeff"])
    return y.transpose(1, 2).reshape(B, C, H8, W8)


def intra_attention(inp_feat: Tensor, sd: Dict[str, Tensor], cfg: OracleConfig) -> Tensor:
    """A3: 'Intra-frame attention' probabilities (network.py:214).  [B,4,N,N]"""
    B, C, H8, W8 = inp_feat.shape
    x = vispos_tokens(inp_feat, sd, "att", cfg.intra_pos_code_weight, drop_site="att.hidden")
    return self_attn_probs(x, sd["att.setrans.query.weight"], sd["att.setrans.key.weight"],
                           sd.get("att.vispos_encoder.pos_coder.biases"), cfg.intra_pos_code_weight,
                           cfg.intra_num_modes, H8, W8, -1, drop_site="att.attn")


def gma_attention(inp_feat: Tensor, sd: Dict[str, Tensor], heads: int, position_only: bool = False,
                  position_and_content: bool = False) -> Tensor:
    """gma.Attention.forward (gma.py:78-100): softmax_j of the content scores scale*q.k, of the relative-position scores
    (RelPosEmb, gma.py:21-50: q(x,y).E_h[u-x] + q(x,y).E_w[v-y] with the scaled q), or of their sum.  [B,heads,N,N]"""
    B, C, H8, W8 = inp_feat.shape
    qk = F.conv2d(inp_feat, sd["att.to_qk.weight"])
    q, k = qk.chunk(2, dim=1)
    dh = q.shape[1] // heads
    q = q.reshape(B, heads, dh, H8 * W8).transpose(2, 3) * (dh ** -0.5)
    k = k.reshape(B, heads, dh, H8 * W8).transpose(2, 3)
    sim = torch.matmul(q, k.transpose(-1, -2))
    if position_only or position_and_content:
        Eh, Ew = sd["att.pos_emb.rel_height.weight"], sd["att.pos_emb.rel_width.weight"]
        P = (Eh.shape[0] + 1) // 2                                        # max_pos_size; rel_ind[i, j] = j - i + P - 1
        ih = torch.arange(H8)[None, :] - torch.arange(H8)[:, None] + P - 1     # [x, u]
        iw = torch.arange(W8)[None, :] - torch.arange(W8)[:, None] + P - 1     # [y, v]
        q5 = q.reshape(B, heads, H8, W8, dh)
        hs = torch.einsum("bhxyd,xud->bhxyu", q5, Eh[ih])                 # [B,h,H8,W8,H8]
        ws = torch.einsum("bhxyd,yvd->bhxyv", q5, Ew[iw])                 # [B,h,H8,W8,W8]
        pos = (hs[..., :, None] + ws[..., None, :]).reshape(B, heads, H8 * W8, H8 * W8)
        sim = pos if position_only else sim + pos
    return torch.softmax(sim, dim=-1)


def gma_aggregate(attn: Tensor, mf: Tensor, sd: Dict[str, Tensor], heads: int) -> Tensor:
    """gma.Aggregate.forward (gma.py:128-140): fmap + gamma * (attn . to_v(fmap)); heads=1 -> no project."""
    B, C, H8, W8 = mf.shape
    v = F.conv2d(mf, sd["update_block.aggregator.to_v.weight"])
    dh = v.shape[1] // heads
    v = v.reshape(B, heads, dh, H8 * W8).transpose(2, 3)
    o = torch.matmul(attn, v).transpose(2, 3).reshape(B, heads * dh, H8, W8)
    if "update_block.aggregator.project.weight" in sd:
        o = F.conv2d(o, sd["update_block.aggregator.project.weight"])
    return mf + sd["update_block.aggregator.gamma"].reshape(()) * o


# ------------------------------------------------------------------------------------------------
# A4. update block (update.py:67-162)
# ------------------------------------------------------------------------------------------------
def _conv(x, sd, name, padding):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=padding)


def motion_encoder(flow: Tensor, corr: Tensor, sd, p="update_block.encoder") -> Tensor:
    """BasicMotionEncoder.forward (update.py:79-87) -> [B,128,H8,W8] (126 conv channels + flow)."""
    cor = F.relu(_conv(corr, sd, f"{p}.convc1", 0))
    cor = F.relu(_conv(cor, sd, f"{p}.convc2", 1))
    flo = F.relu(_conv(flow, sd, f"{p}.convf1", 3))
    flo = F.relu(_conv(flo, sd, f"{p}.convf2", 1))
    out = F.relu(_conv(torch.cat([cor, flo], dim=1), sd, f"{p}.conv", 1))
    return torch.cat([out, flow], dim=1)


def sepconv_gru(h: Tensor, x: Tensor, sd, p="update_block.gru") -> Tensor:
    """SepConvGRU.forward (update.py:49-64): 1x5 pass then 5x1 pass."""
    for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], dim=1)
        z = torch.sigmoid(_conv(hx, sd, f"{p}.convz{sfx}", pad))
        r = torch.sigmoid(_conv(hx, sd, f"{p}.convr{sfx}", pad))
        q = torch.tanh(_conv(torch.cat([r * h, x], dim=1), sd, f"{p}.convq{sfx}", pad))
        h = (1 - z) * h + z * q
    return h


def flow_head(h: Tensor, sd, p="update_block.flow_head") -> Tensor:
    """FlowHead.forward (update.py:15-16)."""
    return _conv(F.relu(_conv(h, sd, f"{p}.conv1", 1)), sd, f"{p}.conv2", 1)


def mask_head(h: Tensor, sd, p="update_block.mask") -> Tensor:
    """mask Sequential + 0.25 scale (update.py:124-127, :161)."""
    return 0.25 * _conv(F.relu(_conv(h, sd, f"{p}.0", 1)), sd, f"{p}.2", 0)


def aggregator(mf: Tensor, attention: Tensor, sd, p="update_block.aggregator") -> Tensor:
    """GMAUpdateBlock setrans branch (update.py:143-149): ExpandedFeatTrans on raw motion features."""
    B, C, H8, W8 = mf.shape
    t = mf.reshape(B, C, H8 * W8).transpose(1, 2)
    y = expanded_feat_trans(t, attention, sd[f"{p}.first_linear.weight"],
                            sd[f"{p}.feat_softaggr.feat2score.weight"], sd[f"{p}.input_skip_coeff"])
    return y.reshape(B, H8, W8, C).permute(0, 3, 1, 2)


def update_block(net, inp, corr, flow, attention, sd, cfg: OracleConfig):
    """GMAUpdateBlock.forward (update.py:137-162) -> (net, mask, delta_flow)."""
    mf = motion_encoder(flow, corr, sd)
    if cfg.use_setrans:
        mfg = aggregator(mf, attention, sd)
    else:
        mfg = gma_aggregate(attention, mf, sd, cfg.num_heads)
    net = sepconv_gru(net, torch.cat([inp, mf, mfg], dim=1), sd)
    return net, mask_head(net, sd), flow_head(net, sd)


def convex_upsample(flow: Tensor, mask: Tensor) -> Tensor:
    """CRAFT.upsample_flow (network.py:151-162), written as an explicit 9-tap sum.
    up[n,c,8y+i,8x+j] = sum_k softmax_k(mask[n,k*64+i*8+j,y,x]) * 8*flow_zero_pad[n,c,y+k//3-1,x+k%3-1]"""
    B, _, H8, W8 = flow.shape
    m = torch.softmax(mask.reshape(B, 9, 8, 8, H8, W8), dim=1)
    fp = F.pad(8.0 * flow, (1, 1, 1, 1))
    up = torch.zeros(B, 2, 8, 8, H8, W8, dtype=flow.dtype)
    for k in range(9):
        ky, kx = divmod(k, 3)
        nb = fp[:, :, ky:ky + H8, kx:kx + W8]                      # [B,2,H8,W8]
        up = up + m[:, k][:, None] * nb[:, :, None, None]
    return up.permute(0, 1, 4, 2, 5, 3).reshape(B, 2, 8 * H8, 8 * W8)


# ------------------------------------------------------------------------------------------------
# CNN encoders (extractor.py:124-196) — not hot path, needed for an end-to-end forward
# ------------------------------------------------------------------------------------------------
def _norm(x, sd, name, kind, bn_stats=None):
    """norm_fn of BasicEncoder / ResidualBlock (extractor.py:21-47, :131-137).  ``bn_stats`` (a dict) selects nn.BatchNorm2d's
    TRAINING behaviour: normalise with the batch statistics and record the momentum-0.1 update of the running statistics
    (unbiased variance) under the buffer names -- what model.train() does to cnet unless freeze_bn() was called."""
    if kind == "instance":
        return F.instance_norm(x, eps=1e-5)
    if bn_stats is not None:
        rm, rv = sd[name + ".running_mean"].detach().clone(), sd[name + ".running_var"].detach().clone()
        y = F.batch_norm(x, rm, rv, sd[name + ".weight"], sd[name + ".bias"], training=True, momentum=0.1, eps=1e-5)
        bn_stats[name + ".running_mean"], bn_stats[name + ".running_var"] = rm, rv
        return y
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], training=False, eps=1e-5)


def _resblock(x, sd, p, kind, stride, bn_stats=None):
    """ResidualBlock.forward (extractor.py:56-64)."""
    y = F.relu(_norm(F.conv2d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], stride=stride, padding=1), sd, p + ".norm1", kind, bn_stats))
    y = F.relu(_norm(F.conv2d(y, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1), sd, p + ".norm2", kind, bn_stats))
    if stride != 1:
        x = _norm(F.conv2d(x, sd[p + ".downsample.0.weight"], sd[p + ".downsample.0.bias"], stride=stride), sd, p + ".norm3", kind, bn_stats)
    return F.relu(x + y)


def basic_encoder(x: Tensor, sd, p: str, kind: str, bn_stats=None) -> Tensor:
    """BasicEncoder.forward (extractor.py:173-196); eval mode unless ``bn_stats`` is given (see _norm)."""
    x = F.relu(_norm(F.conv2d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], stride=2, padding=3), sd, p + ".norm1", kind, bn_stats))
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        x = _resblock(x, sd, f"{p}.layer{li}.0", kind, stride, bn_stats)
        x = _resblock(x, sd, f"{p}.layer{li}.1", kind, 1, bn_stats)
    return F.conv2d(x, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"])


def coords_grid(B: int, H8: int, W8: int, dtype=torch.float32) -> Tensor:
    """utils.py:82-85: channel 0 = x (column index), channel 1 = y (row index)."""
    ys, xs = torch.meshgrid(torch.arange(H8, dtype=dtype), torch.arange(W8, dtype=dtype), indexing="ij")
    return torch.stack([xs, ys], dim=0)[None].expand(B, -1, -1, -1).clone()


# ------------------------------------------------------------------------------------------------
# CRAFT.forward (network.py:164-267), eval mode
# ------------------------------------------------------------------------------------------------
def hot_path(fmap1: Tensor, fmap2: Tensor, net: Tensor, inp: Tensor, sd, cfg: OracleConfig, iters: int,
             flow_init: Optional[Tensor] = None, test_mode: int = 1, capture: Optional[dict] = None,
             cached_positions: Optional[Tensor] = None, code_is_cached: bool = False):
    """Everything in CRAFT.forward after the CNN encoders: fmap1/fmap2 are fnet outputs, net/inp the
    tanh/relu halves of cnet's output (network.py:185-267).  code_is_cached / cached_positions: the state of the inter-frame encoder's
    eval-mode positional-code cache left by an earlier call of the same shape (cached_positions None = the grid); 'lsinu' only."""
    B, C, H8, W8 = fmap1.shape
    fmap2t = f2_transform(fmap2, sd, cfg) if cfg.f2trans != "none" else fmap2
    two_way = cfg.craft and cfg.f1trans != "none"
    # --f1 shared|private (network.py:94-103, :180-183): frame 1 goes through its own (or the shared) transformer and
    # the correlation becomes two-way: (transformed 1, conv 2) and (conv 1, transformed 2), concatenated (corr.py:164-171)
    fmap1t = f2_transform(fmap1, sd, cfg, prefix="f1_trans") if two_way else None
    if cfg.use_setrans:
        attention = intra_attention(inp, sd, cfg)
    else:
        attention = gma_attention(inp, sd, cfg.num_heads, cfg.position_only, cfg.position_and_content)
    pos1 = None
    if flow_init is not None:          # corr.py:153: frame 1 is encoded at coords1 = grid + flow_init, flipped to (y, x)
        pos1 = (coords_grid(B, H8, W8, fmap1.dtype) + flow_init).permute(0, 2, 3, 1).flip(-1).reshape(B, H8 * W8, 2)
    # EVAL-mode quirk of the reference (setrans.py:744-758, pos_code_lookup_cache): the positional code computed for frame 1 is cached
    # by SHAPE and returned for frame 2 of the same call, so with a flow_init the second frame is encoded at coords1 as well (training
    # mode recomputes per call and has no flow_init).  The cache also survives across calls of the same shape: a warm-started sequence
    # (evaluate.py's Sintel submission) keeps the code of its FIRST call, which had no flow_init -> the grid.
    if code_is_cached:
        pos1 = cached_positions
    pos2 = pos1
    if two_way:
        c = [inter_corr_raw(fmap1t, fmap2, sd, cfg, pos1, pos2), inter_corr_raw(fmap1, fmap2t, sd, cfg, pos1, pos2)]
        stats = [global_stats(ci) for ci in c]
        mu, rstd = [st[0] for st in stats], [st[1] for st in stats]
        pyr = [build_pyramid(ci, H8, W8, cfg.corr_levels) for ci in c]
    elif cfg.craft:
        c = inter_corr_raw(fmap1, fmap2t, sd, cfg, pos1, pos2)
        mu, rstd = global_stats(c)
        pyr = build_pyramid(c, H8, W8, cfg.corr_levels)
    else:
        c = plain_corr_raw(fmap1, fmap2t)
        mu, rstd = None, None
        pyr = build_pyramid(c, H8, W8, cfg.corr_levels)
    coords0 = coords_grid(B, H8, W8, fmap1.dtype)
    coords1 = coords0.clone()
    if flow_init is not None:
        coords1 = coords1 + flow_init
    if capture is not None:
        capture.update(fmap2t=fmap2t, fmap1t=fmap1t, attention=attention, corr_raw=c, mu=mu, rstd=rstd)
    preds = []
    for it in range(iters):
        coords1 = coords1.detach()                                     # network.py:232 (a no-op without autograd)
        corr = corr_lookup2(pyr, coords1, cfg.corr_radius, mu, rstd) if two_way else corr_lookup(pyr, coords1, cfg.corr_radius, mu, rstd)
        flow = coords1 - coords0
        net, mask, dflow = update_block(net, inp, corr, flow, attention, sd, cfg)
        coords1 = coords1 + dflow
        preds.append(convex_upsample(coords1 - coords0, mask))
        if capture is not None and it == 0:
            capture.update(corr0=corr, net1=net, mask1=mask, dflow1=dflow)
    if test_mode == 1:
        return coords1 - coords0, preds[-1]
    if test_mode == 2:
        return coords1 - coords0, preds
    return preds


def craft_forward(sd, cfg: OracleConfig, image1: Tensor, image2: Tensor, iters: int = 12,
                  flow_init: Optional[Tensor] = None, test_mode: int = 1, capture: Optional[dict] = None, code_is_cached: bool = False):
    """CRAFT.forward (network.py:164-267): images float32 [B,3,H,W] in 0..255.  code_is_cached: an earlier call of this shape without a
    flow_init filled the inter-frame encoder's positional-code cache (hot_path)."""
    with torch.no_grad():
        im1 = 2 * (image1 / 255.0) - 1.0
        im2 = 2 * (image2 / 255.0) - 1.0
        B = im1.shape[0]
        fm = basic_encoder(torch.cat([im1, im2], dim=0), sd, "fnet", "instance")
        fmap1, fmap2 = fm[:B], fm[B:]
        cn = basic_encoder(im1, sd, "cnet", "batch")
        net = torch.tanh(cn[:, :128])
        inp = torch.relu(cn[:, 128:])
        if capture is not None:
            capture.update(fmap1=fmap1, fmap2=fmap2, net0=net, inp=inp)
        return hot_path(fmap1, fmap2, net, inp, sd, cfg, iters, flow_init, test_mode, capture, None, code_is_cached)


# ------------------------------------------------------------------------------------------------
# training step (train.py:44-73, :228-236): the same forward with autograd on, model.train() semantics
# ------------------------------------------------------------------------------------------------
MAX_FLOW = 400.0         # train.py:30


def sequence_loss(flow_preds: List[Tensor], flow_gt: Tensor, valid: Tensor, gamma: float):
    """train.py:44-73: sum_i gamma^(T-1-i) * mean(valid * |pred_i - gt|) over ALL B*2*H*W elements, valid = (valid >= 0.5) &
    (|gt| < MAX_FLOW); metrics of the last prediction over the valid pixels."""
    T = len(flow_preds)
    v = (valid >= 0.5) & ((flow_gt ** 2).sum(dim=1).sqrt() < MAX_FLOW)
    loss = 0.0
    for i, p in enumerate(flow_preds):
        loss = loss + gamma ** (T - i - 1) * (v[:, None] * (p - flow_gt).abs()).mean()
    epe = ((flow_preds[-1] - flow_gt) ** 2).sum(dim=1).sqrt().reshape(-1)[v.reshape(-1)]
    return loss, {"epe": epe.mean().item(), "1px": (epe < 1).float().mean().item(), "3px": (epe < 3).float().mean().item(),
                  "5px": (epe < 5).float().mean().item()}


def craft_train_forward(sd, cfg: OracleConfig, image1: Tensor, image2: Tensor, iters: int = 12, freeze_bn: bool = False):
    """CRAFT.forward under model.train() (network.py:164-267), every dropout at p = 0 unless DROPOUT_MASKS hands the masks in (see the top of
    this file): differentiable w.r.t. the tensors
    of ``sd`` that require grad; cnet's BatchNorm uses batch statistics unless ``freeze_bn`` (network.py:136-140).
    -> (flow_predictions, {buffer name: updated running statistic})."""
    im1 = 2 * (image1 / 255.0) - 1.0
    im2 = 2 * (image2 / 255.0) - 1.0
    B = im1.shape[0]
    fm = basic_encoder(torch.cat([im1, im2], dim=0), sd, "fnet", "instance")
    bn_stats = None if freeze_bn else {}
    cn = basic_encoder(im1, sd, "cnet", "batch", bn_stats)
    preds = hot_path(fm[:B], fm[B:], torch.tanh(cn[:, :128]), torch.relu(cn[:, 128:]), sd, cfg, iters, None, test_mode=0)
    return preds, (bn_stats or {})


# ------------------------------------------------------------------------------------------------
# warm start
# ------------------------------------------------------------------------------------------------
def forward_interpolate(flow: Tensor) -> Tensor:
    """core/utils/utils.py:34-62 restated without scipy: every pixel (x0, y0) of the grid takes the flow of the NEAREST
    forward-warped source point (x0' + dx, y0' + dy) among those that land strictly inside (0, W) x (0, H); distances in
    float64 like griddata('nearest') on the float64 coordinates the reference builds (int64 grid + float32 flow).  Exact
    ties (measure zero for real flows) go to the lowest source index here; a k-d tree may pick another of the tied points.
    flow [2, H, W] -> [2, H, W] float32; no valid source point at all: zeros (the reference's fill value)."""
    import numpy as np
    f = flow.detach().cpu().numpy().astype(np.float32)
    dx, dy = f[0], f[1]
    H, W = dx.shape
    x0, y0 = np.meshgrid(np.arange(W), np.arange(H))
    x1 = (x0 + dx.astype(np.float64)).reshape(-1)
    y1 = (y0 + dy.astype(np.float64)).reshape(-1)
    valid = (x1 > 0) & (x1 < W) & (y1 > 0) & (y1 < H)
    if not valid.any():
        return torch.zeros(2, H, W)
    xs, ys, vx, vy = x1[valid], y1[valid], dx.reshape(-1)[valid], dy.reshape(-1)[valid]
    out = np.zeros((2, H * W), dtype=np.float32)
    gx, gy = x0.reshape(-1).astype(np.float64), y0.reshape(-1).astype(np.float64)
    for s in range(0, H * W, 2048):                                   # blocks of targets: [2048, n_valid] distances
        d2 = (gx[s:s + 2048, None] - xs[None]) ** 2 + (gy[s:s + 2048, None] - ys[None]) ** 2
        j = np.argmin(d2, axis=1)
        out[0, s:s + 2048], out[1, s:s + 2048] = vx[j], vy[j]
    return torch.from_numpy(out.reshape(2, H, W))
