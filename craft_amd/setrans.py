"""Squeeze-Expansion transformer blocks of CRAFT on HIP kernels.

Same class / parameter names as the reference's ``core/setrans.py`` (so ``state_dict`` keys match
key-for-key and reference checkpoints load), but the forward passes enqueue the gfx950 kernels of
``libcraft_hip.so`` on channels-last "tokens" tensors.  Only the canonical options are implemented
(pos_code_type='bias', pool_modes_feat='softmax', has_FFN=False); anything else raises.

Reference map:  SETransConfig setrans.py:71-157 · SlidingPosBiases2D :644-708 ·
SETransInputFeatEncoder :710-800 · LearnedSoftAggregate :279-300 · ExpandedFeatTrans :304-410 ·
CrossAttFeatTrans :412-566 · SelfAttVisPosTrans :568-619.
"""
from __future__ import annotations

import copy
import math
import os
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .hip import PREC_F32


class SETransConfig:
    """Fields of the reference config that reach the canonical model (setrans.py:71-157)."""

    def __init__(self):
        self.feat_dim = -1
        self.in_feat_dim = -1
        self.pos_dim = 2
        self.pos_code_weight = 1.0
        self.num_modes = 4
        self.tie_qk_scheme = "shared"
        self.attn_clip = 100
        self.attn_diag_cycles = 1000
        self.base_initializer_range = 0.02
        self.qk_have_bias = False
        self.v_has_bias = False
        self.query_idbias_scale = 10
        self.feattrans_lin1_idbias_scale = 10
        self.pool_modes_feat = "softmax"
        self.hidden_dropout_prob = 0.1
        self.attention_probs_dropout_prob = 0.2
        self.drop_path_prob = 0
        self.pos_code_type = "bias"
        self.ablate_multihead = False
        self.out_attn_probs_only = False
        self.out_attn_scores_only = False
        self.attn_mask_radius = -1
        self.pos_bias_radius = 7
        self.has_FFN = False
        self.has_input_skip = False

    def try_assign(self, args, *keys) -> bool:
        ok = False
        src = args if isinstance(args, dict) else vars(args)
        for k in keys:
            if k in src:
                setattr(self, k, src[k])
                ok = True
        return ok

    def update_config(self, args):
        """Copy same-named attributes out of the argparse Namespace (setrans.py:139-157)."""
        self.try_assign(args, "num_modes", "base_initializer_range", "pos_code_type", "ablate_multihead", "attn_clip",
                        "attn_diag_cycles", "tie_qk_scheme", "feattrans_lin1_idbias_scale", "qk_have_bias", "v_has_bias",
                        "out_attn_probs_only", "out_attn_scores_only", "in_feat_dim", "pos_bias_radius")
        if not self.try_assign(args, "out_feat_dim"):
            self.feat_dim = self.in_feat_dim
        else:
            self.feat_dim = self.out_feat_dim


def _prec_of(module) -> int:
    return getattr(module, "hip_prec", PREC_F32)


class SlidingPosBiases2D(nn.Module):
    """Learnable [2R+1, 2R+1] table; pb(i,j) = biases[dh+R, dw+R] inside the window, else 0.
    The [N, N] expansion of the reference (setrans.py:690-708) is never materialised: the score kernels
    evaluate it from (i, j)."""

    def __init__(self, pos_dim: int = 2, pos_bias_radius: int = 7):
        super().__init__()
        if pos_dim != 2:
            raise NotImplementedError("only 2-D positional biases")
        self.R = pos_bias_radius
        self.biases = nn.Parameter(torch.zeros(2 * pos_bias_radius + 1, 2 * pos_bias_radius + 1))


class LearnedSinuPosEmbedder(nn.Module):
    """Learnable sinusoidal positional embedding (setrans.py:624-646, `--interpos / --intrapos lsinu`):
    E(p) = LayerNorm_C( interlace( sin(fc(p)[0::2]), cos(fc(p)[1::2]) ) ), fc = Linear(2, C), eps 1e-12, no affine.  A [N, C] table per
    image size -- a handful of small torch ops (differentiable: the training path gets pos_fc's gradients from torch autograd), NOT a
    hand-written kernel: the non-canonical positional code is supported for completeness, not tuned."""

    def __init__(self, pos_dim: int, pos_embed_dim: int, omega: float = 1.0):
        super().__init__()
        self.pos_dim, self.pos_embed_dim, self.omega = pos_dim, pos_embed_dim, omega
        self.pos_fc = nn.Linear(pos_dim, pos_embed_dim, bias=True)

    def forward(self, pos_normed: torch.Tensor) -> torch.Tensor:
        e0 = torch.nn.functional.linear(pos_normed, self.pos_fc.weight, self.pos_fc.bias)
        mix = torch.stack((torch.sin(self.omega * e0[..., 0::2]), torch.cos(self.omega * e0[..., 1::2])), dim=-1).reshape(e0.shape)
        return torch.nn.functional.layer_norm(mix, (self.pos_embed_dim,), None, None, 1e-12)


class SETransInputFeatEncoder(nn.Module):
    """Visual tokens + positional code -> LayerNorm-ed tokens (setrans.py:710-800).  `bias` (the released configuration): no embedding,
    the [2R+1, 2R+1] table goes into the score kernels.  `lsinu`: tokens + pos_code_weight * E(position / max position) before the
    LayerNorm, no score bias."""

    def __init__(self, config: SETransConfig):
        super().__init__()
        self.feat_dim = config.in_feat_dim
        self.pos_code_type = config.pos_code_type
        if config.pos_code_type == "bias":
            self.pos_code_weight = 0.0
            self.pos_coder = SlidingPosBiases2D(config.pos_dim, config.pos_bias_radius)
        elif config.pos_code_type == "lsinu":
            self.pos_code_weight = float(config.pos_code_weight)
            self.pos_coder = LearnedSinuPosEmbedder(config.pos_dim, self.feat_dim, omega=1.0)
        else:
            raise NotImplementedError("HIP path implements pos_code_type 'bias' (the released configuration) and 'lsinu'")
        self._code, self._code_key = None, None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.drop_code_cache())

    @property
    def pos_table(self):
        """The score kernels' positional-bias table, or None (lsinu: setrans.py:781 pos_biases = None)."""
        return self.pos_coder.biases if self.pos_code_type == "bias" else None

    def embedding(self, hw, device, positions: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """pos_code_weight * E for the pixel grid of an H8 x W8 image ([N, C]) or for explicit (y, x) positions [B, N, 2] (the first
        frame's coordinates after a flow_init, corr.py:153); None for `bias`.  Positions are divided by their maximum (setrans.py:772)."""
        if self.pos_code_type == "bias":
            return None
        if positions is None:
            H8, W8 = hw
            ys, xs = torch.meshgrid(torch.arange(H8, device=device), torch.arange(W8, device=device), indexing="ij")
            positions = torch.stack([ys, xs], dim=-1).reshape(H8 * W8, 2).float()
        return self.pos_code_weight * self.pos_coder(positions / positions.max())

    def ln_tokens(self, tok: torch.Tensor, hw, positions: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Raw tokens [B, N, C] -> LayerNorm-ed tokens (inference: no graph)."""
        if self.pos_code_type == "bias":
            return ops.tokens_norm(tok)
        # the reference's eval-mode cache (setrans.py:744-758, pos_code_lookup_cache): the code is computed on the first call of a
        # [B, N] shape and returned for every later call of that shape -- frame 2 of the same pair, and the following pairs of a
        # warm-started sequence, get the FIRST call's code whatever their positions are.  Kept, because it decides the numbers the
        # reference's evaluation produces; dropped on train() and on load_state_dict (the reference keeps a stale code there).
        key = (tok.shape[0], tok.shape[1], tok.device)
        if self._code_key != key:
            with torch.no_grad():
                self._code, self._code_key = self.embedding(hw, tok.device, positions), key
        return ops.tokens_norm(tok + self._code)

    def drop_code_cache(self):
        self._code, self._code_key = None, None

    def train(self, mode: bool = True):
        self.drop_code_cache()
        return super().train(mode)

    def forward(self, vis_feat: torch.Tensor) -> torch.Tensor:
        """NCHW -> LayerNorm-ed tokens [B, N, C] (setrans.py:791-795)."""
        if self.pos_code_type == "bias":
            return ops.tokens_from_nchw(vis_feat, ln=True)
        return self.ln_tokens(ops.tokens_from_nchw(vis_feat), vis_feat.shape[-2:])


class LearnedSoftAggregate(nn.Module):
    def __init__(self, num_feat: int, group_dim: int, keepdim: bool = False):
        super().__init__()
        self.num_feat, self.group_dim, self.keepdim = num_feat, group_dim, keepdim
        self.feat2score = nn.Linear(num_feat, 1)


class ExpandedFeatTrans(nn.Module):
    """V projection into M modes, O_m = P_m V_m, softmax-over-modes pooling, skip + LayerNorm."""

    def __init__(self, config: SETransConfig, name: str):
        super().__init__()
        if config.has_FFN or config.pool_modes_feat != "softmax" or not config.has_input_skip or config.v_has_bias:
            raise NotImplementedError("HIP path implements has_FFN=False, softmax pooling, input skip, no V bias")
        self.name = name
        self.config = config
        self.in_feat_dim, self.feat_dim, self.num_modes = config.in_feat_dim, config.feat_dim, config.num_modes
        self.first_linear = nn.Linear(self.in_feat_dim, self.feat_dim * self.num_modes, bias=False)
        self.feat_softaggr = LearnedSoftAggregate(self.feat_dim, group_dim=1)
        self.input_skip_coeff = nn.Parameter(torch.ones(1))

    def add_identity_bias(self):
        s = self.config.feattrans_lin1_idbias_scale
        if s > 0:
            eye = torch.eye(self.feat_dim) * self.config.base_initializer_range * s
            with torch.no_grad():
                blk = self.first_linear.weight[: self.feat_dim, : self.feat_dim]
                blk.copy_(blk * 0.5 + eye)

    def forward(self, input_feat: torch.Tensor, attention_probs: torch.Tensor, out: Optional[torch.Tensor] = None,
                prec: Optional[int] = None) -> torch.Tensor:
        """input_feat tokens [B, N, C] (also the skip input), attention_probs [B, M, N, ldp] from
        ``ops.attn_probs`` -> tokens [B, N, C]   (setrans.py:364-410)."""
        prec = _prec_of(self) if prec is None else prec
        ldp = ops.vt_stride(attention_probs)
        vT = ops.linear_t(input_feat, self.first_linear.weight, ldp, prec, Dv=self.feat_dim,
                          packed=ops.linear_pack(self, "first_linear", self.first_linear.weight, prec))
        O = ops.attn_apply(attention_probs, vT, self.feat_dim, prec)
        return ops.mode_pool_ln(O, input_feat, self.feat_softaggr.feat2score.weight, self.input_skip_coeff, out=out)


    def forward_flash(self, input_feat: torch.Tensor, q: torch.Tensor, k: torch.Tensor, hw, scale: float, pos_biases, pos_w: float,
                      mask_radius: int, clamp_ord, prec: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The same layer with the attention fused in (``ops.flash_attention``): q, k projected tokens [B, N, C]."""
        H8, W8 = hw
        ldt = ops.round_up(H8 * W8, 32)
        vT = ops.linear_t(input_feat, self.first_linear.weight, ldt, prec, Dv=self.feat_dim, acc_order=True,
                          packed=ops.linear_pack(self, "first_linear", self.first_linear.weight, prec))
        O = ops.flash_attention(q, k, vT, H8, W8, self.num_modes, self.feat_dim, scale, pos_biases, pos_w, mask_radius, clamp_ord, prec)
        return ops.mode_pool_ln(O, input_feat, self.feat_softaggr.feat2score.weight, self.input_skip_coeff, out=out)


class CrossAttFeatTrans(nn.Module):
    def __init__(self, config: SETransConfig, name: str):
        super().__init__()
        self.config, self.name = config, name
        self.num_modes = config.num_modes
        self.in_feat_dim, self.feat_dim = config.in_feat_dim, config.feat_dim
        self.attention_mode_dim = self.in_feat_dim // self.num_modes
        self.query = nn.Linear(self.in_feat_dim, self.in_feat_dim, bias=config.qk_have_bias)
        self.key = nn.Linear(self.in_feat_dim, self.in_feat_dim, bias=config.qk_have_bias)
        self.out_attn_scores_only = config.out_attn_scores_only
        self.out_attn_probs_only = config.out_attn_probs_only
        if config.ablate_multihead:
            raise NotImplementedError("ablate_multihead is an ablation outside the HIP path")
        if self.out_attn_scores_only or self.out_attn_probs_only:
            self.out_trans = None
            if self.num_modes > 1:
                self.attn_softaggr = LearnedSoftAggregate(1, group_dim=1, keepdim=True)
        else:
            self.out_trans = ExpandedFeatTrans(config, name + "-out_trans")
        self.tie_qk_scheme = config.tie_qk_scheme
        self.pos_code_weight = config.pos_code_weight if config.pos_code_type == "bias" else 1      # (setrans.py:452: unused without a bias table)
        self.attn_clip = config.attn_clip
        self._init_weights()

    def _init_weights(self):
        """normal(0, 0.02) linears, zero biases, Q/K tying, identity bias on the first mode
        (setrans.py:167-187, :470-493)."""
        std = self.config.base_initializer_range
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0.0, std)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        if self.tie_qk_scheme == "shared":
            self.key.weight = self.query.weight
            if self.key.bias is not None:
                self.key.bias = self.query.bias
        elif self.tie_qk_scheme == "loose":
            with torch.no_grad():
                self.key.weight.copy_(self.query.weight)
                if self.key.bias is not None:
                    self.key.bias.copy_(self.query.bias)
        d = self.attention_mode_dim
        eye = (torch.eye(d) * std * self.config.query_idbias_scale).repeat(1, self.in_feat_dim // d)
        with torch.no_grad():
            self.key.weight[:d].copy_(self.key.weight[:d] * 0.5 + eye)
        if self.out_trans is not None:
            self.out_trans.add_identity_bias()

    def project(self, query_feat: torch.Tensor, key_feat: Optional[torch.Tensor], prec: int):
        """Q = query(x_q), K = key(x_k) on tokens (setrans.py:507-508)."""
        q = ops.linear(query_feat, self.query.weight, self.query.bias, prec, packed=ops.linear_pack(self, "query", self.query.weight, prec))
        k = ops.linear(query_feat if key_feat is None else key_feat, self.key.weight, self.key.bias, prec,
                       packed=ops.linear_pack(self, "key", self.key.weight, prec))
        return q, k

    def forward(self, query_feat, key_feat=None, pos_biases=None, attention_mask_radius: int = -1, hw=None,
                prec: Optional[int] = None, defer: bool = False):
        """query_feat/key_feat: LayerNorm-ed tokens; pos_biases: the [2R+1,2R+1] table (not the N x N
        expansion); hw = (H8, W8).  Returns attention probabilities [B, M, N, ldp] (out_attn_probs_only) or
        transformed tokens.  The scores-only variant lives in ``corr.TransCorrBlock`` because its output
        is the correlation pyramid."""
        prec = _prec_of(self) if prec is None else prec
        H8, W8 = hw
        if self.out_attn_scores_only:
            raise RuntimeError("use corr.TransCorrBlock for the scores-only (correlation) instance")
        q, k = self.project(query_feat, key_feat, prec)
        scale = 1.0 / math.sqrt(self.attention_mode_dim)
        mx = ops.score_max(q, k, H8, W8, self.num_modes, scale, prec)
        kf = query_feat if key_feat is None else key_feat
        if (self.out_trans is not None and not defer and not os.environ.get("CRAFT_NO_FLASH")
                and ops.flash_supported(H8 * W8, W8, self.attention_mode_dim, self.out_trans.feat_dim, prec)):
            # probabilities used once: fused scores -> online softmax -> P.V, nothing N x N in memory
            return self.out_trans.forward_flash(kf, q, k, hw, scale, pos_biases, float(self.pos_code_weight),
                                                attention_mask_radius, mx, prec)
        # P that is consumed right here (or by a caller that asked for it, `defer`) skips the normalisation pass
        P = ops.attn_probs(q, k, H8, W8, self.num_modes, scale, pos_biases, float(self.pos_code_weight),
                           attention_mask_radius, mx, prec, defer=defer or not self.out_attn_probs_only)
        if self.out_attn_probs_only:
            return P
        return self.out_trans(kf, P, prec=prec)


class SelfAttVisPosTrans(nn.Module):
    def __init__(self, config: SETransConfig, name: str):
        super().__init__()
        self.config = copy.copy(config)
        self.name = name
        self.out_attn_only = config.out_attn_scores_only or config.out_attn_probs_only
        self.attn_mask_radius = config.attn_mask_radius
        self.setrans = CrossAttFeatTrans(self.config, name)
        self.vispos_encoder = SETransInputFeatEncoder(self.config)

    def forward_tokens(self, x_tokens_ln: torch.Tensor, hw, prec: Optional[int] = None, defer: bool = False):
        return self.setrans(x_tokens_ln, pos_biases=self.vispos_encoder.pos_table,
                            attention_mask_radius=self.attn_mask_radius, hw=hw, prec=prec, defer=defer)

    def forward(self, x: torch.Tensor):
        """NCHW in; NCHW out (feature transformer) or [B, M, N, N] probabilities (setrans.py:578-619)."""
        B, C, H8, W8 = x.shape
        xt = self.vispos_encoder(x)
        y = self.forward_tokens(xt, (H8, W8))
        if self.out_attn_only:
            return y[..., : H8 * W8]
        return ops.tokens_to_nchw(y, H8, W8)
