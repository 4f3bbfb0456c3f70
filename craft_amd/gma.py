"""GMA attention / aggregation (the ``use_setrans=False`` variant) on HIP kernels.

Reference: ``core/gma.py`` — ``Attention`` :53-102 (softmax of the content scores scale*q.k, of the relative-position
scores of ``RelPosEmb`` :6-50 -- ``position_only`` -- or of their sum -- ``position_and_content``),
``Aggregate`` :105-142 (fmap + gamma * attn . to_v(fmap)).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .hip import PREC_F32


class RelPosEmb(nn.Module):
    def __init__(self, max_pos_size: int, dim_head: int):
        super().__init__()
        self.rel_height = nn.Embedding(2 * max_pos_size - 1, dim_head)
        self.rel_width = nn.Embedding(2 * max_pos_size - 1, dim_head)
        deltas = torch.arange(max_pos_size).view(1, -1) - torch.arange(max_pos_size).view(-1, 1)
        self.register_buffer("rel_ind", deltas + max_pos_size - 1)


class Attention(nn.Module):
    def __init__(self, *, args, dim: int, max_pos_size: int = 100, heads: int = 4, dim_head: int = 128):
        super().__init__()
        self.args, self.heads, self.dim_head = args, heads, dim_head
        self.max_pos_size = max_pos_size
        self.pos_embed_weight = 1.0
        self.scale = dim_head ** -0.5
        self.to_qk = nn.Conv2d(dim, heads * dim_head * 2, 1, bias=False)
        self.pos_emb = RelPosEmb(max_pos_size, dim_head)

    def forward_tokens(self, x: torch.Tensor, hw, prec: int, defer: bool = False) -> torch.Tensor:
        """x: tokens [B, N, dim] (not normalised) -> P [B, heads, N, ldp]."""
        H8, W8 = hw
        inner = self.heads * self.dim_head
        w = self.to_qk.weight.view(2 * inner, -1)
        q = ops.linear(x, w[:inner], None, prec)
        k = ops.linear(x, w[inner:], None, prec)
        pos_only = bool(getattr(self.args, "position_only", False))
        relpos = None
        if pos_only or getattr(self.args, "position_and_content", False):
            # RelPosEmb (gma.py:21-50): (scale*q)(x,y).E_h[u - x] + (scale*q)(x,y).E_w[v - y]: per query a row of 2*H8-1 and
            # one of 2*W8-1 scores -- two small GEMMs against the embedding rows of the offsets that can occur
            if max(H8, W8) > self.max_pos_size:
                raise ValueError(f"feature map {H8}x{W8} exceeds RelPosEmb max_pos_size {self.max_pos_size}")
            B, N, _ = x.shape
            P0 = self.max_pos_size - 1
            Eh = self.pos_emb.rel_height.weight[P0 - (H8 - 1): P0 + H8]
            Ew = self.pos_emb.rel_width.weight[P0 - (W8 - 1): P0 + W8]
            qh = q.view(B, N, self.heads, self.dim_head).permute(0, 2, 1, 3).reshape(B * self.heads, N, self.dim_head)
            Hs = ops.linear(qh, Eh, None, prec).view(B, self.heads, N, 2 * H8 - 1)
            Ws = ops.linear(qh, Ew, None, prec).view(B, self.heads, N, 2 * W8 - 1)
            relpos = (Hs, Ws, self.scale * (1.0 if pos_only else self.pos_embed_weight))
        return ops.attn_probs(q, k, H8, W8, self.heads, 0.0 if pos_only else self.scale, None, 0.0, -1, None, prec, defer=defer,
                              relpos=relpos)

    def forward(self, fmap: torch.Tensor) -> torch.Tensor:
        B, C, H8, W8 = fmap.shape
        P = self.forward_tokens(ops.tokens_from_nchw(fmap), (H8, W8), getattr(self, "hip_prec", PREC_F32))
        return P[..., : H8 * W8]


class Aggregate(nn.Module):
    def __init__(self, args, dim: int, heads: int = 4, dim_head: int = 128):
        super().__init__()
        self.args, self.heads, self.dim_head = args, heads, dim_head
        inner = heads * dim_head
        self.to_v = nn.Conv2d(dim, inner, 1, bias=False)
        self.gamma = nn.Parameter(torch.zeros(1))
        # --num_heads > 1 (gma.py:123-126): the heads' outputs are concatenated along the channels and projected back to dim
        self.project = nn.Conv2d(inner, dim, 1, bias=False) if dim != inner else None

    def forward_tokens(self, attn: torch.Tensor, mf: torch.Tensor, prec: int, out: Optional[torch.Tensor] = None):
        """attn [B, heads, N, ldp], mf tokens [B, N, dim] -> tokens [B, N, dim]."""
        ldp = ops.vt_stride(attn)
        vT = ops.linear_t(mf, self.to_v.weight.view(self.heads * self.dim_head, -1), ldp, prec, Dv=self.dim_head)
        O = ops.attn_apply(attn, vT, self.dim_head, prec)                      # [B, heads, N, dim_head]
        if self.project is not None:
            # 'b h (x y) d -> b (h d) x y' + the 1x1 projection (gma.py:135-138): head-major channels per token, one craft_linear
            B, Hd, N, Dv = O.shape
            O = ops.linear(O.permute(0, 2, 1, 3).reshape(B, N, Hd * Dv), self.project.weight.view(self.project.weight.shape[0], -1), None, prec)
        return ops.gma_residual(mf, O, self.gamma, out=out)
