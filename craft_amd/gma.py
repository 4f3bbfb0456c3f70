"""GMA attention / aggregation (the ``use_setrans=False`` variant) on HIP kernels.

Reference: ``core/gma.py`` — ``Attention`` :53-102 (content-only scores softmax(scale*q.k); the
position_only / position_and_content flags are off by default and not implemented here),
``Aggregate`` :105-142 (fmap + gamma * attn . to_v(fmap)).  ``RelPosEmb`` parameters are declared
so checkpoints load, but are unused by the content-only path (exactly as in the reference).
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
        if getattr(args, "position_only", False) or getattr(args, "position_and_content", False):
            raise NotImplementedError("GMA positional scores (gma.py:34-50) are outside the HIP path")
        self.args, self.heads, self.dim_head = args, heads, dim_head
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
        return ops.attn_probs(q, k, H8, W8, self.heads, self.scale, None, 0.0, -1, None, prec, defer=defer)

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
        if dim != inner:
            raise NotImplementedError("multi-head GMA projection (gma.py:123-126) is outside the HIP path")
        self.project = None

    def forward_tokens(self, attn: torch.Tensor, mf: torch.Tensor, prec: int, out: Optional[torch.Tensor] = None):
        """attn [B, heads, N, ldp], mf tokens [B, N, dim] -> tokens [B, N, dim]."""
        ldp = attn.shape[-1]
        vT = ops.linear_t(mf, self.to_v.weight.view(self.heads * self.dim_head, -1), ldp, prec, Dv=self.dim_head)
        O = ops.attn_apply(attn, vT, self.dim_head, prec)
        return ops.gma_residual(mf, O, self.gamma, out=out)
