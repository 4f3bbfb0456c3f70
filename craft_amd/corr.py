"""All-pairs correlation volume, 4-level pyramid and radius-r lookup on HIP kernels.

Interface of the reference's ``core/corr.py``: ``CorrBlock(fmap1, fmap2, ...)`` (plain dot-product
volume, corr.py:16-81) and ``TransCorrBlock(config, ...)`` with ``update(...)`` + ``__call__(coords)``
(cross-attention volume, corr.py:132-207).  The volume is produced by one fused kernel
(``craft_corr_build``): 4-mode Q K^T, clamp, softmax-over-modes pooling, positional bias, written once
into pyramid level 0 together with (sum, sum^2) for the global LayerNorm, which the lookup applies
lazily per in-bounds tap.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .hip import PREC_F32, weights_epoch
from .setrans import CrossAttFeatTrans, SETransConfig, SETransInputFeatEncoder


class CorrBlock:
    """Plain correlation volume <fmap1(:,i), fmap2(:,j)>/sqrt(C) (``craft=False`` variant)."""

    def __init__(self, fmap1: torch.Tensor, fmap2: torch.Tensor, num_levels: int = 4, radius: int = 4,
                 do_corr_global_norm: bool = False, prec: int = PREC_F32):
        B, C, H8, W8 = fmap1.shape
        self.num_levels, self.radius = num_levels, radius
        self.shape = (B, H8, W8)
        t1 = ops.tokens_from_nchw(fmap1)
        t2 = ops.tokens_from_nchw(fmap2)
        self.pyramid = ops.CorrPyramid(B, H8, W8, num_levels, fmap1.device)
        ops.corr_build(t1, t2, H8, W8, 1, 1.0 / math.sqrt(C), None, 0.0, 1.0, None, self.pyramid, do_corr_global_norm, prec)

    def all_pyramids(self):
        """[CorrPyramid] -- two of them for the two-way correlation of ``--f1`` (TransCorrBlock)."""
        return getattr(self, "pyramids", None) or [self.pyramid]

    def lookup_tokens(self, coords_tokens: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        return ops.corr_lookup(self.all_pyramids(), coords_tokens, self.radius, out=out)

    def __call__(self, coords: torch.Tensor) -> torch.Tensor:
        """coords NCHW [B, 2, H8, W8] (x, y) -> [B, L*(2r+1)^2, H8, W8]   (corr.py:47-71)."""
        B, H8, W8 = self.shape
        ct = ops.tokens_from_nchw(coords.float())
        return ops.tokens_to_nchw(self.lookup_tokens(ct), H8, W8)


class TransCorrBlock(CorrBlock, nn.Module):
    def __init__(self, config: SETransConfig, num_levels: int = 4, radius: int = 4, do_corr_global_norm: bool = False):
        nn.Module.__init__(self)
        self.num_levels, self.radius = num_levels, radius
        self.config = config
        self.setrans = CrossAttFeatTrans(config, "Inter-frame correlation block")
        self.vispos_encoder = SETransInputFeatEncoder(config)
        self.do_corr_global_norm = do_corr_global_norm
        self.pyramid = None
        self.pyramids = []
        self.shape = None

    def _w_aggr(self, st) -> float:
        """The scalar weight of the softmax-over-modes pooling (a 1x1 nn.Linear) as a host float, read back once per
        parameter version: a per-forward ``.item()`` is a host sync on the hot path (and cannot be graph-captured)."""
        w = st.attn_softaggr.feat2score.weight
        key = (weights_epoch(), w.data_ptr(), w._version, w.device)
        if getattr(self, "_w_aggr_key", None) != key:
            self._w_aggr_val, self._w_aggr_key = float(w.detach().float().item()), key
        return self._w_aggr_val

    def _build(self, x1_ln: torch.Tensor, x2_ln: torch.Tensor, hw, prec: int, slot: int):
        H8, W8 = hw
        B = x1_ln.shape[0]
        st = self.setrans
        q = ops.linear(x1_ln, st.query.weight, st.query.bias, prec, packed=ops.linear_pack(st, "query", st.query.weight, prec))
        k = ops.linear(x2_ln, st.key.weight, st.key.bias, prec, packed=ops.linear_pack(st, "key", st.key.weight, prec))
        scale = 1.0 / math.sqrt(st.attention_mode_dim)
        mx = ops.score_max(q, k, H8, W8, st.num_modes, scale, prec)
        while len(self.pyramids) <= slot:
            self.pyramids.append(None)
        pyr = self.pyramids[slot]
        tiled = ops.fused_pyramid(q.shape[-1], st.num_modes, prec, self.num_levels, H8, W8) and not os.environ.get("CRAFT_NO_TILED_PYRAMID")
        if pyr is None or (pyr.B, pyr.H8, pyr.W8, pyr.tiled) != (B, H8, W8, tiled) or pyr.lv[0].device != q.device:
            pyr = self.pyramids[slot] = ops.CorrPyramid(B, H8, W8, self.num_levels, q.device, tiled=tiled)
        w_aggr = self._w_aggr(st) if st.num_modes > 1 else 1.0
        ops.corr_build(q, k, H8, W8, st.num_modes, scale, self.vispos_encoder.pos_table, float(st.pos_code_weight),
                       w_aggr, mx, pyr, self.do_corr_global_norm, prec)

    def update_tokens(self, x1_ln: torch.Tensor, x2_ln: torch.Tensor, hw, prec: int, x1o_ln=None, x2o_ln=None):
        """x1_ln / x2_ln: LayerNorm-ed tokens of the two frames' features (transformed where a transformer exists).
        With ``x1o_ln`` / ``x2o_ln`` (the conv features, ``--f1 shared|private``) the correlation is two-way
        (corr.py:164-171): volume 0 = (transformed 1, conv 2), volume 1 = (conv 1, transformed 2)."""
        B = x1_ln.shape[0]
        if x1o_ln is not None and x2o_ln is not None:
            self._build(x1_ln, x2o_ln, hw, prec, 0)
            self._build(x1o_ln, x2_ln, hw, prec, 1)
            del self.pyramids[2:]
        else:
            self._build(x1_ln, x2_ln, hw, prec, 0)
            del self.pyramids[1:]
        self.pyramid = self.pyramids[0]
        self.shape = (B, hw[0], hw[1])

    def update(self, fmap1, fmap2, fmap1o=None, fmap2o=None, coords1=None, coords2=None):
        """corr.py:148-189: single-way, or two-way when both fmap1o and fmap2o (the conv features) are given."""
        B, C, H8, W8 = fmap1.shape
        enc, hw = self.vispos_encoder, (H8, W8)
        pos1 = None
        if coords1 is not None and enc.pos_code_type != "bias":          # corr.py:153: (x, y) grid coordinates -> (y, x) positions
            pos1 = ops.tokens_from_nchw(coords1.float()).flip(-1)
        tok = lambda f, pos=None: enc.ln_tokens(ops.tokens_from_nchw(f.float()), hw, pos)      # noqa: E731
        x1, x2 = tok(fmap1, pos1), tok(fmap2, pos1)          # (frame 2 at coords1 too: the reference's eval-mode cache, setrans.py:744-758)
        two = fmap1o is not None and fmap2o is not None
        x1o = tok(fmap1o, pos1) if two else None
        x2o = tok(fmap2o, pos1) if two else None
        self.update_tokens(x1, x2, (H8, W8), getattr(self, "hip_prec", PREC_F32), x1o, x2o)
