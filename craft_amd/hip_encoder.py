"""BasicEncoder (extractor.py:124-196) on the HIP conv kernels, channels-last, eval mode.

SURVEY.md §8(f) item 2: once the hot path is fused the two CNN encoders (94 of 815 GMAC per pair) are the
next bottleneck — on MIOpen/fp32 they cost ~12 ms of a 37 ms step at configs[1].  Here every convolution
after the 7x7 stem runs on the same MFMA engine as the update block (k_conv_halo for the stride-1 3x3 convs,
k_gemm_conv for the stride-2 ones, the rows GEMM for the final 1x1), and the normalisations never cost a pass:

* BatchNorm (cnet, eval): folded into the conv weights / bias (exact up to rounding), ReLU in the epilogue;
* InstanceNorm (fnet): the producing conv accumulates per-(image, channel) (sum, sum^2) in its epilogue and the
  CONSUMER applies relu((x - mean) * rstd) while it stages its input (conv halo) or in the fused residual tail;
* ``relu(x + y)`` of a residual block is one fused kernel (craft_residual_relu) with both pending norms applied.

The 7x7 / stride-2 / 3-channel stem (3.5 % of the encoder flops) is a direct fp32 kernel (craft_stem_conv7x7) fused
with the input normalisation; no PyTorch / MIOpen compute remains in the forward pass.  Output: tokens
[B, (H/8)*(W/8), output_dim] — exactly the layout the hot path consumes, so no NCHW round trip exists.
Falls back to the PyTorch module (then converted to tokens) for shapes the kernels do not support.
"""
from __future__ import annotations

from typing import Optional

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .extractor import BasicEncoder, ResidualBlock
from .hip import ACT_NONE, ACT_RELU, PREC_F32, STATS_REPLICAS, W_PACKED, call, pick, weights_epoch

IN_EPS = 1e-5   # nn.InstanceNorm2d / nn.BatchNorm2d default eps (extractor.py uses the defaults)


def _fold_bn(conv: nn.Conv2d, bn: Optional[nn.Module]):
    """conv followed by eval-mode BatchNorm -> equivalent (weight, bias)."""
    w, b = conv.weight.detach().float(), conv.bias.detach().float()
    if isinstance(bn, nn.BatchNorm2d):
        s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        w = w * s.view(-1, 1, 1, 1)
        b = (b - bn.running_mean.detach().float()) * s + bn.bias.detach().float()
    return w, b


class _ConvPack:
    """One conv's weights in the layout / precision its kernel wants (halo kernel: packed for `prec`;
    strided or 1x1: raw fp32 [Cout][KH][KW][Cin])."""

    def __init__(self, conv: nn.Conv2d, bn, prec: int):
        w, b = _fold_bn(conv, bn)
        self.cout, self.cin, self.KH, self.KW = w.shape
        self.stride = conv.stride[0]
        self.halo = self.stride == 1 and self.KH * self.KW > 1
        self.packed = self.halo and prec != PREC_F32
        self.w = ops.pack_conv_prec(w, prec) if self.packed else ops.pack_conv(w)
        self.b = b.contiguous()


class HipEncoder:
    """Stateless runner around a ``BasicEncoder`` module (its parameters stay the source of truth)."""

    def __init__(self, enc: BasicEncoder):
        self.enc = enc
        self.kind = enc.norm_fn
        self._key = None
        self._packs = None

    def supported(self, H: int, W: int) -> bool:
        if self.kind not in ("instance", "batch") or self.enc.training:
            return False
        return not (H % 8 or W % 8)

    def _blocks(self):
        e = self.enc
        return [e.layer1[0], e.layer1[1], e.layer2[0], e.layer2[1], e.layer3[0], e.layer3[1]]

    def _get_packs(self, prec: int):
        params = [p for p in self.enc.parameters()] + [b for b in self.enc.buffers()]
        key = (prec, weights_epoch()) + tuple((p.data_ptr(), p._version) for p in params)
        if key != self._key:
            bn = self.kind == "batch"
            packs = []
            with torch.no_grad():
                for blk in self._blocks():
                    d = {"c1": _ConvPack(blk.conv1, blk.norm1 if bn else None, prec),
                         "c2": _ConvPack(blk.conv2, blk.norm2 if bn else None, prec)}
                    if blk.downsample is not None:
                        d["ds"] = _ConvPack(blk.downsample[0], blk.norm3 if bn else None, prec)
                    packs.append(d)
                self._final = (self.enc.conv2.weight.detach().float().view(self.enc.conv2.out_channels, -1).contiguous(),
                               self.enc.conv2.bias.detach().float().contiguous())
                self._final_pk = {}                     # proj precision -> ops.pack_linear_weight of the final 1x1
                w0, b0 = _fold_bn(self.enc.conv1, self.enc.norm1 if bn else None)
                self._stem = (w0.permute(2, 3, 1, 0).reshape(147, 64).contiguous(), b0.contiguous())
                self._stem_mfma = None
                if prec != PREC_F32:
                    # MFMA stem: K ordered (ky, c, kx) with kx padded 7 -> 8 and K padded 168 -> 192 (craft_stem_conv7x7_mfma)
                    wk = torch.zeros(64, 7, 3, 8, device=w0.device, dtype=torch.float32)
                    wk[..., :7] = w0.permute(0, 2, 1, 3)
                    wm = torch.zeros(64, 192, device=w0.device, dtype=torch.float32)
                    wm[:, :168] = wk.reshape(64, 168)
                    planes = 2 if prec == 3 else 1
                    packed = torch.empty(planes * 64 * 192, device=w0.device, dtype=torch.bfloat16 if prec == 1 else torch.float16)
                    call("craft_pack_weights", wm, 64, 192, prec, packed)
                    self._stem_mfma = packed
            self._packs, self._key = packs, key
        return self._packs

    # ------------------------------------------------------------------------------------------
    def _conv(self, x, B, hw_in, pk: _ConvPack, act, prec, in_norm=None, stats=None):
        Hin, Win = hw_in
        Ho, Wo = (Hin // pk.stride, Win // pk.stride)
        y = torch.empty(B, Ho * Wo, pk.cout, device=x.device, dtype=torch.float32)
        call("craft_conv2d_nhwc_ex", x, x.stride(1), pk.cin, Hin, Win, in_norm, pk.w, pk.b, pk.cout, pk.KH, pk.KW, pk.stride, act,
             y, y.stride(1), B, Ho, Wo, stats, prec | (W_PACKED if pk.packed else 0))
        return y, (Ho, Wo)

    def _finalize(self, stats, count):
        _, B, C, _ = stats.shape
        mr = torch.empty(B, C, 2, device=stats.device, dtype=torch.float32)
        call("craft_stats_finalize", stats, B * C, float(count), IN_EPS, mr)
        return mr

    def forward_tokens(self, raw, prec) -> torch.Tensor:
        """raw: images [B, 3, H, W] in 0..255 (the normalisation 2*(x/255)-1 of network.py:169-173 is fused into the
        stem) -> tokens [B, (H/8)*(W/8), output_dim].  ``raw`` may be a pair (frames1, frames2) of equally shaped tensors: the batch is
        their concatenation (extractor.py:171-176), read from the two tensors in place by the MFMA stem."""
        enc = self.enc
        pair = None
        if isinstance(raw, (tuple, list)):
            a_, b_ = raw
            if a_.shape != b_.shape:
                raise ValueError("the two frame batches must have the same shape")
            pair = (a_.float().contiguous(), b_.float().contiguous())
            B, _, H, W = a_.shape
            B *= 2
            raw_dev = a_.device
            cp_ = pick(prec, "enc")
            if not self.supported(H, W) or cp_ == PREC_F32:
                raw, pair = torch.cat(pair, dim=0), None          # (the PyTorch fallback / the direct fp32 stem take one tensor)
        if pair is None:
            B, _, H, W = raw.shape
            raw_dev = raw.device
        cp = pick(prec, "enc")
        if not self.supported(H, W):
            return ops.tokens_from_nchw(enc((2 * (raw / 255.0) - 1.0).contiguous()).float())
        packs = self._get_packs(cp)
        inorm = self.kind == "instance"
        dev = raw_dev
        hw = (H // 2, W // 2)
        # ---- stem: 7x7 / s2 conv (+ folded BatchNorm + ReLU for cnet; raw output + statistics for fnet)
        sw, sb = self._stem
        t = torch.empty(B, hw[0] * hw[1], 64, device=dev, dtype=torch.float32)
        t_norm = None
        def stem(act, stats):
            if self._stem_mfma is not None and pair is not None:
                call("craft_stem_conv7x7_mfma_pair", pair[0], B // 2, pair[1], self._stem_mfma, sb, act, B, H, W, t, stats, cp)
            elif self._stem_mfma is not None:
                call("craft_stem_conv7x7_mfma", raw.contiguous(), self._stem_mfma, sb, act, B, H, W, t, stats, cp)
            else:
                call("craft_stem_conv7x7", raw.contiguous(), sw, sb, act, B, H, W, t, stats)
        if inorm:
            # the (sum, sum^2) tables of ALL convolutions of this forward come out of one zero fill (15 fills of ~1 MB otherwise)
            couts = [64] + [c for pk in packs for c in ([pk["c1"].cout, pk["c2"].cout] + ([pk["ds"].cout] if "ds" in pk else []))]
            pool = torch.zeros(STATS_REPLICAS * B * 2 * sum(couts), device=dev, dtype=torch.float64)
            pool_off = [0]

            def new_stats(cout):
                n = STATS_REPLICAS * B * cout * 2
                v = pool[pool_off[0]:pool_off[0] + n].view(STATS_REPLICAS, B, cout, 2)
                pool_off[0] += n
                return v
            s0 = new_stats(64)
            stem(ACT_NONE, s0)
            t_norm = self._finalize(s0, hw[0] * hw[1])          # norm1 + ReLU are applied lazily by layer1.0
        else:
            stem(ACT_RELU, None)
        for pk in packs:
            if inorm:
                s1 = new_stats(pk["c1"].cout)
                c1, hw1 = self._conv(t, B, hw, pk["c1"], ACT_NONE, cp, in_norm=t_norm, stats=s1)
                n1 = self._finalize(s1, hw1[0] * hw1[1])
                s2 = new_stats(pk["c2"].cout)
                c2, _ = self._conv(c1, B, hw1, pk["c2"], ACT_NONE, cp, in_norm=n1, stats=s2)
                n2 = self._finalize(s2, hw1[0] * hw1[1])
                if "ds" in pk:
                    s3 = new_stats(pk["ds"].cout)
                    xs, _ = self._conv(t, B, hw, pk["ds"], ACT_NONE, cp, stats=s3)
                    n3 = self._finalize(s3, hw1[0] * hw1[1])
                    flags = 1
                else:
                    xs, n3 = t, t_norm                          # identity branch (with the stem's pending norm1 + ReLU)
                    flags = 1 | (2 if t_norm is not None else 0)
                out = torch.empty_like(c2)
                call("craft_residual_relu", xs, xs.stride(1), n3, c2, c2.stride(1), n2, flags, B, hw1[0] * hw1[1], c2.shape[-1],
                     out, out.stride(1))
                t_norm = None
            else:
                # folded BatchNorm: the residual tail relu(x + relu(conv2(..))) is conv2's epilogue (craft_conv2d_nhwc_res)
                c1, hw1 = self._conv(t, B, hw, pk["c1"], ACT_RELU, cp)
                xs = self._conv(t, B, hw, pk["ds"], ACT_NONE, cp)[0] if "ds" in pk else t
                p2 = pk["c2"]
                out = torch.empty(B, hw1[0] * hw1[1], p2.cout, device=dev, dtype=torch.float32)
                call("craft_conv2d_nhwc_res", c1, c1.stride(1), p2.cin, p2.w, p2.b, p2.cout, p2.KH, p2.KW, ACT_RELU, xs, xs.stride(1), out,
                     out.stride(1), B, hw1[0], hw1[1], cp | (W_PACKED if p2.packed else 0))
            t, hw = out, hw1
        wf, bf = self._final
        pp = pick(prec, "proj")
        pkd = None
        if pp != PREC_F32 and not os.environ.get("CRAFT_NO_LINEAR_PACK"):
            pkd = self._final_pk.get(pp)
            if pkd is None:
                pkd = self._final_pk[pp] = ops.pack_linear_weight(wf, pp)
        return ops.linear(t, wf, bf, prec, packed=pkd)
