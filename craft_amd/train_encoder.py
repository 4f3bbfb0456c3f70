"""BasicEncoder (extractor.py:124-196) in TRAINING mode on the HIP kernels: forward with the normalisation layers as real
operators (InstanceNorm for ``fnet``; BatchNorm with batch statistics — or running statistics under ``freeze_bn`` — for ``cnet``)
and a backward made of the library's own kernels, so that ``model.train()`` needs no PyTorch / MIOpen compute kernel.

Composition (every box is one ``torch.autograd.Function`` whose forward / backward call the C ABI):

* ``Stem``      7x7 / stride-2 conv of the raw image (input normalisation fused) — ``craft_stem_conv7x7[_mfma]``; weight gradient
                = ``craft_stem_im2col`` + one k-major ``craft_gemm`` (K = all output pixels); no input gradient (images).
* ``EncConv``   3x3 / 1x1 conv, stride 1 or 2 — ``craft_conv2d_nhwc_ex`` (per-image output statistics from the epilogue).
                Backward: a stride-2 conv is the stride-1 conv sampled at even positions, so its gradient is zero-stuffed
                (``craft_zero_stuff2``) and then goes through the stride-1 machinery: input gradient = the forward kernel with
                flipped / transposed weights, weight gradient = ``craft_conv2d_wgrad``, bias gradient = ``craft_colsum``.
* ``NormAct``   normalisation + ReLU (+ the residual tail ``relu(x + y)``) — ``craft_norm_act_fwd`` / ``_bwd_reduce`` / ``_bwd_apply``.
* the final 1x1 conv is ``autograd.Linear``.

Activations are channels-last tokens ``[B, H*W, C]`` throughout; the output is the ``[B, (H/8)*(W/8), output_dim]`` token tensor the
hot path consumes.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.autograd import Function

from . import autograd as AG
from . import hip, ops
from .extractor import BasicEncoder
from .hip import ACT_NONE, ACT_RELU, PREC_BF16, PREC_F16X3, PREC_F32, STATS_REPLICAS, W_PACKED, call, pick

EPS = 1e-5        # nn.InstanceNorm2d / nn.BatchNorm2d default (extractor.py uses the defaults)


PENDING_BN_COUNTS = []


def flush_bn_counts():
    """num_batches_tracked += 1 of every BatchNorm the training forward went through (nn.BatchNorm2d does it per module: 15 scalar
    kernels per step)."""
    if PENDING_BN_COUNTS:
        with torch.no_grad():
            torch._foreach_add_(PENDING_BN_COUNTS, 1)
        PENDING_BN_COUNTS.clear()


class Stem(Function):
    @staticmethod
    def forward(ctx, raw, w, b, prec, bias_dead=False):
        ctx.set_materialize_grads(False)          # (the statistics output never has a gradient: no zero tensor made for it)
        B, _, H, W = raw.shape
        raw = raw.contiguous().float()
        out = torch.empty(B, (H // 2) * (W // 2), 64, device=raw.device, dtype=torch.float32)
        stats = hip.zeros((STATS_REPLICAS, B, 64, 2,), raw.device, torch.float64)
        bias = b.detach().float().contiguous()
        if prec == PREC_F32:
            wk = w.detach().float().permute(2, 3, 1, 0).reshape(147, 64).contiguous()
            call("craft_stem_conv7x7", raw, wk, bias, ACT_NONE, B, H, W, out, stats)
        else:
            # k = (ky*3 + c)*8 + kx, kx padded 7 -> 8, K padded 168 -> 192 (craft_stem_conv7x7_mfma)
            wk = torch.zeros(64, 7, 3, 8, device=raw.device, dtype=torch.float32)
            wk[..., :7] = w.detach().float().permute(0, 2, 1, 3)
            wm = torch.zeros(64, 192, device=raw.device, dtype=torch.float32)
            wm[:, :168] = wk.reshape(64, 168)
            planes = 2 if prec == PREC_F16X3 else 1
            packed = torch.empty(planes * 64 * 192, device=raw.device, dtype=torch.bfloat16 if prec == PREC_BF16 else torch.float16)
            call("craft_pack_weights", wm, 64, 192, prec, packed)
            call("craft_stem_conv7x7_mfma", raw, packed, bias, ACT_NONE, B, H, W, out, stats, prec)
        ctx.save_for_backward(raw)
        ctx.prec, ctx.bias_dead = prec, bias_dead
        ctx.bw_modes = AG.modes()
        ctx.mark_non_differentiable(stats)
        return out, stats

    @staticmethod
    def backward(ctx, dy, _):
        AG.use_modes(ctx.bw_modes)
        (raw,) = ctx.saved_tensors
        B, _, H, W = raw.shape
        dy = AG._rows(dy)
        P = B * (H // 2) * (W // 2)
        dw = db = None
        if ctx.needs_input_grad[1]:
            cols = torch.empty(P, 160, device=raw.device, dtype=torch.float32)
            call("craft_stem_im2col", raw, B, H, W, cols)
            dwc = hip.zeros((64, 160,), raw.device)
            if AG._use_pk(ctx.prec):
                AG.wgrad_pk([(AG.Packed(dy, AG.gprec(ctx.prec)), AG.Packed(cols, AG.xprec(ctx.prec)))], 1, 1, dwc)
            else:
                AG.gemm(dy, 1, dy.stride(-2), 0, 0, cols, 1, 160, 0, 0, dwc, 160, 0, 0, 1, 1, 64, 160, P, accumulate=True, ksplit=0, prec=ctx.prec)
            dw = dwc[:, :147].reshape(64, 7, 7, 3).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[2]:
            db = hip.zeros((64,), raw.device)
            if not ctx.bias_dead:
                call("craft_colsum", dy, dy.stride(-2), P, 64, db)
        return None, dw, db, None, None


class EncConv(Function):
    """nn.Conv2d(k = 3 pad 1 | k = 1, stride 1 | 2) + bias on tokens, plus the per-(image, channel) (sum, sum^2) of the output."""

    @staticmethod
    def forward(ctx, x, w, b, hw_in, stride, prec, cache, bias_dead=False):
        ctx.set_materialize_grads(False)          # (the statistics output never has a gradient: no zero tensor made for it)
        x = AG._rows(x)
        B, _, Cin = x.shape
        Cout, _, KH, KW = w.shape
        Hin, Win = hw_in
        Ho, Wo = Hin // stride, Win // stride
        halo = stride == 1 and KH * KW > 1 and prec != PREC_F32
        key = (id(w), "fwd", prec)
        wp = cache.get(key)
        if wp is None:
            wp = cache[key] = ops.pack_conv_prec(w, prec) if halo else ops.pack_conv(w)
        y = torch.empty(B, Ho * Wo, Cout, device=x.device, dtype=torch.float32)
        stats = hip.zeros((STATS_REPLICAS, B, Cout, 2,), x.device, torch.float64)
        call("craft_conv2d_nhwc_ex", x, x.stride(1), Cin, Hin, Win, None, wp, b.detach().float().contiguous(), Cout, KH, KW, stride, ACT_NONE,
             y, Cout, B, Ho, Wo, stats, prec | (W_PACKED if halo else 0))
        ctx.save_for_backward(x)
        ctx.w, ctx.cache, ctx.prec, ctx.stride, ctx.hw_in, ctx.bias_dead = w, cache, prec, stride, hw_in, bias_dead
        ctx.bw_modes = AG.modes()
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _):
        AG.use_modes(ctx.bw_modes)
        (x,) = ctx.saved_tensors
        w, prec, stride = ctx.w, ctx.prec, ctx.stride
        B, _, Cin = x.shape
        Cout, _, KH, KW = w.shape
        Hin, Win = ctx.hw_in
        dev = x.device
        dy = AG._rows(dy)
        g = dy
        if stride == 2:
            g = torch.empty(B, Hin * Win, Cout, device=dev, dtype=torch.float32)
            call("craft_zero_stuff2", dy, dy.stride(-2), B, Hin, Win, Cout, g, Cout)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt, zb, flag, _ = AG._conv_weights(w, None, prec, ctx.cache, True)
            dx = torch.empty(B, Hin * Win, Cin, device=dev, dtype=torch.float32)
            call("craft_conv2d_nhwc", g, g.stride(-2), Cout, wt, zb, Cin, KH, KW, ACT_NONE, dx, Cin, B, Hin, Win, prec | flag | AG.dxflag(prec))
        if ctx.needs_input_grad[2]:
            db = hip.zeros((Cout,), dev)
        # a bias in front of a statistics-normalised layer cannot move the loss: its gradient is exactly 0; otherwise the column sums
        # of dY ride on the weight-gradient launch (the zero-stuffed rows of a stride-2 layer add nothing)
        live_db = db if (db is not None and not ctx.bias_dead) else None
        if ctx.needs_input_grad[1]:
            dwp = AG._conv_wgrad(x, g, B, Hin, Win, Cin, Cout, KH, KW, prec, live_db)
            dw = dwp.permute(0, 3, 1, 2)
        elif live_db is not None:
            call("craft_colsum", dy, dy.stride(-2), dy.shape[0] * dy.shape[1], Cout, live_db)
        return dx, dw, db, None, None, None, None, None


class NormAct(Function):
    """out = tail(act((x - mean) * rstd * gamma + beta)); ``mr`` = (mean, rstd) as [B, C, 2] (per image) or [C, 2].
    ``population`` > 0: the statistics are functions of x over that many samples (InstanceNorm: N, BatchNorm training: B*N) and the
    backward carries the two mean terms; 0: constants (BatchNorm with running statistics)."""

    @staticmethod
    def forward(ctx, x, mr, gamma, beta, act, res, population):
        x = AG._rows(x)
        B, N, C = x.shape
        per_image = mr.dim() == 3
        out = torch.empty(B, N, C, device=x.device, dtype=torch.float32)
        r = AG._rows(res) if res is not None else None
        call("craft_norm_act_fwd", x, x.stride(-2), mr, int(per_image), gamma, beta, act, r, r.stride(-2) if r is not None else 0, out, C, B, N, C)
        ctx.save_for_backward(x, mr, gamma, beta, out if res is not None else None)
        ctx.act, ctx.population, ctx.per_image, ctx.has_res = act, population, per_image, res is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        x, mr, gamma, beta, out = ctx.saved_tensors
        B, N, C = x.shape
        dy = AG._rows(dy)
        dev = x.device
        sums = hip.zeros((B, C, 2,), dev, torch.float64)
        ldo = out.stride(-2) if out is not None else 0
        call("craft_norm_act_bwd_reduce", dy, dy.stride(-2), out, ldo, x, x.stride(-2), mr, int(ctx.per_image), gamma, beta, ctx.act, int(ctx.has_res),
             sums, B, N, C)
        red, red_pi = None, int(ctx.per_image)
        if ctx.population:
            red = torch.empty((B, C, 2) if ctx.per_image else (C, 2), device=dev, dtype=torch.float32)
        dgamma = torch.empty(C, device=dev, dtype=torch.float32) if gamma is not None and ctx.needs_input_grad[2] else None
        dbeta = torch.empty(C, device=dev, dtype=torch.float32) if beta is not None and ctx.needs_input_grad[3] else None
        if red is not None or dgamma is not None or dbeta is not None:
            call("craft_norm_bwd_finalize", sums, B, C, float(ctx.population), red_pi, red, dgamma, dbeta)
        dx = torch.empty(B, N, C, device=dev, dtype=torch.float32)
        dres = torch.empty(B, N, C, device=dev, dtype=torch.float32) if ctx.has_res and ctx.needs_input_grad[5] else None
        call("craft_norm_act_bwd_apply", dy, dy.stride(-2), out, ldo, x, x.stride(-2), mr, int(ctx.per_image), gamma, beta, ctx.act, int(ctx.has_res),
             red, red_pi, dx, C, dres, C, B, N, C)
        return dx, None, dgamma, dbeta, None, dres, None


def _norm(y, stats, count, mod, act, res=None):
    """Apply the module ``mod`` (InstanceNorm2d / BatchNorm2d, training or eval) to the raw conv output y with epilogue statistics."""
    B, N, C = y.shape
    if isinstance(mod, nn.InstanceNorm2d):
        mr = torch.empty(B, C, 2, device=y.device, dtype=torch.float32)
        call("craft_stats_finalize", stats, B * C, float(count), EPS, mr)
        return NormAct.apply(y, mr, None, None, act, res, count)
    if isinstance(mod, nn.BatchNorm2d):
        mr = torch.empty(C, 2, device=y.device, dtype=torch.float32)
        if mod.training:                                           # batch statistics + running-statistics update in one tiny kernel
            m = mod.momentum if mod.momentum is not None else 0.1
            with torch.no_grad():
                call("craft_bn_finalize", stats, B, C, float(count), float(mod.eps), float(m), mr, mod.running_mean, mod.running_var)
                PENDING_BN_COUNTS.append(mod.num_batches_tracked)        # += 1, all of them in one launch (flush_bn_counts)
            return NormAct.apply(y, mr, mod.weight, mod.bias, act, res, B * count)
        call("craft_bn_finalize", None, B, C, float(count), float(mod.eps), 0.0, mr, mod.running_mean, mod.running_var)
        return NormAct.apply(y, mr, mod.weight, mod.bias, act, res, 0)
    raise NotImplementedError(f"training encoder: norm layer {type(mod).__name__} (extractor.py norm_fn 'group' / 'none') is not built")


def supported(enc: BasicEncoder, H: int, W: int) -> bool:
    return enc.norm_fn in ("instance", "batch") and H % 8 == 0 and W % 8 == 0 and enc.dropout is None


def encoder_forward_train(enc: BasicEncoder, raw: torch.Tensor, prec) -> torch.Tensor:
    """raw images [B, 3, H, W] in 0..255 -> tokens [B, (H/8)*(W/8), output_dim] with an autograd graph over the HIP kernels."""
    B, _, H, W = raw.shape
    cp = pick(prec, "enc")
    cache = {}
    hw = (H // 2, W // 2)
    def dead(norm):      # does this norm layer recompute its statistics from its input (then the conv bias in front of it is inert)?
        return isinstance(norm, nn.InstanceNorm2d) or (isinstance(norm, nn.BatchNorm2d) and norm.training)

    y, st = Stem.apply(raw, enc.conv1.weight, enc.conv1.bias, cp, dead(enc.norm1))
    x = _norm(y, st, hw[0] * hw[1], enc.norm1, ACT_RELU)
    for blk in (enc.layer1[0], enc.layer1[1], enc.layer2[0], enc.layer2[1], enc.layer3[0], enc.layer3[1]):
        s = blk.conv1.stride[0]
        hw2 = (hw[0] // s, hw[1] // s)
        n2 = hw2[0] * hw2[1]
        y, st = EncConv.apply(x, blk.conv1.weight, blk.conv1.bias, hw, s, cp, cache, dead(blk.norm1))
        y = _norm(y, st, n2, blk.norm1, ACT_RELU)
        y2, st2 = EncConv.apply(y, blk.conv2.weight, blk.conv2.bias, hw2, 1, cp, cache, dead(blk.norm2))
        if blk.downsample is not None:
            d, std = EncConv.apply(x, blk.downsample[0].weight, blk.downsample[0].bias, hw, s, cp, cache, dead(blk.norm3))
            xr = _norm(d, std, n2, blk.norm3, ACT_NONE)
        else:
            xr = x
        x = _norm(y2, st2, n2, blk.norm2, ACT_RELU, res=xr)
        hw = hw2
    cout = enc.conv2.out_channels
    flush_bn_counts()
    return AG.Linear.apply(x, enc.conv2.weight.view(cout, -1), enc.conv2.bias, cp)
