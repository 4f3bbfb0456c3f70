"""Per-iteration update block of CRAFT on HIP kernels.

Parameter layout of the reference's ``core/update.py``: ``BasicMotionEncoder`` :67-87,
``SepConvGRU`` :37-64, ``FlowHead`` :8-16, mask head :124-127, ``GMAUpdateBlock`` :116-162.
The convolutions run as NHWC implicit GEMMs on MFMA with fused epilogues (ReLU, sigmoid gates,
r*h, GRU blend, 0.25 mask scale); every ``torch.cat`` of the reference is a column range of one
512-wide token buffer ``hx = [h | inp | motion | motion_global]``.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .gma import Aggregate
from .hip import conv_group_prec, PREC_F32, W1X1_PACKED, W_PACKED, call, pick, weights_epoch
from .setrans import ExpandedFeatTrans


PACK_SYNC = [True]      # False inside a training pass (single stream: the packed copies are consumed in stream order; a device
                        # synchronisation after every re-pack -- every step -- would drain the pipeline six times per step)


class _PackCache:
    """Caches re-laid-out weights; invalidated when any source parameter is updated in place or replaced."""

    def __init__(self):
        self._key = None
        self._val = None

    def get(self, params, fn, tag=0):
        key = (tag, weights_epoch()) + tuple((p.data_ptr(), p._version, p.device) for p in params)
        if key != self._key:
            with torch.no_grad():
                self._val = fn()
            self._key = key
            # the packed copy may be consumed on other HIP streams (batch-sliced refinement loop): packing happens
            # once per weight update, so simply finish it before anybody can see it
            if torch.cuda.is_available() and PACK_SYNC[0]:
                torch.cuda.current_stream().synchronize()
        return self._val


class FlowHead(nn.Module):
    def __init__(self, input_dim: int = 128, hidden_dim: int = 256):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, 2, 3, padding=1)
        self._pk = _PackCache()

    def packed(self, prec: int = PREC_F32):
        return self._pk.get([self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias],
                            lambda: (ops.pack_conv_prec(self.conv1.weight, prec), self.conv1.bias.detach().float().contiguous(),
                                     ops.pack_conv(self.conv2.weight), self.conv2.bias.detach().float().contiguous()), tag=prec)


class SepConvGRU(nn.Module):
    def __init__(self, hidden_dim: int = 128, input_dim: int = 192 + 128):
        super().__init__()
        c = hidden_dim + input_dim
        self.hidden_dim, self.input_dim = hidden_dim, input_dim
        self.convz1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convr1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convq1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convz2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convr2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convq2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self._pk = _PackCache()
        self._pk_split = _PackCache()

    def packed(self, prec: int = PREC_F32):
        mods = [self.convz1, self.convr1, self.convq1, self.convz2, self.convr2, self.convq2]
        params = [p for m in mods for p in (m.weight, m.bias)]

        def make():
            out = []
            for z, r, q in ((self.convz1, self.convr1, self.convq1), (self.convz2, self.convr2, self.convq2)):
                out += [ops.pack_conv_prec(torch.cat([z.weight, r.weight], 0), prec),
                        torch.cat([z.bias, r.bias], 0).detach().float().contiguous(),
                        ops.pack_conv_prec(q.weight, prec), q.bias.detach().float().contiguous()]
            return tuple(out)
        return self._pk.get(params, make, tag=prec)

    def packed_split(self, prec: int, c_lo: int, c_hi: int):
        """Weights split for the hoisted-context form: channels [c_lo, c_hi) of the conv input are iteration
        invariant (the context features inp); returns (varying-part weights x4, const-part weights + biases x4)."""
        mods = [self.convz1, self.convr1, self.convq1, self.convz2, self.convr2, self.convq2]
        params = [p for m in mods for p in (m.weight, m.bias)]

        def make():
            var, const = [], []
            cin = self.hidden_dim + self.input_dim
            for z, r, q in ((self.convz1, self.convr1, self.convq1), (self.convz2, self.convr2, self.convq2)):
                for w0, w1, b in ((z.weight, r.weight, torch.cat([z.bias, r.bias], 0)), (q.weight, None, q.bias)):
                    if prec != PREC_F32:        # one launch per operand, straight from the nn.Conv2d layout
                        var.append(ops.pack_conv_weights(w0, prec, w1, sel=((0, c_lo), (c_hi, cin))))
                        const += [ops.pack_conv_weights(w0, prec, w1, sel=((c_lo, c_hi), (0, 0))), b.detach().float().contiguous()]
                    else:
                        w = (torch.cat([w0, w1], 0) if w1 is not None else w0).detach()
                        var.append(ops.pack_conv_prec(torch.cat([w[:, :c_lo], w[:, c_hi:]], 1).contiguous(), prec))
                        const += [ops.pack_conv_prec(w[:, c_lo:c_hi].contiguous(), prec), b.detach().float().contiguous()]
            return tuple(var), tuple(const)
        return self._pk_split.get(params, make, tag=(prec, c_lo, c_hi))

    def context_tokens(self, inp: torch.Tensor, hw, prec) -> torch.Tensor:
        """Once per forward: the contribution of the (iteration-invariant) context features to all six gate
        convolutions, biases included -> fields [B, N, 768]."""
        B, N, cc = inp.shape
        cp = conv_group_prec(prec, "gru")
        _, const = self.packed_split(cp, self.hidden_dim, self.hidden_dim + cc)
        fields = torch.empty(B, N, 768, device=inp.device, dtype=torch.float32)
        call("craft_sepconv_gru_context", inp, inp.stride(1), cc, *const, B, hw[0], hw[1], fields, cp | W_PACKED)
        return fields

    def step_tokens(self, hx: torch.Tensor, hw, ws: torch.Tensor, prec, fields: torch.Tensor, c_lo: int, c_hi: int):
        """One GRU update in place on hx = [h | inp | v]: only h and v enter the K loops, `fields` carries the rest."""
        B, N, ctot = hx.shape
        cp = conv_group_prec(prec, "gru")
        var, _ = self.packed_split(cp, c_lo, c_hi)
        call("craft_sepconv_gru_step", hx, hx.stride(1), c_hi, ctot - c_hi, *var, fields, B, hw[0], hw[1], ws, cp | W_PACKED)

    def forward_tokens(self, hx: torch.Tensor, hw, ws: torch.Tensor, prec: int):
        """In place on hx = [h (128) | x (input_dim)] tokens [B, N, 128+input_dim]."""
        B, N, _ = hx.shape
        H8, W8 = hw
        cp = conv_group_prec(prec, "gru")
        wzr1, bzr1, wq1, bq1, wzr2, bzr2, wq2, bq2 = self.packed(cp)
        call("craft_sepconv_gru", hx, hx.stride(1), self.input_dim, wzr1, bzr1, wq1, bq1, wzr2, bzr2, wq2, bq2, B, H8, W8,
             ws, cp | W_PACKED)


def _dev_index(device) -> int:
    """torch.device('cuda') has index None: the streams / events below are keyed on the device torch would actually use."""
    return device.index if device.index is not None else torch.cuda.current_device()


class BasicMotionEncoder(nn.Module):
    def __init__(self, args):
        super().__init__()
        cor_planes = args.corr_levels * args.corr_multiplier * (2 * args.corr_radius + 1) ** 2
        self.cor_planes = cor_planes
        self.convc1 = nn.Conv2d(cor_planes, 256, 1, padding=0)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 192, 128 - 2, 3, padding=1)
        self._pk = _PackCache()

    def packed(self, prec: int = PREC_F32):
        mods = [self.convc1, self.convc2, self.convf1, self.convf2, self.conv]
        params = [p for m in mods for p in (m.weight, m.bias)]

        def b(m):
            return m.bias.detach().float().contiguous()
        def c1():        # convc1 (1x1): MFMA fragment order for k_gemm_rows_wf unless the group runs in fp32 (CRAFT_W1X1_PACKED)
            w = self.convc1.weight.detach().view(256, -1).float().contiguous()
            return w if (prec == PREC_F32 or os.environ.get("CRAFT_NO_LINEAR_PACK")) else ops.pack_linear_weight(w, prec)
        return self._pk.get(params, lambda: (c1(), b(self.convc1),
                                             ops.pack_conv_prec(self.convc2.weight, prec), b(self.convc2),
                                             ops.pack_convf1(self.convf1.weight) if prec == PREC_F32 else ops.pack_convf1_mfma(self.convf1.weight, prec),
                                             b(self.convf1),
                                             ops.pack_conv_prec(self.convf2.weight, prec), b(self.convf2),
                                             ops.pack_conv_prec(self.conv.weight, prec), b(self.conv)), tag=prec)

    def _flow_side(self, device):
        """(side stream, event) for the flow branch, one pair per calling stream: the C ABI owns no stream or event
        (include/craft_hip.h), the caller hands them in."""
        cache = self.__dict__.setdefault("_side", {})
        key = (_dev_index(device), torch.cuda.current_stream().cuda_stream)
        if key not in cache:
            st, ev = torch.cuda.Stream(device=device), torch.cuda.Event()
            ev.record(st)                       # torch creates the hipEvent_t lazily: materialise the handle
            cache[key] = (st, ev)
        return cache[key]

    def prefork(self, device):
        """Let the flow branch's side stream start from HERE (the caller's stream position now) instead of from the forward_tokens call:
        the branch depends only on `flow`, which is final before the correlation lookup, so forking in front of the lookup gives
        convf1 -> convf2 (86 us) the lookup's 33 us as a head start over convc1 -> convc2 (91 us) -- the join in front of `conv` then
        rarely waits (round-5 kernel trace: 10 us of idle main stream per iteration at that join, profiles/r5/trace_gaps.txt)."""
        if os.environ.get("CRAFT_NO_FORK") or os.environ.get("CRAFT_NO_PREFORK"):
            return
        side, _ = self._flow_side(device)
        side.wait_stream(torch.cuda.current_stream())
        self.__dict__.setdefault("_preforked", set()).add((_dev_index(device), torch.cuda.current_stream().cuda_stream))

    def begin_pass(self):
        """Forget head starts that were announced but never consumed (a prefork() followed by fork=False or by an exception): a stale entry
        would let a later forward_tokens skip its wait and read `flow` / `ws` before they are final.  network.py calls this in front of
        every refinement loop."""
        self.__dict__.pop("_preforked", None)

    def forward_tokens(self, flow: torch.Tensor, corr: torch.Tensor, hw, out: torch.Tensor, ws: torch.Tensor, prec: int,
                       fork: bool = True):
        """flow tokens [B,N,2], corr tokens [B,N,cor_planes] -> out tokens view [B,N,128].  ``fork``: the flow branch
        (convf1 -> convf2) runs on a side stream next to the correlation branch."""
        B, N, _ = flow.shape
        H8, W8 = hw
        cp = conv_group_prec(prec, "menc")
        wc1, bc1, wc2, bc2, wf1, bf1, wf2, bf2, wcv, bcv = self.packed(cp)
        st = ev = None
        if not fork and self.__dict__.get("_preforked"):
            self._preforked.discard((_dev_index(flow.device), torch.cuda.current_stream().cuda_stream))      # an unused head start dies here
        if fork and not os.environ.get("CRAFT_NO_FORK"):
            side, event = self._flow_side(flow.device)
            key = (_dev_index(flow.device), torch.cuda.current_stream().cuda_stream)
            pre = self.__dict__.get("_preforked")
            if pre and key in pre:
                pre.discard(key)                                # (prefork(): the side stream already waits for the caller's earlier position)
            else:
                side.wait_stream(torch.cuda.current_stream())   # the side stream sees `flow` / `ws` as the caller left them
            st, ev = side.cuda_stream, event.cuda_event
        call("craft_motion_encoder", corr, corr.stride(1), self.cor_planes, flow, wc1, bc1, wc2, bc2, wf1, bf1, wf2, bf2,
             wcv, bcv, B, H8, W8, out, out.stride(1), ws,
             cp | W_PACKED | (W1X1_PACKED if (cp != PREC_F32 and not os.environ.get("CRAFT_NO_LINEAR_PACK")) else 0), st, ev)


class GMAUpdateBlock(nn.Module):
    def __init__(self, args, hidden_dim: int = 128):
        super().__init__()
        self.args = args
        self.encoder = BasicMotionEncoder(args)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(256, 64 * 9, 1, padding=0))
        self.use_setrans = args.use_setrans
        if self.use_setrans:
            self.intra_trans_config = args.intra_trans_config
            self.aggregator = ExpandedFeatTrans(self.intra_trans_config, "Motion Aggregator")
        else:
            self.aggregator = Aggregate(args=args, dim=128, dim_head=128, heads=args.num_heads)
        self._pk_mask = _PackCache()

    def packed_mask(self, prec: int = PREC_F32):
        m0, m2 = self.mask[0], self.mask[2]
        return self._pk_mask.get([m0.weight, m0.bias, m2.weight, m2.bias],
                                 lambda: (ops.pack_conv_prec(m0.weight, prec), m0.bias.detach().float().contiguous(),
                                          m2.weight.detach().view(576, -1).float().contiguous(),
                                          m2.bias.detach().float().contiguous()), tag=prec)

    # -- token-level steps used by CRAFT.forward ---------------------------------------------------
    def step_tokens(self, hx, corr, flow, attention, hw, ws, prec, gru_fields=None):
        """One refinement step up to the new hidden state (update.py:137-156), in place on
        hx = [net | inp | motion | motion_global] tokens [B, N, 512].  With ``gru_fields`` (from
        ``gru.context_tokens(inp)``) the GRU skips the iteration-invariant context channels."""
        self.encoder.forward_tokens(flow, corr, hw, hx[..., 256:384], ws, prec)
        mf = hx[..., 256:384]
        if self.use_setrans:
            self.aggregator(mf, attention, out=hx[..., 384:512], prec=prec)
        else:
            self.aggregator.forward_tokens(attention, mf, prec, out=hx[..., 384:512])
        if gru_fields is None:
            self.gru.forward_tokens(hx, hw, ws, prec)
        else:
            self.gru.step_tokens(hx, hw, ws, prec, gru_fields, 128, 256)

    def flow_head_tokens(self, hx, hw, coords1, coords0, flow, delta, ws, prec):
        B, N, _ = hx.shape
        cp = conv_group_prec(prec, "fh")
        w1, b1, w2, b2 = self.flow_head.packed(cp)
        call("craft_flow_head", hx, hx.stride(1), w1, b1, w2, b2, B, hw[0], hw[1], coords1, coords0, flow, delta, ws,
             cp | W_PACKED)

    def mask_tokens(self, hx, hw, ws, prec, out: Optional[torch.Tensor] = None):
        B, N, _ = hx.shape
        if out is None:
            out = torch.empty(B, N, 576, device=hx.device, dtype=torch.float32)
        cp = conv_group_prec(prec, "mask")
        w0, b0, w2, b2 = self.packed_mask(cp)
        call("craft_mask_head", hx, hx.stride(1), w0, b0, w2, b2, B, hw[0], hw[1], out, ws, cp | W_PACKED)
        return out

    @staticmethod
    def workspace(B: int, N: int, device) -> torch.Tensor:
        """Scratch for one step: max(motion encoder 640, GRU 256, heads 256) floats per pixel."""
        return torch.empty(B * N * 640, device=device, dtype=torch.float32)

    def forward(self, net, inp, corr, flow, attention):
        """NCHW interface of the reference (update.py:137-162) -> (net, mask, delta_flow), all NCHW.
        ``attention`` is what ``self.att`` returned ([B, M, N, N] view of the padded P)."""
        prec = getattr(self, "hip_prec", PREC_F32)
        B, _, H8, W8 = net.shape
        N = H8 * W8
        dev = net.device
        hx = torch.empty(B, N, 512, device=dev, dtype=torch.float32)
        ops.tokens_from_nchw(net.float(), out=hx[..., 0:128])
        ops.tokens_from_nchw(inp.float(), out=hx[..., 128:256])
        corr_t = ops.tokens_from_nchw_wide(corr.float())
        flow_t = ops.tokens_from_nchw(flow.float())
        P = attention
        if P.shape[-1] == N and P.stride(-2) == N:            # unpadded caller-made tensor: re-pad
            ldp = ops.round_up(N, 32)
            Pp = torch.zeros(*P.shape[:-1], ldp, device=dev, dtype=P.dtype)
            Pp[..., :N] = P
            P = Pp
        elif P.stride(-2) != P.shape[-1]:                      # the [..., :N] view handed out by att.forward
            P = torch.as_strided(P, (*P.shape[:-1], P.stride(-2)), P.stride())
        ws = self.workspace(B, N, dev)
        self.step_tokens(hx, corr_t, flow_t, P, (H8, W8), ws, prec)
        c0, c1, fl = ops.coords_init(None, B, H8, W8, dev)
        delta = torch.empty(B, N, 2, device=dev, dtype=torch.float32)
        self.flow_head_tokens(hx, (H8, W8), c1, c0, fl, delta, ws, prec)
        mask = self.mask_tokens(hx, (H8, W8), ws, prec)
        return (ops.tokens_to_nchw(hx[..., 0:128], H8, W8), ops.tokens_to_nchw(mask, H8, W8),
                ops.tokens_to_nchw(delta, H8, W8))
