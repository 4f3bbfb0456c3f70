"""CRAFT model with the inner-loop hot path on hand-written gfx950 kernels.

Drop-in for the reference's ``core.network.CRAFT`` (network.py:26-267): same constructor
arguments (an argparse Namespace; the ctor writes ``corr_levels``, ``corr_multiplier`` and the
three ``*_trans_config`` objects back into it), same ``forward(image1, image2, iters, flow_init,
upsample, test_mode)`` signature / return conventions, same ``state_dict`` keys (202 in the
canonical ``--craft --f2 full --setrans`` configuration), ``freeze_bn()``.

Execution: the CNN encoders run on the same HIP conv engine (craft_amd/hip_encoder.py; ``args.hip_encoders=False``
keeps them on PyTorch-ROCm / MIOpen); everything after them — F2 transformer,
intra-frame attention, correlation volume + pyramid, and the T refinement iterations — runs through
``libcraft_hip.so`` on channels-last token buffers.  ``model.eval()`` / ``torch.no_grad()`` takes the fused inference
path (no autograd graph); ``model.train()`` with gradients enabled takes the training path (craft_amd/autograd.py:
every hot-path op is an ``autograd.Function`` whose forward AND backward are HIP kernels).

Extra (non-reference) argument fields, all optional:
  ``hip_precision``: "fp32" | "mixed" | "fp16" | "bf16" MFMA operand precision of the hot path, or per role,
                     e.g. "score=bf16,pv=fp16,conv=fp32,proj=fp32" (craft_amd.hip.Precision).  Default: "fp32"
                     when ``mixed_precision`` is False (the reference's fp32 path); "mixed" when it is True (the reference
                     autocasts to fp16 there) = "proj=f16x3,score=f16x3,pv=fp16,conv=f16x3": split-fp16 MFMA (3 fp16
                     MFMAs per product, fp32-class) for projections, Q K^T and all convolutions, fp16 P.V.
"""
from __future__ import annotations

import os

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .corr import CorrBlock, TransCorrBlock
from .extractor import BasicEncoder
from .hip_encoder import HipEncoder
from .gma import Attention
from .hip import ACT_RELU, ACT_TANH, PREC_BF16, PREC_F32, PREC_NAMES, Precision
from .setrans import SelfAttVisPosTrans, SETransConfig
from .update import GMAUpdateBlock


class _JoinOnError:
    """Fork / join hygiene of CRAFT.forward: if the pass leaves between a fork and its join (a launch that fails, a refused shape), the side
    streams are still joined into the caller's stream -- tensors allocated on the caller's stream and written by side-stream kernels go back
    to the caching allocator's pool of the CALLER's stream when the exception unwinds, and its next allocation must not race with them."""

    def __init__(self):
        self.pairs = []

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is not None:
            for main, side in self.pairs:
                try:
                    main.wait_stream(side)
                except Exception:          # noqa: BLE001  (the original exception is the one to report)
                    pass
        self.pairs.clear()
        return False


def _autocast(enabled: bool):
    """The reference runs the CNN encoders under fp16 autocast when mixed_precision is set (network.py:179);
    here the encoders stay fp32 unless ``args.encoder_autocast`` is set, because their rounding dominates the
    end-point deviation of the otherwise fp32-accumulated hot path."""
    return torch.autocast(device_type="cuda", dtype=torch.float16, enabled=bool(enabled) and torch.cuda.is_available())


class GraphedForward:
    """One forward pass of a CRAFT model (inference path) recorded as a hipGraph and replayed per call.

    The inference pass enqueues ~600 kernels on up to three HIP streams with a fixed launch sequence for a given input shape (no host
    read-back, no data-dependent launch geometry), so the whole dependency structure -- including the fork / join of the context chain and
    of the motion encoder's flow branch -- can be handed to the runtime at once: ``torch.cuda.graph`` capture over the model's own streams,
    ``replay()`` per call.  Same kernels, same order of floating-point operations: the replay is bit-identical to the eager pass
    (tests/test_graphed_forward.py); what it removes are the launch gaps between dependent kernels (tools/graph_probe.py: 16.42 -> 16.26 ms
    at 448x1024 x 4, 5.43 -> 5.29 ms at 368x768 x 1).

    ``__call__(image1, image2, flow_init=None)`` copies the inputs into the graph's static buffers, replays, and returns what
    ``CRAFT.forward`` returns for the captured ``test_mode`` -- tensors OWNED BY THE GRAPH, overwritten by the next call (``.clone()`` what must
    outlive it).  The graph reads the weight operands packed at capture time; a parameter or buffer changed since (optimizer step,
    ``load_state_dict``) is detected by its version counter and the pass is re-captured."""

    def __init__(self, model, image1, image2, iters=12, flow_init=None, test_mode=1, warmup=2):
        if model.training:
            raise RuntimeError("GraphedForward records the inference path: call model.eval() first")
        if not image1.is_cuda:
            raise RuntimeError("craft_amd.CRAFT runs its hot path on HIP kernels: inputs must be on the GPU (there is no CPU fallback)")
        if image1.shape != image2.shape:
            raise ValueError("image1 and image2 must have the same shape")
        self.model, self.iters, self.test_mode, self.warmup = model, iters, test_mode, max(1, int(warmup))
        self.im1 = image1.detach().float().contiguous().clone()
        self.im2 = image2.detach().float().contiguous().clone()
        self.flow_init = None if flow_init is None else flow_init.detach().float().contiguous().clone()
        self.graph, self.out, self.versions, self.replays = None, None, None, 0
        self._stream = torch.cuda.Stream(device=image1.device)
        self._capture()

    def _versions(self):
        """What the recorded launches depend on besides the inputs: every parameter / buffer (by version counter) and the precision policy."""
        m = self.model
        a = m.args
        return (str(getattr(a, "hip_precision", None)), bool(getattr(a, "mixed_precision", False)),
                tuple(t._version for t in list(m.parameters()) + list(m.buffers())))

    def _run(self):
        return self.model(self.im1, self.im2, iters=self.iters, flow_init=self.flow_init, test_mode=self.test_mode)

    def _capture(self):
        cur, s = torch.cuda.current_stream(self.im1.device), self._stream
        s.wait_stream(cur)
        with torch.no_grad(), torch.cuda.stream(s):
            for _ in range(self.warmup):        # weight packs, position codes, side streams and workspaces exist before the capture
                self._run()
        cur.wait_stream(s)
        torch.cuda.synchronize(self.im1.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, stream=s):
            self.out = self._run()
        self.versions = self._versions()

    def __call__(self, image1, image2, flow_init=None):
        if tuple(image1.shape) != tuple(self.im1.shape) or tuple(image2.shape) != tuple(self.im2.shape):
            raise ValueError(f"this graph was captured for images of shape {tuple(self.im1.shape)}, got {tuple(image1.shape)}")
        if (flow_init is None) != (self.flow_init is None):
            raise ValueError("this graph was captured " + ("with" if self.flow_init is not None else "without") + " flow_init")
        if self.model.training:
            raise RuntimeError("GraphedForward replays the inference path: the model is in training mode")
        if self._versions() != self.versions:   # weights changed since the capture: the graph holds the OLD packed operands
            self.graph, self.out = None, None
            self._capture()
        self.im1.copy_(image1)
        self.im2.copy_(image2)
        if flow_init is not None:
            self.flow_init.copy_(flow_init)
        self.graph.replay()
        self.replays += 1
        self.model.call_counter += 1
        return self.out


class CRAFT(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.hidden_dim = hdim = 128
        self.context_dim = cdim = 128
        args.corr_levels = 4
        if "dropout" not in vars(args):
            args.dropout = 0
        if getattr(args, "corr_radius", -1) == -1:
            args.corr_radius = 4
        for name, default in (("f1trans", "none"), ("f2trans", "full"), ("mixed_precision", False),
                              ("pos_bias_radius", 7), ("num_heads", 1), ("position_only", False),
                              ("position_and_content", False), ("f2_pos_code_weight", 0.5), ("f2_attn_mask_radius", -1),
                              ("inter_num_modes", 4), ("intra_num_modes", 4), ("f2_num_modes", 4),
                              ("inter_qk_have_bias", True), ("inter_pos_code_type", "bias"), ("inter_pos_code_weight", 0.5),
                              ("intra_pos_code_type", "bias"), ("intra_pos_code_weight", 1.0)):
            if not hasattr(args, name):
                setattr(args, name, default)
        if args.f1trans not in ("none", "shared", "private"):
            raise ValueError(f"--f1 {args.f1trans!r}: expected none | shared | private (network.py:94-103)")
        if args.f1trans != "none" and not args.craft:
            raise NotImplementedError("--f1 needs the cross-attention correlation (--craft): with the plain CorrBlock the "
                                      "reference builds a 324-plane volume for a 648-plane motion encoder and fails")
        if args.f2trans == "none":
            # the reference itself raises AttributeError here (corr_multiplier is never set)
            raise NotImplementedError("--f2 none is not supported (the reference crashes on it as well)")

        if args.craft:
            cfg = SETransConfig()
            cfg.update_config(args)
            cfg.in_feat_dim = cfg.feat_dim = 256
            cfg.max_pos_size = 160
            cfg.out_attn_scores_only = True
            cfg.num_modes = args.inter_num_modes
            cfg.tie_qk_scheme = "shared"
            cfg.qk_have_bias = args.inter_qk_have_bias
            cfg.pos_code_type = args.inter_pos_code_type
            cfg.pos_code_weight = args.inter_pos_code_weight
            self.inter_trans_config = args.inter_trans_config = cfg
            self.corr_fn = TransCorrBlock(cfg, radius=args.corr_radius, do_corr_global_norm=True)

        self.fnet = BasicEncoder(output_dim=256, norm_fn="instance", dropout=args.dropout)
        self.cnet = BasicEncoder(output_dim=hdim + cdim, norm_fn="batch", dropout=args.dropout)

        cfg = SETransConfig()
        cfg.update_config(args)
        cfg.in_feat_dim = cfg.feat_dim = 256
        cfg.has_input_skip = True
        cfg.has_FFN = False
        cfg.attn_mask_radius = args.f2_attn_mask_radius
        cfg.tie_qk_scheme = None
        cfg.qk_have_bias = False
        cfg.out_attn_probs_only = False
        cfg.num_modes = args.f2_num_modes
        cfg.pos_code_type = args.intra_pos_code_type
        cfg.pos_code_weight = args.f2_pos_code_weight
        self.f2_trans_config = args.f2_trans_config = cfg
        self.f2_trans = SelfAttVisPosTrans(cfg, "F2 transformer")
        # --f1 (network.py:94-103): frame 1 gets the same ("shared": one module under two names, so the state dict
        # carries its tensors twice) or its own ("private") transformer, and the correlation becomes two-way
        if args.f1trans == "shared":
            self.f1_trans = self.f2_trans
        elif args.f1trans == "private":
            self.f1_trans = SelfAttVisPosTrans(cfg, "F1 transformer")
        else:
            self.f1_trans = None
        args.corr_multiplier = 2 if self.f1_trans is not None else 1

        if args.use_setrans:
            cfg = SETransConfig()
            cfg.update_config(args)
            cfg.in_feat_dim = cfg.feat_dim = 128
            cfg.has_FFN = False
            cfg.has_input_skip = True
            cfg.attn_mask_radius = -1
            cfg.tie_qk_scheme = None
            cfg.qk_have_bias = False
            cfg.out_attn_probs_only = True
            cfg.num_modes = args.intra_num_modes
            cfg.pos_code_type = args.intra_pos_code_type
            cfg.pos_code_weight = args.intra_pos_code_weight
            self.intra_trans_config = args.intra_trans_config = cfg
            self.att = SelfAttVisPosTrans(cfg, "Intra-frame attention")
        else:
            self.att = Attention(args=args, dim=cdim, heads=args.num_heads, max_pos_size=160, dim_head=cdim)

        self.update_block = GMAUpdateBlock(args, hidden_dim=hdim)
        self.call_counter = 0
        # runners that execute the two BasicEncoders on the HIP conv kernels (eval mode; args.hip_encoders=False
        # keeps them on PyTorch-ROCm / MIOpen)
        self._henc_f = HipEncoder(self.fnet)
        self._henc_c = HipEncoder(self.cnet)

    # ------------------------------------------------------------------------------------------
    def hip_prec(self) -> Precision:
        spec = getattr(self.args, "hip_precision", None)
        if spec is None:
            spec = "mixed" if getattr(self.args, "mixed_precision", False) else "fp32"
        return Precision.parse(spec)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def capture(self, image1, image2, iters=12, flow_init=None, test_mode=1, warmup=2) -> "GraphedForward":
        """The inference pass for inputs of this shape as ONE hipGraph (see GraphedForward): ``g = model.capture(im1, im2); lo, up = g(im1, im2)``."""
        return GraphedForward(self, image1, image2, iters=iters, flow_init=flow_init, test_mode=test_mode, warmup=warmup)

    def initialize_flow(self, img):
        """coords0, coords1 as NCHW grids (network.py:142-149); kept for API parity."""
        B, _, H, W = img.shape
        ys, xs = torch.meshgrid(torch.arange(H // 8, device=img.device, dtype=torch.float32),
                                torch.arange(W // 8, device=img.device, dtype=torch.float32), indexing="ij")
        c = torch.stack([xs, ys], dim=0)[None].expand(B, -1, -1, -1)
        return c, c.clone()

    def upsample_flow(self, flow, mask):
        """NCHW flow [B,2,H8,W8] and mask [B,576,H8,W8] -> [B,2,H,W]  (network.py:151-162)."""
        B, _, H8, W8 = flow.shape
        return ops.convex_upsample(ops.tokens_from_nchw_wide(mask.float()).contiguous(), ops.tokens_from_nchw(flow.float()),
                                   H8, W8)

    # ------------------------------------------------------------------------------------------
    def _streams(self, n: int, dev):
        """Side streams (context chain, batch-sliced refinement loop), created once per device and count."""
        cache = self.__dict__.setdefault("_stream_cache", {})
        key = (dev.index, n)
        if key not in cache:
            cache[key] = [torch.cuda.Stream(device=dev) for _ in range(n)]
        return cache[key]

    def _context_chain(self, cn_tok, hx, hw, prec):
        """Context split + intra-frame attention (network.py:206-214) + the GRU's context fields: fills hx[..., 0:256],
        returns (attention, gru_fields)."""
        ops.tokens_slice(cn_tok, 0, 128, act=ACT_TANH, out=hx[..., 0:128])
        ops.tokens_slice(cn_tok, 128, 128, act=ACT_RELU, out=hx[..., 128:256])
        if self.args.use_setrans:
            xc = self.att.vispos_encoder.ln_tokens(hx[..., 128:256], hw)
            attention = self.att.forward_tokens(xc, hw, prec=prec, defer=True)    # [B, 4, N, ldp] (+ row sums)
        else:
            attention = self.att.forward_tokens(hx[..., 128:256], hw, prec, defer=True)
        # the context features are the same in every iteration: hoist their share of the GRU convolutions
        return attention, self.update_block.gru.context_tokens(hx[..., 128:256], hw, prec)

    def forward(self, image1, image2, iters=12, flow_init=None, upsample=True, test_mode=0):
        """Estimate optical flow between a pair of frames (network.py:164-267)."""
        if not image1.is_cuda:
            raise RuntimeError("craft_amd.CRAFT runs its hot path on HIP kernels: inputs must be on the GPU "
                               "(there is no CPU fallback)")
        if torch.is_grad_enabled() and self.training:
            # model.train() with autograd on: the differentiable composition of the same operators (craft_amd/train_forward.py,
            # backward kernels in craft_amd/autograd.py); returns what the reference returns for the requested test_mode
            if test_mode != 0:
                raise NotImplementedError("model.train() with gradients enabled returns the list of predictions (test_mode=0), the "
                                          "way train.py:228 calls it; use model.eval() / torch.no_grad() for test_mode 1 / 2")
            from .train_forward import forward_train
            preds = forward_train(self, image1, image2, iters=iters, flow_init=flow_init)
            self.call_counter += 1
            return preds
        if self.training:
            # model.train() under torch.no_grad(): the reference would run dropout and BatchNorm batch statistics without a graph.
            # The inference kernels implement neither (BatchNorm folded into the weights, no dropout), so refuse instead of
            # silently computing something else; validation calls model.eval() first (train.py:253-262 / evaluate.py)
            raise NotImplementedError("model.train() under torch.no_grad(): call model.eval() for inference (the inference path has no "
                                      "dropout / batch-statistics mode), or enable gradients for a training pass")
        args = self.args
        prec = self.hip_prec()
        raw1, raw2 = image1.float().contiguous(), image2.float().contiguous()
        B, _, H, W = image1.shape
        if H % 8 or W % 8:
            raise ValueError("image height and width must be multiples of 8 (use InputPadder)")
        H8, W8, N = H // 8, W // 8, (H // 8) * (W // 8)
        if min(H8, W8) < 2 ** (args.corr_levels - 1):
            raise ValueError(f"image {H}x{W} is too small: level {args.corr_levels - 1} of the correlation pyramid would be empty "
                             "(the reference's avg_pool2d fails the same way)")
        hw = (H8, W8)
        dev = image1.device

        with torch.no_grad(), _JoinOnError() as joins:
            use_henc = getattr(args, "hip_encoders", True)
            # The context chain (cnet -> net / inp -> intra-frame attention -> GRU context fields) and the feature
            # chain (fnet -> F2 transformer -> correlation volume) are independent until the refinement loop: the
            # context chain is enqueued on a side stream (fork / join by events; every side-stream tensor is consumed
            # on the main stream only after the join, and the side stream always starts by waiting for the main one,
            # so the caching allocator's per-stream reuse stays ordered).
            main = torch.cuda.current_stream()
            fork = use_henc and getattr(args, "hip_fork", True) and not os.environ.get("CRAFT_NO_FORK")
            side = self._streams(1, dev)[0] if fork else main
            if side is not main:
                joins.pairs.append((main, side))
            hx = torch.empty(B, N, 512, device=dev, dtype=torch.float32)              # [net | inp | mf | mfg]
            if use_henc:
                # CNN encoders on the HIP conv engine, channels-last end to end
                if side is not main:
                    side.wait_stream(main)
                with torch.cuda.stream(side):
                    cn_tok = self._henc_c.forward_tokens(raw1, prec)                            # [B, N, 256]
                    attention, gru_fields = self._context_chain(cn_tok, hx, hw, prec)
                fm = self._henc_f.forward_tokens((raw1, raw2), prec)                           # [2B, N, 256] (both frames, one batch)
                f1_tok, f2_tok = fm[:B], fm[B:]
            else:
                image1 = (2 * (raw1 / 255.0) - 1.0).contiguous()
                image2 = (2 * (raw2 / 255.0) - 1.0).contiguous()
                with _autocast(getattr(args, "encoder_autocast", False)):
                    fmap1, fmap2 = self.fnet([image1, image2])
                    cnet_feat = self.cnet(image1)
                f1_tok = ops.tokens_from_nchw(fmap1.float())
                f2_tok = ops.tokens_from_nchw(fmap2.float())
                cn_tok = ops.tokens_from_nchw(cnet_feat.float())
                attention, gru_fields = self._context_chain(cn_tok, hx, hw, prec)

            # ---- F2 transformer (network.py:185-187): tokens in, LayerNorm-ed tokens out ------------
            x2 = self.f2_trans.vispos_encoder.ln_tokens(f2_tok, hw)
            fmap2_t = self.f2_trans.forward_tokens(x2, hw, prec=prec)                 # [B, N, 256]

            # ---- correlation volume + pyramid (network.py:196-197, :225-228) ---------------------
            if args.craft:
                # the inter-frame encoder's tokens (corr.py:148-160): frame 1 at coords1 (= the grid, or grid + flow_init), frame 2 at the grid
                venc = self.corr_fn.vispos_encoder
                pos1 = None
                if flow_init is not None and venc.pos_code_type != "bias":
                    ys, xs = torch.meshgrid(torch.arange(H8, device=dev), torch.arange(W8, device=dev), indexing="ij")
                    grid = torch.stack([ys, xs], dim=-1).reshape(1, N, 2).float()
                    pos1 = grid + ops.tokens_from_nchw(flow_init.float()).flip(-1)          # (y, x)
                # (the encoder keeps the reference's eval-mode code cache, setrans.py:744-758: the FIRST call of a shape decides the code
                # of all later ones -- frame 2 of this pair and the next pairs of a warm-started sequence; see ln_tokens)
                x1 = venc.ln_tokens(f1_tok, hw, pos1)
                x2t = venc.ln_tokens(fmap2_t, hw)
                if self.f1_trans is not None:       # two-way: (transformed 1, conv 2) and (conv 1, transformed 2)
                    fmap1_t = self.f1_trans.forward_tokens(self.f1_trans.vispos_encoder.ln_tokens(f1_tok, hw), hw, prec=prec)
                    self.corr_fn.update_tokens(venc.ln_tokens(fmap1_t, hw, pos1), x2t, hw, prec, x1, venc.ln_tokens(f2_tok, hw))
                else:
                    self.corr_fn.update_tokens(x1, x2t, hw, prec)
                corr_fn = self.corr_fn
            else:
                corr_fn = CorrBlock.__new__(CorrBlock)
                corr_fn.num_levels, corr_fn.radius, corr_fn.shape = 4, args.corr_radius, (B, H8, W8)
                corr_fn.pyramid = ops.CorrPyramid(B, H8, W8, 4, dev)
                import math
                ops.corr_build(f1_tok.contiguous(), fmap2_t, H8, W8, 1, 1.0 / math.sqrt(256), None, 0.0, 1.0, None,
                               corr_fn.pyramid, False, prec)
                self.corr_fn = corr_fn

            if side is not main:
                main.wait_stream(side)
            coords0, coords1, flow = ops.coords_init(flow_init, B, H8, W8, dev)
            pyramids = corr_fn.all_pyramids()
            nch = corr_fn.num_levels * (2 * corr_fn.radius + 1) ** 2 * len(pyramids)
            corr = torch.empty(B, N, nch, device=dev, dtype=torch.float32)
            mask = torch.empty(B, N, 576, device=dev, dtype=torch.float32)
            need_all = test_mode != 1
            radius = corr_fn.radius
            n_pred = iters if need_all else min(iters, 1)
            flow_ups = [torch.empty(B, 2, H, W, device=dev, dtype=torch.float32) for _ in range(n_pred)]

            # ---- iterative refinement (network.py:230-260).  Every per-sample tensor is batch-major, so the loop of a
            # batch slice touches only its own rows: the batch can be cut into `hip_streams` slices whose loops are enqueued
            # interleaved on separate HIP streams.  No kernel of the loop couples samples, so the results are the same; the
            # HBM-bound kernels (P.V, lookup, conv epilogues) of one slice overlap the MFMA-bound convolutions of the other.
            # Measured pairs/s (1 stream / 2 streams): batch 4: 206 / 200 (half-size kernels lose more than the overlap
            # wins), batch 6: 208 / 216, batch 8: 216 / 223 (4 streams: 211), batch 16: 225 / 225 -- hence the auto rule.
            nstr = int(os.environ.get("CRAFT_HIP_STREAMS", getattr(args, "hip_streams", 0)))
            if nstr <= 0:                      # auto: two slices of 3..6 samples each (measured below); else one stream
                nstr = 2 if 6 <= B <= 12 else 1
            if torch.cuda.is_current_stream_capturing():
                # under a hipGraph capture (GraphedForward) the batch stays in one piece: the graph already carries the dependency
                # structure, and hipStreamEndCapture of a pass with two sliced loops at 448x1024 x 4 crashed inside the runtime (round 6)
                nstr = 1
            nstr = max(1, min(nstr, B))
            cuts = [B * i // nstr for i in range(nstr + 1)]
            parts = []
            for i in range(nstr):
                b0, b1 = cuts[i], cuts[i + 1]
                parts.append(dict(b=(b0, b1), hx=hx[b0:b1], corr=corr[b0:b1], flow=flow[b0:b1], att=ops.probs_slice(attention, b0, b1),
                                  c0=coords0[b0:b1], c1=coords1[b0:b1], mask=mask[b0:b1], fields=gru_fields[b0:b1],
                                  pyr=[pv.batch_slice(b0, b1) for pv in pyramids], ws=None))
            streams = [main] if nstr == 1 else self._streams(nstr, dev)
            if nstr > 1:
                joins.pairs.extend((main, st) for st in streams)
            if nstr > 1:
                # (wait_stream, not hand-made Events: an Event object created here dies while a hipGraph capture of this pass is still
                # recording -- hipEventDestroy under capture took the process down in GraphedForward with two slices)
                for st in streams:
                    st.wait_stream(main)
            for pt, st in zip(parts, streams):
                with torch.cuda.stream(st):
                    pt["ws"] = GMAUpdateBlock.workspace(pt["b"][1] - pt["b"][0], N, dev)
            self.update_block.encoder.begin_pass()
            for itr in range(iters):
                for pt, st in zip(parts, streams):
                    with torch.cuda.stream(st):
                        self.update_block.encoder.prefork(dev)                                 # (the flow branch may start beside the lookup)
                        ops.corr_lookup(pt["pyr"], pt["c1"], radius, out=pt["corr"])           # network.py:235
                        self.update_block.step_tokens(pt["hx"], pt["corr"], pt["flow"], pt["att"], hw, pt["ws"], prec,
                                                      pt["fields"])                            # :244 (to the new net)
                        self.update_block.flow_head_tokens(pt["hx"], hw, pt["c1"], pt["c0"], pt["flow"], None, pt["ws"], prec)
                        # the mask head + convex upsampling only feed the returned predictions: in test_mode=1 only
                        # the last one is returned (network.py:262-263), so earlier ones are skipped (same result).
                        if need_all or itr == iters - 1:
                            self.update_block.mask_tokens(pt["hx"], hw, pt["ws"], prec, out=pt["mask"])
                            b0, b1 = pt["b"]
                            ops.convex_upsample(pt["mask"], pt["flow"], H8, W8, out=flow_ups[itr if need_all else 0][b0:b1])  # :258
            if nstr > 1:
                for st in streams:
                    main.wait_stream(st)
            flow_predictions = flow_ups
            flow_up = flow_ups[-1] if flow_ups else None
            flow_lo = ops.tokens_to_nchw(flow, H8, W8)

        self.call_counter += 1
        if test_mode == 1:
            return flow_lo, flow_up
        if test_mode == 2:
            return flow_lo, flow_predictions
        return flow_predictions
