"""Training-step pieces around the hot path.

The loss (``sequence_loss``, train.py:44-73) with its gradient w.r.t. the predictions, the OneCycle schedule of
``fetch_optimizer`` (train.py:76-85), a fused AdamW with global-norm clipping over ONE flat parameter / gradient buffer
(train.py:234 + torch.optim.AdamW semantics), the gradient exchange of train_ddp.py (one all-reduce of that flat buffer per
step: RCCL over xGMI under the "nccl" backend, gloo in the tests), checkpoint save / resume in the reference's layout
(train.py:132-175), and ``Trainer.step`` = one iteration of train.py:215-236 / train_ddp.py:230-262: zero_grad -> forward
(model.train(): craft_amd/train_forward.py) -> sequence_loss -> backward (HIP kernels, craft_amd/autograd.py) -> all-reduce ->
clip -> AdamW -> scheduler.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Iterable, List, Optional

import torch

from .dist import collective_group_active
from .hip import call

MAX_FLOW = 400.0            # train.py:30


# ------------------------------------------------------------------------------------------------
# loss
# ------------------------------------------------------------------------------------------------
def sequence_loss(flow_preds: List[torch.Tensor], flow_gt: torch.Tensor, valid: torch.Tensor, gamma: float = 0.8,
                  max_flow: float = MAX_FLOW, want_grad: bool = False):
    """train.py:44-73 on the device: sum_i gamma^(T-1-i) * mean(valid * |pred_i - gt|) and the EPE / 1-3-5 px metrics of
    the last prediction -> (loss [0-dim float64 tensor], metrics dict, grads or None).  ``grads[i]`` = d loss / d pred_i."""
    from .evaluate import FlowMetrics
    T = len(flow_preds)
    B, _, H, W = flow_gt.shape
    dev = flow_preds[0].device
    gt = flow_gt.to(dev).float().contiguous()
    va = valid.to(dev).float().contiguous()
    loss = torch.zeros((), device=dev, dtype=torch.float64)
    grads = [] if want_grad else None
    for i, p in enumerate(flow_preds):
        w = gamma ** (T - i - 1)
        g = torch.empty_like(gt) if want_grad else None
        call("craft_flow_l1_loss", p.float().contiguous(), gt, va, B, H, W, float(w), float(max_flow), loss, g)
        if want_grad:
            grads.append(g)
    m = FlowMetrics(dev, max_mag=max_flow)
    m.update(flow_preds[-1], gt, va)
    r = m.result()
    return loss, {"epe": r["epe"], "1px": r["px1"], "3px": r["px3"], "5px": r["px5"]}, grads


# ------------------------------------------------------------------------------------------------
# learning-rate schedule
# ------------------------------------------------------------------------------------------------
class OneCycleLR:
    """optim.lr_scheduler.OneCycleLR(max_lr, total_steps, pct_start, cycle_momentum=False, anneal_strategy='linear') as
    fetch_optimizer builds it (train.py:81-83; torch defaults div_factor=25, final_div_factor=1e4, two phases)."""

    def __init__(self, max_lr: float, total_steps: int, pct_start: float = 0.05, div_factor: float = 25.0,
                 final_div_factor: float = 1e4):
        self.max_lr, self.total_steps, self.pct_start = max_lr, total_steps, pct_start
        self.initial_lr = max_lr / div_factor
        self.min_lr = self.initial_lr / final_div_factor
        self.last_epoch = 0

    def lr_at(self, step: int) -> float:
        if step > self.total_steps:
            raise ValueError(f"OneCycleLR: step {step} beyond total_steps {self.total_steps}")
        end1 = float(self.pct_start * self.total_steps) - 1.0
        end2 = float(self.total_steps) - 1.0
        if step <= end1:
            pct = step / end1
            return (self.max_lr - self.initial_lr) * pct + self.initial_lr
        pct = (step - end1) / (end2 - end1)
        return (self.min_lr - self.max_lr) * pct + self.max_lr

    def get_last_lr(self):
        return [self.lr_at(self.last_epoch)]

    def step(self):
        self.last_epoch += 1

    def state_dict(self) -> Dict:
        return {"total_steps": self.total_steps, "last_epoch": self.last_epoch, "max_lr": self.max_lr, "pct_start": self.pct_start,
                "_last_lr": self.get_last_lr()}

    def load_state_dict(self, sd: Dict):
        self.last_epoch = int(sd["last_epoch"])
        self.total_steps = int(sd.get("total_steps", self.total_steps))


# ------------------------------------------------------------------------------------------------
# flat parameters + fused AdamW
# ------------------------------------------------------------------------------------------------
class FlatAdamW:
    """AdamW (torch.optim.AdamW update rule) over one flat fp32 buffer.

    The parameters of ``params`` are re-pointed into ``self.flat`` (views, same values) and their ``.grad`` into
    ``self.flat_grad``, so (a) the optimizer is one kernel over 6.3 M elements instead of 145 small ones, (b) the
    data-parallel exchange is ONE all-reduce of a 25 MB buffer, and (c) gradient clipping is one reduction.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, unused: Iterable[torch.nn.Parameter] = ()):
        """``unused``: parameters that never receive a gradient (the reference's DDP runs with find_unused_parameters=True,
        train_ddp.py:196-198; their .grad stays None there, so torch.optim.AdamW skips them entirely -- no weight decay, no
        state).  They stay in the flat buffers (index layout of model.parameters()) but the update kernel skips their ranges."""
        self.params = list(params)
        unused_ids = {id(p) for p in unused}
        if not self.params:
            raise ValueError("FlatAdamW: no parameters")
        if any(not p.requires_grad for p in self.params):
            # the state dict is indexed like torch.optim.AdamW(model.parameters()) (train.py:78): a frozen parameter would
            # shift every later index, and the reference trainers freeze nothing
            raise ValueError("FlatAdamW: every parameter must require grad (state-dict indices follow model.parameters())")
        dev = self.params[0].device
        # every parameter starts on a 128-byte boundary of the flat buffers (the kernels read weights with 16-byte vector
        # loads; the padding elements stay zero: zero gradient, zero value, no effect on the norm or the update)
        ALIGN = 32
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off                                   # elements of the flat buffers (padding included)
        self.n_params = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.unused_index, segs, start = [], [], 0
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view_as(p.data)
            p.grad = self.flat_grad[off:off + n].view_as(p.data)
            if id(p) in unused_ids:
                self.unused_index.append(i)
                if off > start:
                    segs.append((start, off))
                start = off + n
        if self.numel > start:
            segs.append((start, self.numel))
        self.segments = segs                     # [begin, end) ranges of the flat buffers that the optimizer updates
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self._sumsq = torch.zeros((), device=dev, dtype=torch.float64)
        # GradScaler's state on the device (include/craft_hip.h, craft_loss_scale_update): 8 x 32-bit words
        # {scale f32, growth_tracker, opt_step (APPLIED updates), skipped, found_inf, gm f32, bc1 f32, bc2_sqrt f32}
        self._scaler = torch.zeros(8, device=dev, dtype=torch.int32)
        self._scaler_f = self._scaler.view(torch.float32)
        self._scaler_f[0] = 1.0
        self.growth, self.backoff, self.growth_interval = 2.0, 0.5, 0          # interval 0: static scale (set_loss_scale)
        self._host_steps = 0                 # step() calls (applied + skipped); the applied count lives on the device
        self._snap = torch.zeros(8, dtype=torch.int32).pin_memory() if dev.type == "cuda" else torch.zeros(8, dtype=torch.int32)
        self._snap_event = None

    def zero_grad(self):
        self.flat_grad.zero_()

    def load_grads(self, grads):
        """The gradients of ``self.params`` (``torch.autograd.grad`` order; None = no gradient) into the flat buffer in ONE multi-tensor
        copy -- instead of ``zero_grad()`` + ``backward()``, whose AccumulateGrad nodes add every gradient into its (zeroed) view with a
        kernel of its own (127 launches of a configs[3] step).  A view whose parameter had a gradient before and has none now is zeroed;
        the alignment padding between the views stays zero from the allocation."""
        views = self.grad_views()
        had = self.__dict__.setdefault("_had_grad", [False] * len(views))
        dst, src, offs, stale = [], [], [], []
        for i, (v, g) in enumerate(zip(views, grads)):
            if g is None:
                if had[i]:
                    stale.append(v)
                had[i] = False
            else:
                if g.shape != v.shape:
                    raise ValueError(f"gradient {i} has shape {tuple(g.shape)}, its parameter {tuple(v.shape)}")
                dst.append(v)
                src.append(g.detach())
                offs.append(self.offsets[i])
                had[i] = True
        with torch.no_grad():
            if dst and self.flat_grad.is_cuda and all(g.dtype == torch.float32 for g in src):
                # ONE launch (craft_multi_copy); torch._foreach_copy_ issues a copyBuffer per tensor on this build (131 per configs[3] step)
                import ctypes
                from .hip import call, carray
                # conv weight gradients arrive as the permuted view of the [cout][KH][KW][cin] buffer the weight-gradient kernels write: the
                # launch transposes them on the way (a .contiguous() per weight was 45 clone launches per configs[3] step)
                cl = [0] * len(src)
                for i, g in enumerate(src):
                    if not g.is_contiguous():
                        if g.dim() == 4 and g.permute(0, 2, 3, 1).is_contiguous() and g.shape[2] * g.shape[3] < 1024 and g.shape[1] < (1 << 21):
                            cl[i] = g.shape[1] * 1024 + g.shape[2] * g.shape[3]
                        else:
                            src[i] = g.contiguous()
                call("craft_multi_copy", carray(ctypes.c_void_p, [g.data_ptr() for g in src]), carray(ctypes.c_long, [g.numel() for g in src]),
                     carray(ctypes.c_long, offs), carray(ctypes.c_long, cl), len(src), self.flat_grad)
            elif dst:
                torch._foreach_copy_(dst, src)
            if stale:
                torch._foreach_zero_(stale)

    def grad_views(self):
        v = self.__dict__.get("_grad_views")
        if v is None:
            v = self._grad_views = [self.flat_grad[off:off + p.numel()].view_as(p.data) for p, off in zip(self.params, self.offsets)]
        return v

    def allreduce_grads(self, group=None) -> float:
        """Sum the flat gradient over the data-parallel group (one collective); returns the 1/world factor that
        ``step`` folds into the update (train_ddp.py's DDP averages the gradients)."""
        import torch.distributed as dist
        if not collective_group_active(group):
            return 1.0
        if dist.get_backend(group) == "gloo" and self.flat_grad.is_cuda:
            # test configuration (several ranks sharing one GPU over gloo): stage through the host; RCCL ("nccl") reduces in place
            host = self.flat_grad.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            self.flat_grad.copy_(host)
        else:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=group)
        return 1.0 / dist.get_world_size(group)

    def broadcast_state(self, src: int = 0, group=None):
        """Make every rank's weights and optimizer state rank ``src``'s (what torch DDP does for parameters at construction,
        train_ddp.py:196-200; the optimizer state matters after a resume that only one rank loaded): four collectives over the
        flat buffers."""
        import torch.distributed as dist
        if not collective_group_active(group):
            return
        for t in (self.flat, self.exp_avg, self.exp_avg_sq, self._scaler):      # (the scaler record carries the applied-step count)
            _broadcast(t, src, group)
        from .hip import bump_weights_epoch
        bump_weights_epoch()

    # ---- loss scaling (torch.cuda.amp.GradScaler, train.py:215, 231-238) -- all of it on the device, no host read-back on the step
    def set_loss_scale(self, scale: float, dynamic: bool = False, growth: float = 2.0, backoff: float = 0.5, growth_interval: int = 2000):
        """``scale``: what the caller multiplies the loss gradient by (``scale_seed()``).  dynamic: GradScaler's rule -- an overflow
        halves it, ``growth_interval`` clean steps double it; static: the scale stays, an overflow still skips (and counts) the step."""
        self._scaler_f[0] = float(scale)
        self._scaler[1] = 0
        self.growth, self.backoff = (float(growth), float(backoff)) if dynamic else (1.0, 1.0)
        self.growth_interval = int(growth_interval) if dynamic else 0

    def scale_seed(self) -> torch.Tensor:
        """The current loss scale as a 0-dim device tensor: ``loss.backward(opt.scale_seed())``.  A copy made in stream order, so
        the update kernel of the same step can change the record afterwards."""
        return self._scaler_f[0].clone()

    @property
    def step_count(self) -> int:
        """APPLIED updates (torch.optim.AdamW's state['step']): skipped steps do not count.  Reads the device record (one sync):
        for checkpoints and tests, not for the step."""
        return int(self._scaler[2].item())

    @step_count.setter
    def step_count(self, v: int):
        self._scaler[2] = int(v)

    def scaler_snapshot(self, wait: bool = False) -> Optional[Dict[str, float]]:
        """{'loss_scale', 'applied_steps', 'skipped_steps', 'found_inf'} of the most recent step whose asynchronous read-back has
        landed (None before the first).  wait=True blocks on the last step's copy."""
        ev = self._snap_event
        if ev is None:
            return None
        if wait:
            ev.synchronize()
        elif not ev.query():
            return getattr(self, "_snap_last", None)
        w = self._snap.clone()
        self._snap_last = {"loss_scale": float(w.view(torch.float32)[0]), "applied_steps": int(w[2]), "skipped_steps": int(w[3]),
                           "found_inf": int(w[4])}
        return self._snap_last

    def step(self, lr: Optional[float] = None, max_norm: float = 0.0, grad_mul: float = 1.0):
        """One update; ``max_norm`` > 0 applies clip_grad_norm_(params, max_norm) (train.py:234) without a host sync.  The flat
        gradient is expected to hold (loss scale) x the gradient (``set_loss_scale`` / ``scale_seed``; scale 1 by default) -- the
        un-scaling, the 1/world of a summed all-reduce (``grad_mul``) and the clip coefficient are one multiplier computed on the
        device.  A non-finite gradient norm -- with or without clipping -- skips the update: weights, moments and the applied-step
        count (bias correction) stay, the skip is counted, a dynamic scale backs off."""
        self._check_views()
        self._host_steps += 1
        self._sumsq.zero_()
        call("craft_sumsq", self.flat_grad, self.numel, self._sumsq)
        call("craft_loss_scale_update", self._sumsq, self._scaler, float(grad_mul), float(max_norm), float(self.betas[0]), float(self.betas[1]),
             float(self.growth), float(self.backoff), int(self.growth_interval))
        for a, b in self.segments:
            call("craft_adamw_step_dyn", self.flat[a:b], self.flat_grad[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b], b - a,
                 float(self.lr if lr is None else lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                 float(self.weight_decay), self._scaler)
        if self.flat.is_cuda:              # asynchronous read-back of the record (pinned buffer + event): metrics / logs, never the step
            self._snap.copy_(self._scaler, non_blocking=True)
            self._snap_event = torch.cuda.Event()
            self._snap_event.record()
        # the kernel wrote through raw pointers: p._version / data_ptr() did not move, so tell the packed-weight caches
        from .hip import bump_weights_epoch
        bump_weights_epoch()

    def _check_views(self):
        """model.zero_grad(set_to_none=True), model.to(...) or p.grad = ... silently detach parameters / gradients from the
        flat buffers; the fused kernel would then update stale memory.  Cheap host check (145 pointer compares)."""
        for p, off in zip(self.params, self.offsets):
            if p.data_ptr() != self.flat.data_ptr() + 4 * off:
                raise RuntimeError("FlatAdamW: a parameter no longer lives in the flat buffer (model.to() / load with "
                                   "assign=True after the optimizer was built?)")
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                raise RuntimeError("FlatAdamW: a .grad no longer aliases the flat gradient buffer "
                                   "(use optimizer.zero_grad(), not model.zero_grad(set_to_none=True))")

    # ---- torch.optim.AdamW-compatible state (the 'optimizer' entry of the reference's checkpoints, train.py:139)
    def state_dict(self) -> Dict:
        state = {}
        applied = float(self.step_count)
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            if i in self.unused_index:          # torch.optim.AdamW keeps no state for a parameter without a gradient
                continue
            state[i] = {"step": torch.tensor(applied),
                        "exp_avg": self.exp_avg[off:off + n].view_as(p.data).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + n].view_as(p.data).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: Dict):
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            st = sd["state"].get(i)
            if st is not None:
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                self.step_count = int(float(st["step"]))
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.weight_decay = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]


def _broadcast(t: torch.Tensor, src: int, group=None):
    """dist.broadcast in place; under gloo a device tensor is staged through the host (test configuration: several ranks share
    one GPU), RCCL ("nccl") broadcasts device memory directly; a host tensor under nccl travels through the device."""
    import torch.distributed as dist
    backend = dist.get_backend(group)
    if backend == "gloo" and t.is_cuda:
        host = t.cpu()
        dist.broadcast(host, src, group=group)
        t.copy_(host)
    elif backend == "nccl" and not t.is_cuda:
        d = t.cuda()
        dist.broadcast(d, src, group=group)
        t.copy_(d.cpu())
    else:
        dist.broadcast(t, src, group=group)


def unused_parameters(model: torch.nn.Module) -> List[torch.nn.Parameter]:
    """Parameters that exist for state-dict parity but never enter the forward pass: the attn_softaggr of an attention that
    returns probabilities (the intra-frame attention ``att``: setrans.py:455-458 creates it whenever num_modes > 1, the
    out_attn_probs_only branch never calls it).  The reference capture (tests/golden/train_*.npz 'unused') lists the same."""
    out = []
    att = getattr(model, "att", None)
    st = getattr(att, "setrans", None)
    if st is not None and getattr(st, "out_attn_probs_only", False) and hasattr(st, "attn_softaggr"):
        out += list(st.attn_softaggr.parameters())
    # gma.Attention: the relative-position embeddings only enter the scores with --position_only / --position_and_content
    # (gma.py:84-98); content-only attention (the default) leaves them without a gradient
    args = getattr(model, "args", None)
    pos = getattr(att, "pos_emb", None)
    if pos is not None and not (getattr(args, "position_only", False) or getattr(args, "position_and_content", False)):
        out += list(pos.parameters())
    return out


def fetch_optimizer(model: torch.nn.Module, lr: float, wdecay: float, epsilon: float, num_steps: int):
    """train.py:76-85: AdamW(lr, weight_decay, eps) + OneCycleLR(max_lr=lr, total_steps=num_steps+100, pct_start=0.05,
    linear anneal, no momentum cycling)."""
    opt = FlatAdamW(model.parameters(), lr=lr, weight_decay=wdecay, eps=epsilon, unused=unused_parameters(model))
    return opt, OneCycleLR(lr, num_steps + 100, pct_start=0.05)


def auto_loss_scale(n_loss_elements: int) -> float:
    """Power-of-two loss scale that centres the backward pass on fp16's exponent range.  ``sequence_loss`` is a MEAN over B*2*H*W
    elements (train.py:58): its gradient per element is 1 / (B*2*H*W) -- 3.4e-7 at 368x496 with batch 8, BELOW fp16's smallest normal
    number (6.1e-5) and within a few of its subnormal steps (6e-8).  The 16-bit MFMA operand modes of the backward contractions
    (fp16, and f16x3, whose hi / lo planes are fp16 too) would round such operands to ~10 % (measured: 33 % relative L2 error of the
    correlation path's weight gradients at 368x496, 12 iterations; the same step with this scale: fp32-class).  The reference trains
    with fp16 autocast under torch.cuda.amp.GradScaler for the same reason (train.py:215, :231-238).
    Scale = 2^(ceil(log2(N)) - 6): the initial gradient g0 becomes 2^-7 .. 2^-6.  Measured over the backward operators of configs[3] /
    configs[4] steps (tools/grad_range.py, profiles/r3/grad_range_*.txt): the gradient tensors' RMS spans 0.008 g0 .. 1 100 g0, their
    largest element 1.1e5 g0 -- scaled, RMS >= 1.2e-4 (above fp16's smallest normal 6.1e-5) and max <= 1 800 (36 x below fp16's largest
    65 504; with g0 scaled to ~1 the context encoder's first layers overflowed).  A power of two scales exactly; the un-scaling is folded
    into the optimizer's gradient multiplier, and a gradient that overflows anyway (non-finite norm) skips the update instead of
    poisoning the weights."""
    return float(2 ** max(0, math.ceil(math.log2(max(1, int(n_loss_elements)))) - 6))


class Trainer:
    """One process of train.py / train_ddp.py around a ``craft_amd.CRAFT`` on one GPU.

    ``step`` = train.py:215-236: optional input noise (:220-223), ``optimizer.zero_grad()``, forward in training mode, the
    sequence loss, backward, gradient clipping at ``clip``, AdamW, OneCycle.  Data-parallel (train_ddp.py:187-200): every rank
    holds a full replica and its own pairs; the ONLY data-path collective is one all-reduce (sum) of the flat 25 MB gradient
    buffer per step -- RCCL over xGMI with the "nccl" backend -- folded into the update as a 1/world factor (DDP's gradient
    averaging), plus a 2-double all-reduce for the logged loss / EPE.  ``reference_loss_scaling``: train_ddp.py:60,84-88
    back-propagates the all-reduced loss divided by the world size, so its gradients carry a second 1/world on top of DDP's
    average; True reproduces that, False (default) is the plain data-parallel mean.  ``loss_scale``: "auto"
    (default: DYNAMIC, GradScaler's rule -- start at ``auto_loss_scale`` of the loss' element count, halve on an overflow, double after
    2 000 clean steps), a number (static), or None / 1.0 (off) -- the counterpart of train.py's GradScaler: the loss gradient is
    multiplied by the scale before the backward pass and the gradients are un-scaled inside the fused AdamW (clipping sees the
    un-scaled norm).  A non-finite gradient skips the update exactly like GradScaler.step: weights, moments and the optimizer's step
    count stay (the LR schedule advances, as in train.py:236-237), the skip is counted (``metrics['skipped_steps']``) and a dynamic
    scale backs off.  The scale, the counters and the decision live in a device record (FlatAdamW._scaler); the host reads them back
    asynchronously for the metrics only."""

    def __init__(self, model: torch.nn.Module, lr: float = 4e-4, wdecay: float = 1e-4, epsilon: float = 1e-8, num_steps: int = 100000,
                 clip: float = 1.0, gamma: float = 0.8, iters: int = 12, add_noise: bool = False, freeze_bn: bool = False, group=None,
                 reference_loss_scaling: bool = False, loss_scale="auto"):
        self.model = model
        self.loss_scale = loss_scale
        if getattr(model, "args", None) is not None:      # fp16 operand roles are legal under the loss scale (train_forward.training_precision)
            model.args.hip_loss_scaled = loss_scale == "auto" or bool(loss_scale and float(loss_scale) > 1.0)
        self.optimizer, self.scheduler = fetch_optimizer(model, lr, wdecay, epsilon, num_steps)
        self.clip, self.gamma, self.iters, self.add_noise, self.freeze_bn, self.group = clip, gamma, iters, add_noise, freeze_bn, group
        self.reference_loss_scaling = reference_loss_scaling
        self.total_steps = 0
        self._seed = None                   # the loss scale (0-dim device tensor) the last step's backward ran under; None before the first step
        from .ops import WeightPackRegistry
        # conv-weight operands (forward and transposed forms): ONE re-pack launch per step, right behind the optimizer update
        self._weight_packs = None if os.environ.get("CRAFT_NO_PACK_REGISTRY") else WeightPackRegistry(self.optimizer.flat.untyped_storage().data_ptr())
        self._ar_events = []                # (start, end) HIP events around the gradient all-reduce of recent steps
        self._scale_set = False             # "auto" needs the loss' element count: the scaler is armed on the first step
        if loss_scale != "auto":
            self.optimizer.set_loss_scale(float(loss_scale or 1.0), dynamic=False)
            self._scale_set = True
        model.train()
        if freeze_bn:
            model.freeze_bn()
        self.sync_replicas()

    @property
    def last_loss_scale(self) -> float:
        """The loss scale the last step's backward ran under (reads the device: tests / debugging)."""
        if self._seed is None:
            raise RuntimeError("Trainer.last_loss_scale: no step has run yet")
        return float(self._seed)

    def sync_replicas(self, src: int = 0):
        """Rank ``src``'s parameters, optimizer state and module buffers on every rank.  torch DDP broadcasts parameters and buffers
        at construction and (broadcast_buffers=True, the default train_ddp.py:198-200 runs with) the buffers again before every
        forward -- cnet's BatchNorm running statistics are rank 0's everywhere.  Called here at construction; call it again after
        a checkpoint was loaded on one rank; ``sync_buffers`` alone before a validation pass (train_ddp.py's val_freq)."""
        if collective_group_active(self.group):
            self.optimizer.broadcast_state(src, self.group)
            self.sync_buffers(src)

    def sync_buffers(self, src: int = 0):
        """BatchNorm running statistics (and every other module buffer) of rank ``src`` on all ranks: DDP's broadcast_buffers.
        The training forward in batch-statistics mode does not read them, so once per validation interval is enough."""
        if collective_group_active(self.group):
            for b in self.model.buffers():
                if b.is_floating_point() or b.dtype in (torch.int64, torch.int32):
                    _broadcast(b, src, self.group)

    def allreduce_ms(self) -> Optional[float]:
        """Mean device time of the gradient all-reduce over the recorded steps (None: single process / nothing recorded)."""
        ev = [(s, e) for s, e in self._ar_events if e.query()]
        if not ev:
            return None
        return round(sum(s.elapsed_time(e) for s, e in ev) / len(ev), 4)

    def _world(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def step(self, image1, image2, flow, valid) -> Dict[str, float]:
        from .autograd import sequence_loss as seq_loss
        model, opt = self.model, self.optimizer
        dev = next(model.parameters()).device
        image1, image2 = image1.to(dev).float(), image2.to(dev).float()
        if self.add_noise:                                             # train.py:220-223
            stdv = float(torch.empty(1).uniform_(0.0, 5.0))
            image1 = (image1 + stdv * torch.randn_like(image1)).clamp(0.0, 255.0)
            image2 = (image2 + stdv * torch.randn_like(image2)).clamp(0.0, 255.0)
        if not model.training:
            model.train()
            if self.freeze_bn:
                model.freeze_bn()
        direct = not os.environ.get("CRAFT_TRAINER_BACKWARD")        # (developer A/B: 1 = zero_grad() + loss.backward() as before round 4)
        if not direct:
            opt.zero_grad()
        from . import ops as _ops
        _ops.ACTIVE_WEIGHT_PACKS[0] = self._weight_packs if image1.is_cuda else None
        try:
            return self._step_body(model, opt, seq_loss, image1, image2, flow, valid, direct, dev)
        finally:
            _ops.ACTIVE_WEIGHT_PACKS[0] = None

    def _step_body(self, model, opt, seq_loss, image1, image2, flow, valid, direct, dev):
        preds = model(image1, image2, iters=self.iters)
        loss, metrics = seq_loss(preds, flow, valid, self.gamma, defer_metrics=True)     # (read back at the end of the step: no sync in front of the backward)
        if os.environ.get("CRAFT_SYNC_METRICS"):           # developer A/B: the read-back between forward and backward, as before round 5
            r_ = metrics.resolve()
            metrics.resolve = lambda: r_
        if not self._scale_set:            # "auto": start from auto_loss_scale and let GradScaler's rule move it (train.py:215)
            opt.set_loss_scale(auto_loss_scale(flow.numel()), dynamic=True)
            self._scale_set = True
        self._seed = opt.scale_seed()                                  # the scale is read on the device, in stream order
        if direct:
            # the gradients are captured by the engine and copied into the flat buffer together (FlatAdamW.load_grads); p.grad keeps
            # pointing at its view of that buffer
            opt.load_grads(torch.autograd.grad(loss, opt.params, grad_outputs=self._seed.to(loss.dtype), allow_unused=True))
        else:
            loss.backward(self._seed.to(loss.dtype))
        from .autograd import pending_uses
        if pending_uses(model.__dict__.get("_train_pass_cache")):
            raise RuntimeError("backward left accumulated weight gradients incomplete (a layer call was pruned from the graph)")
        timed = collective_group_active(self.group) and opt.flat_grad.is_cuda
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        mul = opt.allreduce_grads(self.group)                          # ONE collective over the flat gradient buffer
        if timed:
            ev1.record()
            self._ar_events = (self._ar_events + [(ev0, ev1)])[-64:]
        if self.reference_loss_scaling:
            mul = mul / self._world()
        # (the flat gradient buffer holds loss_scale x the gradient: un-scaled on the device, craft_loss_scale_update)
        if self._weight_packs is not None and image1.is_cuda:
            self._weight_packs.prepare()                               # (host-side job table: fails HERE, before the weights move)
        opt.step(lr=self.scheduler.get_last_lr()[0], max_norm=self.clip, grad_mul=mul)
        if self._weight_packs is not None and image1.is_cuda:
            self._weight_packs.repack()                                # every conv-weight operand of the next step, one launch (71 before)
        self.scheduler.step()
        self.total_steps += 1
        if self.total_steps == 2:
            # Python's cyclic collector: its first full pass over everything the imports and the model construction allocated takes
            # ~40 ms and lands a few steps into training (measured: one 120 ms step among 78 ms ones).  Collect once now: the next full
            # pass is then due only after the long-lived set has grown by a quarter.  CRAFT_GC_FREEZE=1 additionally moves the survivors
            # to the permanent generation (gc.freeze) -- a process-wide side effect a library must not impose by default (round 5 did:
            # in a long-lived host process every object alive at that moment, e.g. an earlier model, would never be collected again).
            import gc
            gc.collect()
            if os.environ.get("CRAFT_GC_FREEZE"):
                gc.freeze()
        metrics = dict(metrics.resolve(), loss=float(loss.detach()))
        metrics["loss_rank"] = metrics["loss"]        # this rank's own loss ("loss" / "epe" become the mean over the ranks below)
        snap = opt.scaler_snapshot()       # (float(loss) above drained the stream: this is the step just taken)
        if snap is not None:
            metrics.update(loss_scale=snap["loss_scale"], skipped_steps=snap["skipped_steps"], applied_steps=snap["applied_steps"])
        if collective_group_active(self.group):                        # logged numbers: mean over ranks (train_ddp.py:84-94)
            import torch.distributed as dist
            backend = dist.get_backend(self.group)
            t = torch.tensor([metrics["loss"], metrics["epe"]], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, group=self.group)
            metrics["loss"], metrics["epe"] = float(t[0]) / self._world(), float(t[1]) / self._world()
        return metrics


# ------------------------------------------------------------------------------------------------
# checkpoints (train.py:132-175)
# ------------------------------------------------------------------------------------------------
def save_checkpoint(path: str, model: torch.nn.Module, optimizer: FlatAdamW, lr_scheduler: OneCycleLR, logger: Optional[dict] = None,
                    data_parallel_prefix: bool = True):
    """{'model', 'optimizer', 'lr_scheduler', 'logger'} with the DataParallel 'module.' key prefix the reference's
    trainers produce."""
    sd = model.state_dict()
    if data_parallel_prefix:
        sd = {"module." + k: v for k, v in sd.items()}
    torch.save({"model": sd, "optimizer": optimizer.state_dict(), "lr_scheduler": lr_scheduler.state_dict(),
                "logger": dict(logger or {})}, path)


def load_checkpoint(path: str, model: torch.nn.Module, optimizer: Optional[FlatAdamW] = None,
                    lr_scheduler: Optional[OneCycleLR] = None, load_optimizer_state: bool = False, load_scheduler_state: bool = False,
                    trusted: bool = False):
    """New dict layout or legacy bare state dict, strict=False; optimizer / scheduler only on request (train.py:147-175,
    --loadopt / --loadsched).  Data-parallel: either every rank loads the same file, or one rank does and then calls
    ``Trainer.sync_replicas(src)``."""
    from .utils import load_checkpoint as load_model, read_checkpoint
    ck = read_checkpoint(path, trusted=trusted)
    msg = load_model(model, ck)             # (bumps the weights epoch: every packed-weight cache re-packs on its next use)
    if optimizer is not None:                      # load_state_dict re-pointed nothing: the flat views stay valid, but refresh
        for p, off in zip(optimizer.params, optimizer.offsets):      # the flat copy in case a parameter was replaced
            n = p.numel()
            if p.data.data_ptr() != optimizer.flat[off:off + n].data_ptr():
                optimizer.flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = optimizer.flat[off:off + n].view_as(p.data)
    if load_optimizer_state and optimizer is not None and isinstance(ck, dict) and "optimizer" in ck:
        optimizer.load_state_dict(ck["optimizer"])
    logger = None
    if load_scheduler_state and lr_scheduler is not None and isinstance(ck, dict) and "lr_scheduler" in ck:
        lr_scheduler.load_state_dict(ck["lr_scheduler"])
        logger = ck.get("logger")
    return msg, logger
