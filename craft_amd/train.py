"""Training-step pieces around the hot path.

What exists: the loss (``sequence_loss``, train.py:44-73) with its gradient w.r.t. the predictions, the OneCycle schedule
of ``fetch_optimizer`` (train.py:76-85), a fused AdamW with global-norm clipping over ONE flat parameter / gradient buffer
(train.py:234 + torch.optim.AdamW semantics), the gradient exchange of train_ddp.py (one all-reduce of that flat buffer per
step: RCCL over xGMI under the "nccl" backend, gloo in the CPU tests), and checkpoint save / resume in the reference's
layout (train.py:132-175).  What does NOT exist yet: the backward kernels of the model itself -- so the gradients these
pieces consume are whatever the caller provides, and ``CRAFT.forward`` still refuses to run in training mode.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional

import torch

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
                 weight_decay: float = 1e-2):
        self.params = list(params)
        if not self.params:
            raise ValueError("FlatAdamW: no parameters")
        if any(not p.requires_grad for p in self.params):
            # the state dict is indexed like torch.optim.AdamW(model.parameters()) (train.py:78): a frozen parameter would
            # shift every later index, and the reference trainers freeze nothing
            raise ValueError("FlatAdamW: every parameter must require grad (state-dict indices follow model.parameters())")
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.empty(self.numel, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        off = 0
        for p in self.params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view_as(p.data)
            p.grad = self.flat_grad[off:off + n].view_as(p.data)
            off += n
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self._sumsq = torch.zeros((), device=dev, dtype=torch.float64)

    def zero_grad(self):
        self.flat_grad.zero_()

    def allreduce_grads(self, group=None) -> float:
        """Sum the flat gradient over the data-parallel group (one collective); returns the 1/world factor that
        ``step`` folds into the update (train_ddp.py's DDP averages the gradients)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return 1.0
        dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=group)
        return 1.0 / dist.get_world_size(group)

    def step(self, lr: Optional[float] = None, max_norm: float = 0.0, grad_mul: float = 1.0):
        """One update; ``max_norm`` > 0 applies clip_grad_norm_(params, max_norm) (train.py:234) without a host sync."""
        self._check_views()
        self.step_count += 1
        sumsq = None
        if max_norm > 0:
            self._sumsq.zero_()
            call("craft_sumsq", self.flat_grad, self.numel, self._sumsq)
            sumsq = self._sumsq
        call("craft_adamw_step", self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.numel,
             float(self.lr if lr is None else lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
             float(self.weight_decay), self.step_count, float(grad_mul), sumsq, float(max_norm))
        # the kernel wrote through raw pointers: p._version / data_ptr() did not move, so tell the packed-weight caches
        from .hip import bump_weights_epoch
        bump_weights_epoch()

    def _check_views(self):
        """model.zero_grad(set_to_none=True), model.to(...) or p.grad = ... silently detach parameters / gradients from the
        flat buffers; the fused kernel would then update stale memory.  Cheap host check (145 pointer compares)."""
        off = 0
        for p in self.params:
            n = p.numel()
            if p.data_ptr() != self.flat.data_ptr() + 4 * off:
                raise RuntimeError("FlatAdamW: a parameter no longer lives in the flat buffer (model.to() / load with "
                                   "assign=True after the optimizer was built?)")
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                raise RuntimeError("FlatAdamW: a .grad no longer aliases the flat gradient buffer "
                                   "(use optimizer.zero_grad(), not model.zero_grad(set_to_none=True))")
            off += n

    # ---- torch.optim.AdamW-compatible state (the 'optimizer' entry of the reference's checkpoints, train.py:139)
    def state_dict(self) -> Dict:
        state, off = {}, 0
        for i, p in enumerate(self.params):
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[off:off + n].view_as(p.data).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + n].view_as(p.data).clone()}
            off += n
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: Dict):
        off = 0
        for i, p in enumerate(self.params):
            n = p.numel()
            st = sd["state"].get(i)
            if st is not None:
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                self.step_count = int(float(st["step"]))
            off += n
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.weight_decay = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]


def fetch_optimizer(model: torch.nn.Module, lr: float, wdecay: float, epsilon: float, num_steps: int):
    """train.py:76-85: AdamW(lr, weight_decay, eps) + OneCycleLR(max_lr=lr, total_steps=num_steps+100, pct_start=0.05,
    linear anneal, no momentum cycling)."""
    opt = FlatAdamW(model.parameters(), lr=lr, weight_decay=wdecay, eps=epsilon)
    return opt, OneCycleLR(lr, num_steps + 100, pct_start=0.05)


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
    --loadopt / --loadsched)."""
    from .utils import load_checkpoint as load_model, read_checkpoint
    ck = read_checkpoint(path, trusted=trusted)
    msg = load_model(model, ck)
    if optimizer is not None:                      # load_state_dict re-pointed nothing: the flat views stay valid, but refresh
        off = 0                                    # the flat copy in case a parameter was replaced rather than copied into
        for p in optimizer.params:
            n = p.numel()
            if p.data.data_ptr() != optimizer.flat[off:off + n].data_ptr():
                optimizer.flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = optimizer.flat[off:off + n].view_as(p.data)
            off += n
    if load_optimizer_state and optimizer is not None and isinstance(ck, dict) and "optimizer" in ck:
        optimizer.load_state_dict(ck["optimizer"])
    logger = None
    if load_scheduler_state and lr_scheduler is not None and isinstance(ck, dict) and "lr_scheduler" in ck:
        lr_scheduler.load_state_dict(ck["lr_scheduler"])
        logger = ck.get("logger")
    return msg, logger
