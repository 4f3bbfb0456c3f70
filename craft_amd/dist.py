"""Multi-GPU helpers: CRAFT's path shards by image pair.

Inference needs no data-path collective: every rank holds a full replica (25 MB of weights) and
processes its own pairs; only the timing protocol is collective (barrier + max over ranks).  The
training exchange lives in craft_amd/train.py: ``Trainer`` broadcasts rank 0's parameters, optimizer state and
buffers at construction (what DDP does, train_ddp.py:196-200) and ``FlatAdamW.allreduce_grads`` is ONE RCCL
all-reduce of the flat 25.2 MB gradient buffer per step (train_ddp.py:187-200).
"""
from __future__ import annotations

import time
from typing import Callable, List, Tuple


def shard_batch(n_items: int, rank: int, world: int) -> List[int]:
    """Contiguous, balanced partition of range(n_items) (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def collective_group_active(group=None) -> bool:
    """True when the data-parallel exchange has to run: a process group of more than one rank -- or of ONE rank with
    CRAFT_FORCE_COLLECTIVES=1 in the environment, which makes every collective of the training step (the flat-gradient all-reduce,
    the replica broadcasts, the timing protocol's reductions) execute on a one-GPU box: RCCL's communicator set-up and its kernels on
    the device buffers run for real instead of being short-circuited (tests/test_bench_contract.py::test_rccl_world1_*)."""
    import os
    d = _dist()
    if d is None:
        return False
    return d.get_world_size(group) > 1 or os.environ.get("CRAFT_FORCE_COLLECTIVES", "0") not in ("", "0")


def timed_steps(step: Callable[[], None], steps: int, warmup: int, sync: Callable[[], None]) -> float:
    """`warmup` untimed steps, then exactly `steps` timed steps bracketed by barrier + device sync on both
    sides.  Returns this rank's elapsed seconds (which includes waiting for the slowest rank at the closing barrier)."""
    d = _dist()
    for _ in range(warmup):
        step()
    sync()
    if d:
        d.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if d:
        d.barrier()
    sync()
    return time.perf_counter() - t0


def aggregate_throughput(pairs_per_rank_step: int, steps: int, dt: float) -> Tuple[float, float]:
    """(whole-job pairs/s, max-over-ranks seconds): total pairs of all ranks / slowest rank's time."""
    import torch
    d = _dist()
    total, dt_max = pairs_per_rank_step * steps, dt
    if d:
        backend = d.get_backend()
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        d.all_reduce(t, op=d.ReduceOp.MAX)
        n = torch.tensor([float(total)], dtype=torch.float64, device=dev)
        d.all_reduce(n, op=d.ReduceOp.SUM)
        dt_max, total = float(t.item()), float(n.item())
    return total / dt_max, dt_max
