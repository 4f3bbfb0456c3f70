"""Training-side input feed (core/datasets.py:52-141 ``FlowDataset.__getitem__`` in training mode, :509-580 ``fetch_dataloader``):
files -> GPU -> augmentation on the GPU (craft_amd/augment.py) -> batches for ``Trainer.step``.

The reference decodes and augments every sample with cv2 / PIL on CPU workers and ships crops to the GPU; here a sample is decoded
on the host (craft_amd/flow_io.py), uploaded ONCE at its full size and cropped / jittered / erased / shifted by HIP kernels, so the
per-sample CPU work is the file decode only.  The random-draw order per sample follows the reference's augmentors (numpy /
``random`` global generators), the dataset mixing follows ``fetch_dataloader`` (replication factors, shuffle, drop_last), and
data-parallel ranks take the strided shard ``DistributedSampler`` would give them.

Only the dataset walkers that exist in craft_amd/flow_datasets.py can be mixed (FlyingChairs, MPI-Sintel, KITTI): the Things /
HD1K / AutoFlow / VIPER directory layouts are not restated, their stage parameters are listed for completeness.
"""
from __future__ import annotations

import random
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .augment import FlowAugmentor, SparseFlowAugmentor
from .flow_datasets import FlowDataset

# fetch_dataloader's per-stage augmentation parameters (datasets.py:512-567); crop_size = args.image_size
STAGE_AUG = {
    "chairs": dict(min_scale=-0.1, max_scale=1.0, do_flip=True),
    "things": dict(min_scale=-0.4, max_scale=0.8, do_flip=True),
    "autoflow": dict(min_scale=-0.2, max_scale=0.8, spatial_aug_prob=1, do_flip=True),
    "sintel": dict(min_scale=-0.2, max_scale=0.6, do_flip=True),
    "sintel/kitti": dict(min_scale=-0.3, max_scale=0.5, do_flip=True),        # the KITTI share of the sintel stage
    "sintel/hd1k": dict(min_scale=-0.5, max_scale=0.2, do_flip=True),
    "kitti": dict(min_scale=-0.2, max_scale=0.4, do_flip=False),
    "kittitrain": dict(min_scale=-0.2, max_scale=0.4, do_flip=False),
    "viper": dict(min_scale=-1, max_scale=-0.5, spatial_aug_prob=1, do_flip=False),
}


def make_augmentor(dataset: FlowDataset, stage_key: str, crop_size: Sequence[int], shift_prob: float = 0.0, shift_sigmas=(16, 10)):
    """The augmentor FlowDataset.__init__ builds from aug_params (datasets.py:27-36): sparse datasets get SparseFlowAugmentor."""
    p = dict(STAGE_AUG[stage_key], crop_size=tuple(crop_size), shift_prob=shift_prob, shift_sigmas=shift_sigmas)
    cls = SparseFlowAugmentor if dataset.sparse else FlowAugmentor
    return cls(getattr(dataset, "ds_name", stage_key), **p)


class TrainSource:
    """One dataset + its augmentor; ``sample(i, device)`` = FlowDataset.__getitem__ (datasets.py:52-141) with the augmentation on
    the device -> (img1 [3,h,w], img2, flow [2,h,w], valid [h,w]) float32 on ``device``."""

    def __init__(self, dataset: FlowDataset, augmentor, repeat: int = 1):
        if dataset.is_test:
            raise ValueError("a test split has no ground truth to train on")
        self.dataset, self.augmentor, self.repeat = dataset, augmentor, int(repeat)

    def __len__(self) -> int:
        return len(self.dataset) * self.repeat                    # `v * dataset` of the reference (datasets.py:143-148)

    def sample(self, index: int, device) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        return self.augment(self.decode(index), device)

    def decode(self, index: int, pin: bool = False):
        """The host half of a sample: read and decode the files (no random draws: safe on any worker thread, in any order)."""
        img1, img2, flow, valid, _ = self.dataset[index % len(self.dataset)]
        out = (img1, img2, flow, valid)
        return tuple(t.pin_memory() for t in out) if pin else out

    def augment(self, decoded, device) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """The device half: upload (on the current stream; asynchronous from pinned memory) and augment with the HIP kernels.  Draws
        from the global numpy / ``random`` generators in the reference's order -- call it in sample order from ONE thread."""
        img1, img2, flow, valid = decoded
        nb = img1.is_pinned()
        a = img1.to(device, non_blocking=nb).permute(1, 2, 0).contiguous()         # HWC, 0..255
        b = img2.to(device, non_blocking=nb).permute(1, 2, 0).contiguous()
        f = flow.to(device, non_blocking=nb).permute(1, 2, 0).contiguous()
        if self.dataset.sparse:
            a, b, f, v = self.augmentor(a, b, f, valid.to(device, non_blocking=nb))
        else:
            a, b, f, v = self.augmentor(a, b, f)
            if v is None:
                v = ((f[..., 0].abs() < 1000) & (f[..., 1].abs() < 1000)).float()
        return a.permute(2, 0, 1).contiguous(), b.permute(2, 0, 1).contiguous(), f.permute(2, 0, 1).contiguous(), v.float()


def train_batches(sources: List[TrainSource], batch_size: int, device, seed: int = 0, rank: int = 0, world: int = 1,
                  epochs: Optional[int] = None) -> Iterator[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]]:
    """The reference's DataLoader over the concatenated datasets (datasets.py:569-580): shuffle=True, drop_last=True; with
    ``world`` > 1 every rank walks its strided share of the epoch's permutation (DistributedSampler semantics).  Yields
    (image1 [B,3,h,w], image2, flow [B,2,h,w], valid [B,h,w]) on ``device`` forever (or for ``epochs`` passes)."""
    table = [(si, i) for si, s in enumerate(sources) for i in range(len(s))]
    if len(table) < batch_size * world:
        raise ValueError(f"{len(table)} training samples for {world} rank(s) x batch {batch_size}")
    epoch = 0
    while epochs is None or epoch < epochs:
        order = np.random.RandomState(seed + epoch).permutation(len(table))
        if world > 1:                                             # DistributedSampler: pad to a multiple of `world` so that every rank
            pad = (-len(order)) % world                           # runs the same number of steps (an all-reduce per step needs them all)
            order = np.concatenate([order, order[:pad]])
        mine = order[rank::world]
        for b0 in range(0, len(mine) - batch_size + 1, batch_size):
            items = [sources[table[j][0]].sample(table[j][1], device) for j in mine[b0:b0 + batch_size]]
            yield tuple(torch.stack([it[k] for it in items]) for k in range(4))
        epoch += 1


def _epoch_indices(sources, batch_size, seed, rank, world, epochs):
    """The (source, index) lists of successive batches, in the order train_batches visits them."""
    table = [(si, i) for si, s in enumerate(sources) for i in range(len(s))]
    if len(table) < batch_size * world:
        raise ValueError(f"{len(table)} training samples for {world} rank(s) x batch {batch_size}")
    epoch = 0
    while epochs is None or epoch < epochs:
        order = np.random.RandomState(seed + epoch).permutation(len(table))
        if world > 1:
            order = np.concatenate([order, order[:(-len(order)) % world]])
        mine = order[rank::world]
        for b0 in range(0, len(mine) - batch_size + 1, batch_size):
            yield [table[j] for j in mine[b0:b0 + batch_size]]
        epoch += 1


def train_batches_async(sources: List[TrainSource], batch_size: int, device, seed: int = 0, rank: int = 0, world: int = 1,
                        epochs: Optional[int] = None, workers: int = 12, prefetch: int = 2):
    """``train_batches`` with the input pipeline off the training step's critical path -- the counterpart of the reference's
    DataLoader(num_workers=4, pin_memory=True) (datasets.py:569-580, train.py:337): the same batches in the same order
    (tests/test_train_data.py), produced ahead of the consumer:

    * ``workers`` threads decode files into pinned host memory (zlib and the PNG un-filter helper release the GIL), ``prefetch``
      batches ahead, out of order but handed on in order;
    * ONE producer thread uploads each sample (asynchronous copies from pinned memory) and runs the augmentation kernels on a SIDE
      HIP stream -- the random draws stay on that one thread, in sample order, so a seed reproduces ``train_batches`` exactly;
    * the consumer's stream waits for the batch's event (no host synchronisation) and the caching allocator is told that the batch's
      tensors are used on the consumer's stream (``record_stream``).

    Yields (image1, image2, flow, valid) like ``train_batches``.  The generator owns its threads; closing it (or exhausting the epochs)
    stops them."""
    import queue
    import threading
    from concurrent.futures import ThreadPoolExecutor
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    side = torch.cuda.Stream(device=dev)
    q: "queue.Queue" = queue.Queue(maxsize=max(1, prefetch))
    stop = threading.Event()
    pool = ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="craft-decode")
    # the producer thread draws from the GLOBAL numpy / random generators (the augmentors' contract): hand it the caller's state
    np_state, py_state = np.random.get_state(), random.getstate()

    def producer():
        try:
            np.random.set_state(np_state)
            random.setstate(py_state)
            torch.cuda.set_device(dev)
            batches = _epoch_indices(sources, batch_size, seed, rank, world, epochs)
            window = []                                            # decode futures of the next batches, in order
            ahead = max(1, prefetch) + 1

            def fill():
                while len(window) < ahead:
                    try:
                        items = next(batches)
                    except StopIteration:
                        return
                    window.append([pool.submit(sources[si].decode, i, True) for si, i in items] + [items])
            fill()
            while window and not stop.is_set():
                futs = window.pop(0)
                items = futs.pop()
                fill()
                with torch.cuda.stream(side):
                    outs = [sources[si].augment(f.result(), dev) for f, (si, _) in zip(futs, items)]
                    batch = tuple(torch.stack([o[k] for o in outs]) for k in range(4))
                    ev = torch.cuda.Event()
                    ev.record(side)
                while not stop.is_set():
                    try:
                        q.put((batch, ev), timeout=0.1)
                        break
                    except queue.Full:
                        continue
            q.put(None)
        except BaseException as e:  # noqa: BLE001   (hand the failure to the consumer instead of dying silently)
            q.put(e)

    th = threading.Thread(target=producer, name="craft-feed", daemon=True)
    th.start()
    try:
        while True:
            item = q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            batch, ev = item
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(ev)
            for t in batch:
                t.record_stream(cur)
            yield batch
    finally:
        stop.set()
        while th.is_alive():                                       # unblock a producer stuck on a full queue
            try:
                q.get_nowait()
            except queue.Empty:
                th.join(timeout=0.05)
        pool.shutdown(wait=False, cancel_futures=True)


def seed_workers(seed: int) -> None:
    """The augmentors draw from numpy's and ``random``'s global generators like the reference's (datasets.py:69-75 seeds them per worker)."""
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)
