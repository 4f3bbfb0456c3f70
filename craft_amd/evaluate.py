"""Evaluation harness around the HIP hot path (SURVEY.md §8(a) row H', §8(f) item 1).

The reference's ``evaluate.py`` is a 1600-line script tied to cv2 / torchvision / fvcore; this is its minimal
counterpart for the path we ship: load a checkpoint -> pad -> ``model(image1, image2, iters, test_mode=1)`` -> unpad ->
metrics, for FlyingChairs (evaluate.py:248-280), MPI-Sintel (evaluate.py:445-602) and KITTI-2015 (evaluate.py:757-927),
plus the leaderboard writers (evaluate.py:106-150, :196-245).  Metrics are reduced on the GPU by ``craft_flow_metrics``
(EPE, 1/3/5 px rates, KITTI Fl-all, EPE by flow magnitude) -- the predictions never leave HBM except for submissions.

    python -m craft_amd.evaluate --model checkpoints/craft-sintel.pth --dataset sintel --root datasets/Sintel --iters 32
"""
from __future__ import annotations

import argparse
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import flow_io
from .flow_datasets import KITTI, FlyingChairs, FlowDataset, MpiSintel
from .hip import call
from .utils import InputPadder

MAG_ENDPOINTS = (1, 10, 20, 30, float("inf"))      # evaluate.py:452


class FlowMetrics:
    """Accumulates the harness metrics over batches on the device (one 16-double table, see craft_hip.h)."""

    def __init__(self, device, max_mag: float = 0.0):
        self.acc = torch.zeros(16, device=device, dtype=torch.float64)
        self.max_mag = max_mag           # > 0: also drop |gt| >= max_mag (training metrics, train.py:53)

    def update(self, flow_pr: torch.Tensor, flow_gt: torch.Tensor, valid: Optional[torch.Tensor] = None, gt_offset=(0.0, 0.0)):
        """flow_pr, flow_gt [B, 2, H, W]; valid [B, H, W] (>= 0.5 counts) or None."""
        B, _, H, W = flow_pr.shape
        pr = flow_pr.float().contiguous()
        gt = flow_gt.to(pr.device).float().contiguous()
        va = None if valid is None else valid.to(pr.device).float().contiguous()
        call("craft_flow_metrics", pr, gt, va, B, H, W, float(gt_offset[0]), float(gt_offset[1]), float(self.max_mag), self.acc)

    def result(self) -> Dict[str, float]:
        a = self.acc.cpu().numpy()
        n = max(a[1], 1.0)
        out = {"epe": a[0] / n, "px1": a[2] / n, "px3": a[3] / n, "px5": a[4] / n, "f1": 100.0 * a[5] / n, "count": a[1]}
        lo = 0
        for k, hi in enumerate(MAG_ENDPOINTS):
            out[f"epe_{lo}-{hi}"] = a[6 + k] / a[11 + k] if a[11 + k] > 0 else 0.0      # evaluate.py:552-556
            lo = hi
        return out


def shift_pixels(img: torch.Tensor, flow: Optional[torch.Tensor], xy_shift):
    """evaluate.py:44-89 (the shift-robustness experiment of shifteval.sh): frame 1 moved by (x_shift, y_shift) pixels, vacated
    pixels zero, the ground truth moved with it and reduced by the shift -> (img, flow, mask [H, W] bool: pixels that still hold
    image content).  None / (0, 0): unchanged, mask all true.  Like the reference, only shifts with BOTH components non-zero move
    anything (its four quadrant branches test `> 0` / `< 0` on both axes, :64-83): a shift along one axis alone yields an all-zero
    frame 1 and an empty mask there, and does here."""
    if xy_shift is None or (int(xy_shift[0]) == 0 and int(xy_shift[1]) == 0):
        return img, flow, torch.ones(img.shape[-2:], dtype=torch.bool, device=img.device)
    xs, ys = int(xy_shift[0]), int(xy_shift[1])
    H, W = img.shape[-2:]
    img2 = torch.zeros_like(img)
    flow2 = None if flow is None else torch.zeros_like(flow)
    mask = torch.zeros(H, W, dtype=torch.bool, device=img.device)
    if xs != 0 and ys != 0 and abs(xs) < W and abs(ys) < H:
        dst_y, src_y = (slice(ys, None), slice(None, -ys)) if ys > 0 else (slice(None, ys), slice(-ys, None))
        dst_x, src_x = (slice(xs, None), slice(None, -xs)) if xs > 0 else (slice(None, xs), slice(-xs, None))
        img2[..., dst_y, dst_x] = img[..., src_y, src_x]
        mask[dst_y, dst_x] = True
        if flow is not None:
            flow2[..., dst_y, dst_x] = flow[..., src_y, src_x]
    if flow2 is not None:
        off = torch.tensor([xs, ys], dtype=flow2.dtype, device=flow2.device).reshape([1, 2, 1, 1] if flow2.dim() == 4 else [2, 1, 1])
        flow2 = flow2 - off                               # (u, v) = (x, y): the content of frame 1 moved, its displacement shrinks
    return img2, flow2, mask


def _shifted(image1, flow_gt, xy_shift, valid=None):
    """shift frame 1 and its ground truth; -> (image1, flow_gt, valid [B, H, W] float or None, gt_offset for the magnitude bins).
    KITTI's sparse validity map is NOT moved with the flow in the reference, only masked (evaluate.py:812-814): same here."""
    if xy_shift is None:
        return image1, flow_gt, valid, (0.0, 0.0)
    image1, flow_gt, mask = shift_pixels(image1, flow_gt, xy_shift)
    m = mask.unsqueeze(0).expand(image1.shape[0], -1, -1).float()
    valid = m if valid is None else valid.to(m.device).float() * m
    return image1, flow_gt, valid, (float(xy_shift[0]), float(xy_shift[1]))


def _predict(model, image1, image2, iters, pad_mode, device, flow_init=None):
    """pad -> forward (test_mode=1) -> unpad; images [B, 3, H, W] float 0..255 on any device."""
    image1, image2 = image1.to(device), image2.to(device)
    padder = InputPadder(image1.shape, mode=pad_mode, mod=8)
    image1, image2 = padder.pad(image1, image2)
    flow_low, flow_up = model(image1, image2, iters=iters, flow_init=flow_init, test_mode=1)
    return flow_low, padder.unpad(flow_up)


def _batches(ds: FlowDataset, batch_size: int, max_count: int):
    n = len(ds) if max_count < 0 else min(max_count, len(ds))
    for i in range(0, n, batch_size):
        items = [ds[j] for j in range(i, min(i + batch_size, n))]
        yield [torch.stack([it[k] for it in items]) for k in range(4)]


@torch.no_grad()
def validate_chairs(model, root="datasets/FlyingChairs_release/data", iters=6, batch_size=1, split_file=None, max_val_count=-1,
                    device="cuda", xy_shift=None):
    """FlyingChairs validation split (evaluate.py:248-280) -> {'chairs_epe': mean EPE over all pixels}; ``xy_shift`` = (x, y):
    frame 1 and its ground truth shifted first, EPE over the pixels that kept content (:270-276)."""
    model.eval()
    if xy_shift is not None:
        print(f"Apply x,y shift {int(xy_shift[0])},{int(xy_shift[1])}")
    ds = FlyingChairs(split="validation", root=root, split_file=split_file)
    m = FlowMetrics(device)
    for image1, image2, flow_gt, _ in _batches(ds, batch_size, max_val_count):
        image1, flow_gt, valid, off = _shifted(image1.to(device), flow_gt.to(device), xy_shift)
        _, flow = _predict(model, image1, image2, iters, "sintel", device)
        m.update(flow, flow_gt, valid, gt_offset=off)
    r = m.result()
    print("Validation Chairs EPE: %f" % r["epe"])
    return {"chairs_epe": r["epe"]}


@torch.no_grad()
def validate_sintel(model, root="datasets/Sintel", iters=6, dstype="both", batch_size=1, max_val_count=-1, device="cuda", xy_shift=None):
    """MPI-Sintel training split, clean and/or final pass (evaluate.py:445-602) -> {dstype: mean EPE}; prints the
    1/3/5 px rates and the EPE per ground-truth magnitude range like the reference.  ``xy_shift``: the shift experiment
    (:510-511; the magnitude ranges are those of the UNSHIFTED ground truth, :534)."""
    model.eval()
    if xy_shift is not None:
        print(f"Apply x,y shift {int(xy_shift[0])},{int(xy_shift[1])}")
    results = {}
    for dst in (["clean", "final"] if dstype == "both" else [dstype]):
        ds = MpiSintel(split="training", root=root, dstype=dst)
        m = FlowMetrics(device)
        for image1, image2, flow_gt, _ in _batches(ds, batch_size, max_val_count):
            image1, flow_gt, valid, off = _shifted(image1.to(device), flow_gt.to(device), xy_shift)
            _, flow = _predict(model, image1, image2, iters, "sintel", device)
            m.update(flow, flow_gt, valid, gt_offset=off)     # (no shift: the reference counts every pixel, val_mask is all ones)
        r = m.result()
        line = "Iter 0, Valid (%s) EPE: %f, 1px: %f, 3px: %f, 5px: %f" % (dst, r["epe"], r["px1"], r["px3"], r["px5"])
        lo = 0
        for hi in MAG_ENDPOINTS:
            line += f", {lo}-{hi} {r[f'epe_{lo}-{hi}']:.2f}"
            lo = hi
        print(line)
        results[dst] = r["epe"]
        results[dst + "_metrics"] = r
    return results


@torch.no_grad()
def validate_kitti(model, root="datasets/KITTI", iters=6, batch_size=1, max_val_count=-1, device="cuda", xy_shift=None):
    """KITTI-2015 training split (evaluate.py:757-927): sparse ground truth, bottom padding; -> {'epe', 'f1'} with
    f1 = 100 * mean(epe > 3 and epe / |gt| > 0.05) over valid pixels."""
    model.eval()
    if xy_shift is not None:
        print(f"Apply x,y shift {int(xy_shift[0])},{int(xy_shift[1])}")
    ds = KITTI(split="training", root=root)
    m = FlowMetrics(device)
    for image1, image2, flow_gt, valid_gt in _batches(ds, batch_size, max_val_count):
        image1, flow_gt, valid_gt, off = _shifted(image1.to(device), flow_gt.to(device), xy_shift, valid_gt)
        _, flow = _predict(model, image1, image2, iters, "kitti", device)
        m.update(flow, flow_gt, valid_gt, gt_offset=off)
    r = m.result()
    print("Iter 0, Valid EPE: %.4f, F1: %.4f, 1px: %.4f, 3px: %.4f, 5px: %.4f" % (r["epe"], r["f1"], r["px1"], r["px3"], r["px5"]))
    return {"epe": r["epe"], "f1": r["f1"], "metrics": r}


@torch.no_grad()
def create_sintel_submission(model, root="datasets/Sintel", output_path="sintel_submission", iters=32, split="test",
                             device="cuda", warm_start=False):
    """One ``frameXXXX.flo`` per pair under <output_path>/<clean|final>/<scene>/ (evaluate.py:106-150).
    ``warm_start``: the low-resolution flow of a frame, forward-interpolated on the GPU (``utils.forward_interpolate``),
    initialises the next frame of the same scene (RAFT's warm start).  The reference computes that field
    (evaluate.py:146-147) but never passes it to the model, so its submissions equal
    ``warm_start=False``, the default here."""
    from .utils import forward_interpolate
    model.eval()
    for dst in ("clean", "final"):
        ds = MpiSintel(split=split, root=root, dstype=dst)
        ds.is_test = True
        flow_prev, scene_prev = None, None
        for i in range(len(ds)):
            image1, image2, (scene, frame_id) = ds[i]
            if scene != scene_prev:
                flow_prev = None
            scene_prev = scene
            flow_low, flow = _predict(model, image1[None], image2[None], iters, "sintel", device, flow_init=flow_prev)
            if warm_start:
                flow_prev = forward_interpolate(flow_low[0])[None]
            out_dir = os.path.join(output_path, dst, scene)
            os.makedirs(out_dir, exist_ok=True)
            flow_io.write_flo(os.path.join(out_dir, "frame%04d.flo" % (frame_id + 1)), flow[0].permute(1, 2, 0).cpu().numpy())


@torch.no_grad()
def create_kitti_submission(model, root="datasets/KITTI", output_path="kitti_submission", iters=24, device="cuda"):
    """16-bit flow PNGs named like the input frames (evaluate.py:196-245)."""
    model.eval()
    ds = KITTI(split="testing", root=root)
    os.makedirs(output_path, exist_ok=True)
    for i in range(len(ds)):
        image1, image2, (frame_id,) = ds[i]
        _, flow = _predict(model, image1[None], image2[None], iters, "kitti", device)
        flow_io.write_flow_kitti(os.path.join(output_path, frame_id), flow[0].permute(1, 2, 0).cpu().numpy())


def build_model(ns: argparse.Namespace, device="cuda"):
    """CRAFT(args) + checkpoint, the way evaluate.py:1524-1560 sets it up (DataParallel-style 'module.' keys accepted)."""
    from . import CRAFT, default_args
    from .utils import load_checkpoint
    over = {k: v for k, v in vars(ns).items() if (k in vars(default_args()) or k == "hip_precision") and v is not None}
    model = CRAFT(default_args(**over))
    if ns.model:
        load_checkpoint(model, ns.model, trusted=getattr(ns, "trust_checkpoint", False))
    return model.to(device).eval()


def main(argv=None):
    ap = argparse.ArgumentParser(description="Evaluate a CRAFT checkpoint on the HIP path")
    ap.add_argument("--model", help="checkpoint (.pth) in the reference's layout; omit for random weights")
    ap.add_argument("--dataset", required=True, choices=["chairs", "sintel", "kitti", "sintel_submission", "kitti_submission"])
    ap.add_argument("--root", help="dataset root (default: the reference's datasets/<name>)")
    ap.add_argument("--iters", type=int, default=None)
    ap.add_argument("--dstype", default="both", choices=["both", "clean", "final"])
    ap.add_argument("--batch_size", type=int, default=1)
    ap.add_argument("--max_val_count", type=int, default=-1)
    ap.add_argument("--output", default=None, help="output directory of the submission writers")
    ap.add_argument("--fullprec", dest="mixed_precision", action="store_false", help="exact fp32 MFMA path (evaluate.py:1455)")
    ap.add_argument("--hip_precision", default=None)
    ap.add_argument("--trust-checkpoint", dest="trust_checkpoint", action="store_true",
                    help="allow the full unpickler if the safe one (tensors + numpy scalars) cannot read the file")
    ap.add_argument("--warm_start", action="store_true", help="sintel_submission: initialise each frame with the previous flow")
    ap.add_argument("--xshifts", dest="x_shifts", default=None, help="comma-separated x shifts of frame 1 (evaluate.py:1469; shifteval.sh)")
    ap.add_argument("--yshifts", dest="y_shifts", default=None, help="comma-separated y shifts, paired with --xshifts")
    ns = ap.parse_args(argv)
    ns.mixed_precision = bool(ns.mixed_precision)
    model = build_model(ns)
    # evaluate.py:1572-1577, :1604: one validation pass per (x, y) pair
    shifts = list(zip([int(x) for x in ns.x_shifts.split(",")], [int(y) for y in ns.y_shifts.split(",")])) if ns.x_shifts and ns.y_shifts else [None]
    if ns.dataset in ("chairs", "sintel", "kitti"):
        out = None
        for xy in shifts:
            if ns.dataset == "chairs":
                out = validate_chairs(model, ns.root or "datasets/FlyingChairs_release/data", ns.iters or 6, ns.batch_size,
                                      max_val_count=ns.max_val_count, xy_shift=xy)
            elif ns.dataset == "sintel":
                out = validate_sintel(model, ns.root or "datasets/Sintel", ns.iters or 32, ns.dstype, ns.batch_size, ns.max_val_count, xy_shift=xy)
            else:
                out = validate_kitti(model, ns.root or "datasets/KITTI", ns.iters or 24, ns.batch_size, ns.max_val_count, xy_shift=xy)
        return out
    if ns.dataset == "sintel_submission":
        return create_sintel_submission(model, ns.root or "datasets/Sintel", ns.output or "sintel_submission", ns.iters or 32,
                                        warm_start=ns.warm_start)
    return create_kitti_submission(model, ns.root or "datasets/KITTI", ns.output or "kitti_submission", ns.iters or 24)


if __name__ == "__main__":
    main()
