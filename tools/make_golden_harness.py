#!/usr/bin/env python3
"""Pin the harness row H'  to the reference's OWN code, run here.

`/root/reference/evaluate.py` and `train.py` cannot be imported (cv2, torchvision, fvcore, imageio are absent and they
call `.cuda()`), but the functions that matter are self-contained torch/numpy code.  This script parses those files with
`ast`, compiles the reference's own FunctionDef nodes -- `shift_pixels`, `validate_chairs`, `validate_sintel`,
`validate_kitti` (evaluate.py:44-89, :247-281, :445-602, :757-927) and `sequence_loss` (train.py:44-73) -- unchanged, and
runs them on synthetic data.  Supplied from outside (data providers only, no arithmetic): `datasets.<Name>(...)` returning
miniature synthetic datasets, a deterministic stand-in for the network (a pure function of the padded images), and
`Tensor.cuda` as the identity (no GPU in this container).  `InputPadder` is the reference's class, imported from
core/utils/utils.py.  What is committed (tests/golden/harness.npz) is data: the inputs and the numbers the reference's code
returned / printed.

    python tools/make_golden_harness.py
"""
import ast
import contextlib
import io
import os
import re
import sys

import numpy as np
import torch
import torch.nn.functional as F
import torch.utils.data as data

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(REF, "core"))

from utils.utils import InputPadder  # noqa: E402  (the reference's)


def ref_functions(path, names, ns):
    """Compile the named top-level FunctionDefs of a reference file into `ns` (decorators kept)."""
    tree = ast.parse(open(path).read(), path)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in picked) == sorted(names), [n.name for n in picked]
    mod = ast.Module(body=picked, type_ignores=[])
    exec(compile(mod, path, "exec"), ns)
    return ns


class StandInNet(torch.nn.Module):
    """Deterministic stand-in for the network: a pure function of the (padded) images with a 5x5 spatial support, so the
    cropped prediction depends on how the harness padded.  Returns (low-res placeholder, full-resolution flow)."""

    def forward(self, image1, image2, iters=6, flow_init=None, test_mode=1, **kw):
        d = image1[:, :2].float() - image2[:, 1:3].float()
        up = F.avg_pool2d(d, 5, stride=1, padding=2, count_include_pad=False) / 2.0
        lo = F.avg_pool2d(up, 8) / 8.0
        if test_mode == 2:
            return lo, [up * (0.5 + 0.5 * (i + 1) / iters) for i in range(iters)]
        return lo, up


def make_set(n, H, W, seed, sparse):
    """uint8 images whose stand-in prediction is ground truth + noise of mixed scale; gt on the 1/64 px grid (the KITTI
    PNG encoding is then lossless)."""
    r = np.random.RandomState(seed)
    h, w = (H + 7) // 8, (W + 7) // 8

    def blocks(a):                       # piecewise-constant 8x8 blocks: survives the stand-in's 5x5 average
        return np.repeat(np.repeat(a, 8, axis=2), 8, axis=3)[:, :, :H, :W]
    gt = np.round(blocks(r.standard_normal((n, 2, h, w)) * r.choice([0.3, 4.0, 10.0, 16.0], size=(n, 1, h, w))) * 64) / 64
    noise = blocks(r.standard_normal((n, 2, h, w)) * r.choice([0.2, 1.5, 4.0], size=(n, 1, h, w)))
    field = np.clip(np.round((gt + noise) * 2), -100, 100)
    im1 = r.randint(100, 156, size=(n, 3, H, W)).astype(np.float32)
    im2 = r.randint(0, 256, size=(n, 3, H, W)).astype(np.float32)
    im2[:, 1:3] = im1[:, :2] - field
    assert im2.min() >= 0 and im2.max() <= 255
    valid = (r.random_sample((n, H, W)) > 0.45).astype(np.float32) if sparse else np.ones((n, H, W), np.float32)
    return dict(im1=im1.astype(np.uint8), im2=im2.astype(np.uint8), gt=gt.astype(np.float32), valid=valid)


class SetDS(data.Dataset):
    def __init__(self, s):
        self.s = s

    def __len__(self):
        return len(self.s["im1"])

    def __getitem__(self, i):
        s = self.s
        return (torch.from_numpy(s["im1"][i].astype(np.float32)), torch.from_numpy(s["im2"][i].astype(np.float32)),
                torch.from_numpy(s["gt"][i]), torch.from_numpy(s["valid"][i]), i)


class Provider:
    """`datasets` as the reference's evaluate.py sees it: constructors that hand out the synthetic sets."""

    def __init__(self, sets):
        self.sets = sets

    def MpiSintel(self, split="training", aug_params=None, dstype="clean"):
        return SetDS(self.sets["sintel_" + dstype])

    def KITTI(self, split="training", **kw):
        return SetDS(self.sets["kitti"])

    def FlyingChairs(self, split="validation", **kw):
        return SetDS(self.sets["chairs"])


def run_captured(fn, *a, **kw):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **kw)
    return out, buf.getvalue()


def main():
    sets = {"sintel_clean": make_set(3, 132, 250, 1, False), "sintel_final": make_set(3, 132, 250, 2, False),
            "kitti": make_set(2, 123, 310, 3, True), "chairs": make_set(2, 96, 128, 4, False)}
    ns = {"torch": torch, "np": np, "data": data, "InputPadder": InputPadder, "datasets": Provider(sets), "F": F}
    ref_functions(os.path.join(REF, "evaluate.py"), ["shift_pixels", "validate_chairs", "validate_sintel", "validate_kitti"], ns)
    tns = {"torch": torch, "MAX_FLOW": None}
    tree = ast.parse(open(os.path.join(REF, "train.py")).read())
    for n in tree.body:      # MAX_FLOW = 400 (train.py:30), taken from the file rather than restated
        if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", None) == "MAX_FLOW":
            tns["MAX_FLOW"] = ast.literal_eval(n.value)
    ref_functions(os.path.join(REF, "train.py"), ["sequence_loss"], tns)

    out = {}
    for k, s in sets.items():
        for f, v in s.items():
            out[f"{k}.{f}"] = v
    net = StandInNet()
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **kw: self
    try:
        res, txt = run_captured(ns["validate_sintel"], net, iters=4, test_mode=1, batch_size=2)
        out["sintel.clean"], out["sintel.final"] = np.float64(res["clean"]), np.float64(res["final"])
        rows = re.findall(r"Valid \((\w+)\) EPE: ([\d.]+), 1px: ([\d.]+), 3px: ([\d.]+), 5px: ([\d.]+)((?:, [\d.a-z-]+ [\d.]+)+)", txt)
        assert len(rows) == 2, txt
        for dst, epe, p1, p3, p5, mags in rows:
            out[f"sintel.{dst}.px"] = np.array([float(p1), float(p3), float(p5)])
            out[f"sintel.{dst}.mag"] = np.array([float(x.split()[1]) for x in mags.strip(", ").split(", ")])
        res, txt = run_captured(ns["validate_sintel"], net, iters=4, test_mode=1, batch_size=1, xy_shift=None, dstype="clean")
        assert abs(res["clean"] - out["sintel.clean"]) < 1e-6
        res, txt = run_captured(ns["validate_kitti"], net, iters=4, test_mode=1, batch_size=1)
        out["kitti.epe"], out["kitti.f1"] = np.float64(res["epe"]), np.float64(res["f1"])
        m = re.search(r"Valid EPE: ([\d.]+), F1: ([\d.]+), 1px: ([\d.]+), 3px: ([\d.]+), 5px: ([\d.]+)((?:, [\d.a-z-]+ [\d.]+)+)", txt)
        out["kitti.px"] = np.array([float(m.group(i)) for i in (3, 4, 5)])
        out["kitti.mag"] = np.array([float(x.split()[1]) for x in m.group(6).strip(", ").split(", ")])
        res, txt = run_captured(ns["validate_chairs"], net, iters=4, test_mode=1, batch_size=2)
        out["chairs.epe"] = np.float64(res["chairs_epe"])
        # the shift-robustness experiment (evaluate.py:44-89, xy_shift of every validator; shifteval.sh): the reference's validators on
        # the same sets with frame 1 shifted -- one pair per quadrant of shift_pixels, plus the magnitudes of the UNSHIFTED ground truth
        shifts = {"sintel": [(7, -5), (-9, 4)], "kitti": [(-6, -4)], "chairs": [(5, 3)]}
        out["shiftval.sintel"], out["shiftval.kitti"], out["shiftval.chairs"] = (np.array(shifts[k]) for k in ("sintel", "kitti", "chairs"))
        for i, xy in enumerate(shifts["sintel"]):
            res, txt = run_captured(ns["validate_sintel"], net, iters=4, test_mode=1, batch_size=2, xy_shift=xy, dstype="clean")
            out[f"shiftval.sintel.{i}.epe"] = np.float64(res["clean"])
            rows = re.findall(r"Valid \((\w+)\) EPE: ([\d.]+), 1px: ([\d.]+), 3px: ([\d.]+), 5px: ([\d.]+)((?:, [\d.a-z-]+ [\d.]+)+)", txt)
            assert len(rows) == 1, txt
            out[f"shiftval.sintel.{i}.px"] = np.array([float(x) for x in rows[0][2:5]])
            out[f"shiftval.sintel.{i}.mag"] = np.array([float(x.split()[1]) for x in rows[0][5].strip(", ").split(", ")])
        for i, xy in enumerate(shifts["kitti"]):
            res, txt = run_captured(ns["validate_kitti"], net, iters=4, test_mode=1, batch_size=1, xy_shift=xy)
            out[f"shiftval.kitti.{i}.epe"], out[f"shiftval.kitti.{i}.f1"] = np.float64(res["epe"]), np.float64(res["f1"])
        for i, xy in enumerate(shifts["chairs"]):
            res, txt = run_captured(ns["validate_chairs"], net, iters=4, test_mode=1, batch_size=2, xy_shift=xy)
            out[f"shiftval.chairs.{i}.epe"] = np.float64(res["chairs_epe"])
        # shift_pixels itself: all four quadrants, a shift along one axis only (moves nothing: see craft_amd.evaluate.shift_pixels), 3-D input
        gs = torch.Generator().manual_seed(21)
        pim = torch.randint(0, 256, (2, 3, 11, 14), generator=gs).float()
        pfl = torch.randn(2, 2, 11, 14, generator=gs) * 4
        out["shiftpx.img"], out["shiftpx.flow"] = pim.numpy(), pfl.numpy()
        cases = [(3, 2), (4, -3), (-2, 5), (-1, -6), (0, 3), (5, 0), (0, 0)]
        out["shiftpx.cases"] = np.array(cases)
        for i, xy in enumerate(cases):
            a, b, mk = ns["shift_pixels"](pim.clone(), pfl.clone(), xy)
            out[f"shiftpx.{i}.img"], out[f"shiftpx.{i}.flow"], out[f"shiftpx.{i}.mask"] = a.numpy(), b.numpy(), mk.numpy()
        a, b, mk = ns["shift_pixels"](pim[0].clone(), pfl[0].clone(), (3, 2))
        out["shiftpx.3d.img"], out["shiftpx.3d.flow"] = a.numpy(), b.numpy()
    finally:
        torch.Tensor.cuda = orig_cuda

    # InputPadder: pads of the reference's class for a set of sizes / modes / moduli, and one padded + unpadded tensor
    dims = [(436, 1024), (375, 1242), (132, 250), (128, 256), (370, 1224), (1, 9)]
    pads = []
    for d in dims:
        for mode in ("sintel", "kitti"):
            for mod in (8, 16):
                pads.append([d[0], d[1], 0 if mode == "sintel" else 1, mod] + list(InputPadder((1, 3) + d, mode=mode, mod=mod)._pad))
    out["padder.table"] = np.array(pads, dtype=np.int64)
    x = torch.arange(2 * 3 * 13 * 21, dtype=torch.float32).reshape(2, 3, 13, 21)
    for mode in ("sintel", "kitti"):
        p = InputPadder(x.shape, mode=mode)
        (xp,) = p.pad(x)
        out[f"padder.{mode}.padded"] = xp.numpy()
        assert torch.equal(p.unpad(xp), x)
    out["padder.x"] = x.numpy()

    # sequence_loss (train.py:44-73): loss, metrics and d loss / d pred_i from autograd through the reference's function
    g = torch.Generator().manual_seed(11)
    B, H, W, T = 2, 40, 56, 5
    gt = torch.randn(B, 2, H, W, generator=g) * torch.tensor([3.0, 30.0]).view(2, 1, 1, 1)
    gt[0, :, :4, :4] = 500.0                                 # beyond MAX_FLOW: excluded
    valid = (torch.rand(B, H, W, generator=g) > 0.3).float()
    preds = [(gt + torch.randn(B, 2, H, W, generator=g) * (9.0 / (i + 1))).requires_grad_(True) for i in range(T)]
    loss, metrics = tns["sequence_loss"](preds, gt, valid, 0.8)
    loss.backward()
    out["loss.gt"], out["loss.valid"] = gt.numpy(), valid.numpy()
    out["loss.preds"] = torch.stack([p.detach() for p in preds]).numpy()
    out["loss.value"] = np.float64(loss.item())
    out["loss.metrics"] = np.array([metrics["epe"], metrics["1px"], metrics["3px"], metrics["5px"]])
    out["loss.grads"] = torch.stack([p.grad for p in preds]).numpy()
    out["loss.max_flow"] = np.float64(tns["MAX_FLOW"])

    # random_shift (core/utils/augmentor.py:16-78): the reference's own function (pure numpy + the `random` module; the file
    # itself needs cv2 / torchvision, so only this FunctionDef is compiled), seeded, on integer-valued images
    import random
    ans = {"np": np, "random": random}
    ref_functions(os.path.join(REF, "core", "utils", "augmentor.py"), ["random_shift"], ans)
    r = np.random.RandomState(5)
    Hh, Ww = 48, 64
    a1 = r.randint(0, 256, size=(Hh, Ww, 3)).astype(np.float32)
    a2 = r.randint(0, 256, size=(Hh, Ww, 3)).astype(np.float32)
    fl = (r.standard_normal((Hh, Ww, 2)) * 5).astype(np.float32)
    out["shift.img1"], out["shift.img2"], out["shift.flow"] = a1, a2, fl
    seeds = [1, 2, 3, 4, 5, 6, 7, 8]
    out["shift.seeds"] = np.array(seeds)
    for sd in seeds:
        random.seed(sd); np.random.seed(sd)
        o1, o2, of, vm = ans["random_shift"](a1, a2, fl, shift_sigmas=(16, 10))
        out[f"shift.{sd}.img1"], out[f"shift.{sd}.img2"] = o1.astype(np.float32), o2.astype(np.float32)
        out[f"shift.{sd}.flow"], out[f"shift.{sd}.valid"] = of.astype(np.float32), vm.astype(np.float32)

    path = os.path.join(ROOT, "tests", "golden", "harness.npz")
    np.savez_compressed(path, **out)
    print(path, f"{os.path.getsize(path) / 1024:.0f} KiB", {k: float(v) for k, v in out.items() if np.ndim(v) == 0})


if __name__ == "__main__":
    main()
