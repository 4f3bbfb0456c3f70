#!/usr/bin/env python3
"""tests/golden/augment_ref.npz: outputs of the reference augmentor's PURE-NUMPY methods, executed from its own source.

core/utils/augmentor.py cannot be imported here (it imports cv2 and torchvision at the top), but three of its methods use numpy only:
``FlowAugmentor.eraser_transform`` (:127-140), ``SparseFlowAugmentor.eraser_transform`` (:241-252) and
``SparseFlowAugmentor.resize_sparse_flow_map`` (:254-288) -- and with ``spatial_aug_prob = 0`` the two ``spatial_transform`` methods (:141-183,
:290-330) never reach their ``cv2.resize`` lines either: the scale / stretch / flip / crop draws, the flips and the crop run in numpy.  They are compiled out of the file's AST (the class bodies' FunctionDef nodes, as
tools/make_golden_harness.py does for random_shift) and run on seeded inputs; the fixture holds inputs, seeds and outputs.
tests/test_augment.py holds craft_amd.augment's eraser (same draws, truncating uint8 assignment) and craft_aug_sparse to them bit for bit.

Only runs in the build container.      python tools/make_golden_augment.py
"""
import ast
import os
import sys
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/core/utils/augmentor.py"


def ref_methods():
    tree = ast.parse(open(REF).read(), REF)
    out = {}
    for cls in [n for n in tree.body if isinstance(n, ast.ClassDef)]:
        for fn in [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("eraser_transform", "resize_sparse_flow_map", "spatial_transform")]:
            ns = {"np": np}
            exec(compile(ast.Module(body=[fn], type_ignores=[]), REF, "exec"), ns)
            out[f"{cls.name}.{fn.name}"] = ns[fn.name]
    return out


def main():
    m = ref_methods()
    assert set(m) == {"FlowAugmentor.eraser_transform", "SparseFlowAugmentor.eraser_transform", "SparseFlowAugmentor.resize_sparse_flow_map",
                      "FlowAugmentor.spatial_transform", "SparseFlowAugmentor.spatial_transform"}, m.keys()
    rs = np.random.RandomState(7)
    out = {}
    # ---- eraser (dense: bounds [50, 100) passed as default argument; sparse: literal 50 .. 100), images 120 x 150
    a1 = rs.randint(0, 256, size=(120, 150, 3)).astype(np.uint8)
    a2 = rs.randint(0, 256, size=(120, 150, 3)).astype(np.uint8)
    out["erase.img1"], out["erase.img2"] = a1, a2
    seeds = [0, 1, 2, 3, 5, 8, 13]
    out["erase.seeds"] = np.array(seeds)
    self_ = SimpleNamespace(eraser_aug_prob=0.5)
    for kind in ("FlowAugmentor", "SparseFlowAugmentor"):
        for sd in seeds:
            np.random.seed(sd)
            o1, o2 = m[f"{kind}.eraser_transform"](self_, a1.copy(), a2.copy())
            assert np.array_equal(o1, a1)
            out[f"erase.{kind}.{sd}"] = o2
    # ---- sparse flow map resize: KITTI-like sparsity, several scale pairs (the trainer uses fx = fy; the function takes both)
    H, W = 47, 83
    flow = (rs.randn(H, W, 2) * 6).astype(np.float32)
    valid = (rs.rand(H, W) > 0.6).astype(np.float32)
    out["sparse.flow"], out["sparse.valid"] = flow, valid
    scales = [(1.0, 1.0), (1.31, 1.31), (0.77, 0.77), (1.5, 0.9), (0.5, 0.5)]
    out["sparse.scales"] = np.array(scales, dtype=np.float64)
    for k, (fx, fy) in enumerate(scales):
        f, v = m["SparseFlowAugmentor.resize_sparse_flow_map"](SimpleNamespace(), flow.copy(), valid.copy(), fx=fx, fy=fy)
        out[f"sparse.{k}.flow"], out[f"sparse.{k}.valid"] = f, v
    # ---- spatial_transform without the resize (spatial_aug_prob = 0: cv2 is never touched): draw order, flips, crop (dense: crop inside the
    # frame; sparse: drawn with margins, clipped back), flow sign under the flips
    Hs_, Ws_ = 72, 104
    b1 = rs.randint(0, 256, size=(Hs_, Ws_, 3)).astype(np.uint8)
    b2 = rs.randint(0, 256, size=(Hs_, Ws_, 3)).astype(np.uint8)
    bf = (rs.randn(Hs_, Ws_, 2) * 4).astype(np.float32)
    bv = (rs.rand(Hs_, Ws_) > 0.5).astype(np.float32)
    out["spatial.img1"], out["spatial.img2"], out["spatial.flow"], out["spatial.valid"] = b1, b2, bf, bv
    sp_seeds = [0, 1, 2, 3, 4, 6, 9, 11]
    out["spatial.seeds"] = np.array(sp_seeds)
    crop = (40, 64)
    dense = SimpleNamespace(crop_size=crop, min_scale=-0.2, max_scale=0.5, spatial_aug_prob=0.0, stretch_prob=0.8, max_stretch=0.2, do_flip=True,
                            h_flip_prob=0.5, v_flip_prob=0.1)
    sparse = SimpleNamespace(crop_size=crop, min_scale=-0.2, max_scale=0.5, spatial_aug_prob=0.0, do_flip=True)
    flips = set()
    for sd in sp_seeds:
        np.random.seed(sd)
        o1, o2, of = m["FlowAugmentor.spatial_transform"](dense, b1.copy(), b2.copy(), bf.copy())
        out[f"spatial.dense.{sd}.img1"], out[f"spatial.dense.{sd}.img2"] = np.ascontiguousarray(o1), np.ascontiguousarray(o2)
        out[f"spatial.dense.{sd}.flow"] = np.ascontiguousarray(of).astype(np.float32)
        np.random.seed(sd)
        o1, o2, of, ov = m["SparseFlowAugmentor.spatial_transform"](sparse, b1.copy(), b2.copy(), bf.copy(), bv.copy())
        out[f"spatial.sparse.{sd}.img1"], out[f"spatial.sparse.{sd}.img2"] = np.ascontiguousarray(o1), np.ascontiguousarray(o2)
        out[f"spatial.sparse.{sd}.flow"], out[f"spatial.sparse.{sd}.valid"] = np.ascontiguousarray(of).astype(np.float32), np.ascontiguousarray(ov)
        flips.add(bool((np.asarray(of)[..., 0] * 1).size))
    out["spatial.crop"] = np.array(crop)
    path = os.path.join(ROOT, "tests", "golden", "augment_ref.npz")
    if "--check" in sys.argv:
        z = np.load(path)
        bad = [k for k in out if not np.array_equal(z[k], out[k])]
        print("differs:", bad if bad else "nothing")
        return 1 if bad else 0
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
