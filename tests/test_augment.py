"""Input pipeline on the GPU (reference core/utils/augmentor.py).

* random_shift: bit-for-bit against outputs of the reference's own function (tests/golden/harness.npz 'shift.*'), with (dx, dy)
  re-drawn by our sampler from the same seeds -- so the draw order is pinned too (CPU part) and the kernel (GPU part).
* ColorJitter's four operations: bit for bit against outputs of Pillow itself (tests/golden/photo_pil.npz; torchvision's PIL path wraps
  Pillow) -- the oracle (CPU part) and the kernel (GPU part).
* eraser (both augmentors) and the sparse flow-map resize: against the reference's OWN pure-numpy methods, compiled out of augmentor.py's AST
  (tests/golden/augment_ref.npz, tools/make_golden_augment.py).
* spatial gather (resize -> flips -> crop), blur: against numpy restatements written here (cv2 is not in this image: its 8-bit
  fixed-point filters are not pinned).
* Gaussian blur (augmentor.py:195-198): against a numpy restatement of cv2.getGaussianKernel (sigma > 0) + BORDER_REFLECT_101.
* FlowAugmentor end to end: shapes, ranges, flow consistency under a pure flip / crop."""
import os
import random

import numpy as np
import pytest
import torch

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "harness.npz"))


def test_shift_draw_order_matches_reference_outputs():
    """Our (dx, dy) sampler, seeded like the fixture, must reproduce the crop geometry the reference produced."""
    from craft_amd.augment import draw_shift
    H, W = Z["shift.img1"].shape[:2]
    for sd in Z["shift.seeds"].tolist():
        random.seed(sd); np.random.seed(sd)
        dx, dy = draw_shift((16, 10))
        valid = Z[f"shift.{sd}.valid"]
        assert dx % 2 == 0 and dy % 2 == 0
        assert int(valid.sum()) == (H - abs(dy)) * (W - abs(dx)), (sd, dx, dy)
        assert valid.shape == (H, W)


@pytest.mark.gpu
def test_random_shift_kernel_bit_exact(device):
    from craft_amd.augment import draw_shift, random_shift
    a1, a2, fl = (torch.from_numpy(Z[k]).to(device) for k in ("shift.img1", "shift.img2", "shift.flow"))
    seen = set()
    for sd in Z["shift.seeds"].tolist():
        random.seed(sd); np.random.seed(sd)
        dx, dy = draw_shift((16, 10))
        seen.add((np.sign(dx), np.sign(dy)))
        o1, o2, of, vm = random_shift(a1, a2, fl, dx, dy)
        assert np.array_equal(o1.cpu().numpy(), Z[f"shift.{sd}.img1"]), (sd, dx, dy)
        assert np.array_equal(o2.cpu().numpy(), Z[f"shift.{sd}.img2"])
        assert np.array_equal(of.cpu().numpy(), Z[f"shift.{sd}.flow"])
        assert np.array_equal(vm.cpu().numpy(), Z[f"shift.{sd}.valid"])
    assert len(seen) >= 3, "the fixture should cover several sign combinations of the shift"


def _resize_ref(a, fx, fy):
    """cv2.resize(a, None, fx, fy, INTER_LINEAR) in float: dst size round(n*f), src = (dst + .5)/f - .5, replicate border."""
    H, W, C = a.shape
    Hs, Ws = int(round(H * fy)), int(round(W * fx))
    ys = (np.arange(Hs) + 0.5) / np.float32(fy) - 0.5
    xs = (np.arange(Ws) + 0.5) / np.float32(fx) - 0.5
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
    wy = (ys - y0)[:, None, None].astype(np.float32); wx = (xs - x0)[None, :, None].astype(np.float32)
    cy = lambda v: np.clip(v, 0, H - 1); cx = lambda v: np.clip(v, 0, W - 1)
    A = a[cy(y0)][:, cx(x0)]; B = a[cy(y0)][:, cx(x0 + 1)]; D = a[cy(y0 + 1)][:, cx(x0)]; E = a[cy(y0 + 1)][:, cx(x0 + 1)]
    return (A * (1 - wx) + B * wx) * (1 - wy) + (D * (1 - wx) + E * wx) * wy


@pytest.mark.gpu
@pytest.mark.parametrize("do_resize,hflip,vflip", [(True, False, False), (True, True, True), (False, True, False), (False, False, True)])
def test_spatial_gather(device, do_resize, hflip, vflip):
    from craft_amd.augment import spatial
    r = np.random.RandomState(3)
    H, W, crop = 60, 90, (40, 56)
    img = r.randint(0, 256, size=(H, W, 3)).astype(np.float32)
    flow = (r.standard_normal((H, W, 2)) * 4).astype(np.float32)
    fx, fy = 1.37, 0.93
    for a, is_flow in ((img, False), (flow, True)):
        ref = _resize_ref(a, fx, fy) if do_resize else a.copy()
        if is_flow and do_resize:
            ref = ref * np.array([fx, fy], dtype=np.float32)
        if hflip:
            ref = ref[:, ::-1] * (np.array([-1.0, 1.0], dtype=np.float32) if is_flow else 1.0)
        if vflip:
            ref = ref[::-1] * (np.array([1.0, -1.0], dtype=np.float32) if is_flow else 1.0)
        y0, x0 = 7, 11
        ref = ref[y0:y0 + crop[0], x0:x0 + crop[1]]
        if not is_flow and do_resize:
            ref = np.clip(np.rint(ref), 0, 255)
        got = spatial(torch.from_numpy(a).to(device), crop, y0, x0, fx, fy, do_resize, hflip, vflip, is_flow).cpu().numpy()
        if is_flow or not do_resize:
            assert np.allclose(got, ref, rtol=1e-5, atol=2e-4)
        else:       # rounding to integer levels: values within float noise of a .5 boundary may land on either side
            assert np.abs(got - ref).max() <= 1.0 and (got != ref).mean() < 2e-3


PHOTO = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "photo_pil.npz"))
AUGREF = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment_ref.npz"))      # the reference's own numpy methods
PHOTO_IMAGES = ("rand", "grey", "dark", "prim")


def test_photo_oracle_matches_pillow_fixture():
    """oracle/augment_oracle.py (numpy restatement of Pillow's Blend.c / Convert.c arithmetic behind torchvision's ColorJitter) against the
    outputs Pillow itself produced (tools/make_golden_photo.py): every operation, factors inside and outside [0, 1], negative / zero hue
    shifts, exact greys, primaries, and three full jitter chains -- bit for bit.  With Pillow importable the same on larger random images."""
    from oracle import augment_oracle as A
    for name in PHOTO_IMAGES:
        a = PHOTO[f"{name}.in"]
        for op in range(4):
            for k, f in enumerate(PHOTO[f"factors{op}"]):
                assert np.array_equal(A.OPS[op](a, float(f)), PHOTO[f"{name}.op{op}.{k}"]), (name, op, f)
        for k in range(len(PHOTO["chain_orders"])):
            assert np.array_equal(A.color_jitter(a, PHOTO["chain_orders"][k], PHOTO["chain_factors"][k]), PHOTO[f"{name}.chain{k}"])
    try:
        from PIL import Image, ImageEnhance
    except ImportError:
        return
    rs = np.random.RandomState(5)
    a = rs.randint(0, 256, size=(192, 256, 3)).astype(np.uint8)
    for f in (0.61, 1.0, 1.39):
        assert np.array_equal(A.adjust_brightness(a, f), np.array(ImageEnhance.Brightness(Image.fromarray(a)).enhance(f)))
        assert np.array_equal(A.adjust_contrast(a, f), np.array(ImageEnhance.Contrast(Image.fromarray(a)).enhance(f)))
        assert np.array_equal(A.adjust_saturation(a, f), np.array(ImageEnhance.Color(Image.fromarray(a)).enhance(f)))
    hsv = np.array(Image.fromarray(a).convert("HSV"))
    uh, us, uv = A.rgb_to_hsv8(a)
    assert np.array_equal(np.stack([uh, us, uv], -1).astype(np.uint8), hsv)
    assert np.array_equal(A.hsv8_to_rgb(*(a[..., k].astype(np.int64) for k in range(3))), np.array(Image.fromarray(a, "HSV").convert("RGB")))


def test_hue_shift_is_torchvisions_truncation():
    from craft_amd.augment import hue_shift
    from oracle import augment_oracle as A
    for f, want in ((0.1, 25), (-0.1, 231), (0.159, 40), (-0.159, 216), (0.0, 0), (-0.001, 0), (0.003, 0), (0.5, 127), (-0.5, 129)):
        assert hue_shift(f) == A.hue_shift(f) == want


@pytest.mark.gpu
def test_photo_steps_match_pillow_bit_for_bit(device):
    """k_aug_photo against the Pillow-produced fixture (every array of it) and against the oracle on a larger random image."""
    from craft_amd.augment import photo_step
    from oracle import augment_oracle as A
    up = lambda a: torch.from_numpy(a.astype(np.float32)).to(device)
    for name in PHOTO_IMAGES:
        a = PHOTO[f"{name}.in"]
        for op in range(4):
            for k, f in enumerate(PHOTO[f"factors{op}"]):
                got = photo_step(up(a), op, float(f)).cpu().numpy()
                assert np.array_equal(got, PHOTO[f"{name}.op{op}.{k}"].astype(np.float32)), (name, op, f)
        for k in range(len(PHOTO["chain_orders"])):
            x = up(a)
            for op in PHOTO["chain_orders"][k]:
                photo_step(x, int(op), float(PHOTO["chain_factors"][k][int(op)]))
            assert np.array_equal(x.cpu().numpy(), PHOTO[f"{name}.chain{k}"].astype(np.float32)), (name, "chain", k)
    rs = np.random.RandomState(6)
    a = rs.randint(0, 256, size=(200, 312, 3)).astype(np.uint8)
    for op, f in ((0, 0.731), (0, 1.377), (1, 0.68), (1, 1.22), (2, 0.9), (2, 1.4), (3, 0.121), (3, -0.07)):
        assert np.array_equal(photo_step(up(a), op, f).cpu().numpy(), A.OPS[op](a, f).astype(np.float32)), (op, f)


@pytest.mark.gpu
def test_eraser(device):
    from craft_amd.augment import erase
    r = np.random.RandomState(4)
    img = r.randint(0, 256, size=(32, 40, 3)).astype(np.float32)
    e = erase(torch.from_numpy(img.copy()).to(device), [(5, 3, 10, 7), (30, 25, 50, 50)], (1.0, 2.0, 3.0)).cpu().numpy()
    ref = img.copy()
    ref[3:10, 5:15] = (1.0, 2.0, 3.0)
    ref[25:75, 30:80] = (1.0, 2.0, 3.0)
    assert np.array_equal(e, ref)
    # eraser_transform end to end against the REFERENCE'S OWN METHODS (tests/golden/augment_ref.npz: FlowAugmentor / SparseFlowAugmentor
    # .eraser_transform compiled out of augmentor.py's AST and run on seeded draws, tools/make_golden_augment.py): same seed -> same
    # rectangles, and the float64 mean colour TRUNCATED by the assignment into the uint8 image
    from craft_amd.augment import FlowAugmentor, SparseFlowAugmentor
    a1, a2 = AUGREF["erase.img1"], AUGREF["erase.img2"]
    fired = 0
    for kind, cls in (("FlowAugmentor", FlowAugmentor), ("SparseFlowAugmentor", SparseFlowAugmentor)):
        for seed in AUGREF["erase.seeds"].tolist():
            want = AUGREF[f"erase.{kind}.{seed}"]
            fired += int(not np.array_equal(want, a2))
            np.random.seed(seed)
            aug = cls("chairs", (96, 128))
            g1, g2 = aug.eraser_transform(torch.from_numpy(a1.astype(np.float32)).to(device), torch.from_numpy(a2.astype(np.float32)).to(device))
            assert np.array_equal(g2.cpu().numpy(), want.astype(np.float32)), (kind, seed)
            assert np.array_equal(g1.cpu().numpy(), a1.astype(np.float32))
    assert fired >= 4, "the fixture must contain draws that erase something"


@pytest.mark.gpu
def test_sparse_flow_resize_matches_the_reference_method(device):
    """craft_aug_sparse (resize part: crop = the whole resized map) against SparseFlowAugmentor.resize_sparse_flow_map run from the
    reference's own source (tests/golden/augment_ref.npz): which target pixels are valid, and their flow."""
    from craft_amd.augment import sparse_resize_crop
    flow, valid = AUGREF["sparse.flow"], AUGREF["sparse.valid"]
    H, W = valid.shape
    for k, (fx, fy) in enumerate(AUGREF["sparse.scales"].tolist()):
        want_f, want_v = AUGREF[f"sparse.{k}.flow"], AUGREF[f"sparse.{k}.valid"]
        Hs, Ws = int(round(H * fy)), int(round(W * fx))
        assert want_v.shape == (Hs, Ws)
        f, v = sparse_resize_crop(torch.from_numpy(flow).to(device), torch.from_numpy(valid).to(device), (Hs, Ws), 0, 0, fx, fy, False)
        assert np.array_equal(v.cpu().numpy(), want_v.astype(np.float32)), (fx, fy)
        # (the reference multiplies float32 flow by the float64 scale and stores float32: one rounding; the kernel multiplies in float32)
        assert np.allclose(f.cpu().numpy(), want_f, rtol=2e-7, atol=0), (fx, fy)


@pytest.mark.gpu
def test_spatial_transforms_without_resize_match_the_reference_methods(device):
    """FlowAugmentor / SparseFlowAugmentor.spatial_transform with spatial_aug_prob = 0 (the branch that calls cv2.resize is never entered:
    the reference's own methods run in numpy, tools/make_golden_augment.py): the scale / stretch / flip / crop DRAW ORDER, the flips with
    their flow signs, the dense crop and the sparse augmentor's crop with margins -- bit for bit."""
    from craft_amd.augment import FlowAugmentor, SparseFlowAugmentor
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.float32)).to(device)
    b1, b2, bf, bv = (AUGREF[f"spatial.{k}"] for k in ("img1", "img2", "flow", "valid"))
    crop = tuple(int(v) for v in AUGREF["spatial.crop"])
    flipped = 0
    for sd in AUGREF["spatial.seeds"].tolist():
        np.random.seed(sd)
        o1, o2, of = FlowAugmentor("chairs", crop, spatial_aug_prob=0.0).spatial_transform(up(b1), up(b2), up(bf))
        for got, key in ((o1, "img1"), (o2, "img2"), (of, "flow")):
            assert np.array_equal(got.cpu().numpy(), AUGREF[f"spatial.dense.{sd}.{key}"].astype(np.float32)), ("dense", sd, key)
        np.random.seed(sd)
        o1, o2, of, ov = SparseFlowAugmentor("kitti", crop, spatial_aug_prob=0.0, do_flip=True).spatial_transform(up(b1), up(b2), up(bf), up(bv))
        for got, key in ((o1, "img1"), (o2, "img2"), (of, "flow"), (ov, "valid")):
            assert np.array_equal(got.cpu().numpy(), AUGREF[f"spatial.sparse.{sd}.{key}"].astype(np.float32)), ("sparse", sd, key)
        flipped += int(np.sign(of.cpu().numpy()[..., 0]).sum() != np.sign(bf[..., 0]).sum())
    assert flipped >= 1


def _np_gaussian_blur(img, K, sigma):
    """cv2.GaussianBlur(img, (K, K), sigma) restated: getGaussianKernel for sigma > 0, separable, BORDER_REFLECT_101, float arithmetic."""
    x = np.arange(K, dtype=np.float64) - 0.5 * (K - 1)
    w = np.exp(-x * x / (2.0 * sigma * sigma))
    w /= w.sum()
    r = K // 2
    p = np.pad(img.astype(np.float64), ((r, r), (r, r), (0, 0)), mode="reflect")           # numpy 'reflect' == cv2 BORDER_REFLECT_101
    H, W = img.shape[:2]
    rows = sum(w[d] * p[:, d:d + W] for d in range(K))
    return sum(w[d] * rows[d:d + H] for d in range(K))


@pytest.mark.gpu
@pytest.mark.parametrize("K,sigma,shape", [(5, 1.3, (37, 53, 3)), (3, 0.6, (8, 9, 3)), (9, 2.5, (40, 33, 3)), (5, 1.0, (3, 2, 1)), (1, 1.0, (6, 7, 3))])
def test_gaussian_blur(device, K, sigma, shape):
    from craft_amd.augment import gaussian_blur
    rng = np.random.RandomState(K * 100 + shape[0])
    img = rng.randint(0, 256, size=shape).astype(np.float32)
    got = gaussian_blur(torch.from_numpy(img).to(device), K, sigma).cpu().numpy()
    ref = _np_gaussian_blur(img, K, sigma)
    assert got.shape == img.shape and np.array_equal(got, np.rint(got)) and got.min() >= 0 and got.max() <= 255
    # integer levels: equal to the rounded float reference except where the float sum sits within 1e-3 of a .5 boundary
    d = np.abs(got - np.clip(np.rint(ref), 0, 255))
    near_half = np.abs(ref - np.floor(ref) - 0.5) < 1e-3
    assert d[~near_half].max(initial=0.0) == 0.0 and d.max(initial=0.0) <= 1.0
    if K == 1:
        assert np.array_equal(got, img)


@pytest.mark.gpu
def test_flow_augmentor_blur_branch(device):
    """blur_sigma > 0 (augmentor.py:195-198): both frames are blurred after the spatial transform / shift, the flow is not; the random
    draws are those of the run without blur (cv2.GaussianBlur draws nothing)."""
    from craft_amd.augment import FlowAugmentor, gaussian_blur
    rng = np.random.RandomState(3)
    img1 = torch.from_numpy(rng.randint(0, 256, size=(160, 200, 3)).astype(np.float32)).to(device)
    img2 = torch.from_numpy(rng.randint(0, 256, size=(160, 200, 3)).astype(np.float32)).to(device)
    flow = torch.from_numpy(rng.randn(160, 200, 2).astype(np.float32)).to(device)
    outs = []
    for sig in (-1, 1.5):
        random.seed(11); np.random.seed(11)
        outs.append(FlowAugmentor("chairs", (96, 128), blur_kernel=5, blur_sigma=sig, shift_prob=0.5)(img1, img2, flow))
    (a1, a2, af, av), (b1, b2, bf, bv) = outs
    assert torch.equal(af, bf) and (av is None) == (bv is None)
    assert torch.equal(b1, gaussian_blur(a1, 5, 1.5)) and torch.equal(b2, gaussian_blur(a2, 5, 1.5))
    assert not torch.equal(a1, b1)
    with pytest.raises(ValueError, match="odd"):
        FlowAugmentor("chairs", (96, 128), blur_kernel=4, blur_sigma=1.0)
