"""Input pipeline on the GPU (reference core/utils/augmentor.py).

* random_shift: bit-for-bit against outputs of the reference's own function (tests/golden/harness.npz 'shift.*'), with (dx, dy)
  re-drawn by our sampler from the same seeds -- so the draw order is pinned too (CPU part) and the kernel (GPU part).
* ColorJitter's four operations: bit for bit against outputs of Pillow itself (tests/golden/photo_pil.npz; torchvision's PIL path wraps
  Pillow) -- the oracle (CPU part) and the kernel (GPU part).
* spatial gather (resize -> flips -> crop), eraser: against numpy restatements written here (cv2 is not in this image: its 8-bit
  fixed-point resize is not pinned).
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
    # eraser_transform end to end against the reference's statements (augmentor.py:127-139, pure numpy: restated here on a uint8 array --
    # the float64 mean colour is TRUNCATED by the assignment into the uint8 image), same seed -> same rectangles
    from craft_amd.augment import FlowAugmentor
    a1 = r.randint(0, 256, size=(120, 150, 3)).astype(np.uint8)
    a2 = r.randint(0, 256, size=(120, 150, 3)).astype(np.uint8)
    for seed in (0, 1, 2, 3, 5):
        np.random.seed(seed)
        want = a2.copy()
        fired = np.random.rand() < 0.5
        if fired:
            mean_color = np.mean(want.reshape(-1, 3), axis=0)
            for _ in range(np.random.randint(1, 3)):
                x0, y0 = np.random.randint(0, 150), np.random.randint(0, 120)
                dx, dy = np.random.randint(50, 100), np.random.randint(50, 100)
                want[y0:y0 + dy, x0:x0 + dx, :] = mean_color
        np.random.seed(seed)
        aug = FlowAugmentor("chairs", (96, 128))
        g1, g2 = aug.eraser_transform(torch.from_numpy(a1.astype(np.float32)).to(device), torch.from_numpy(a2.astype(np.float32)).to(device))
        assert np.array_equal(g2.cpu().numpy(), want.astype(np.float32)), seed
        assert np.array_equal(g1.cpu().numpy(), a1.astype(np.float32))


@pytest.mark.gpu
def test_flow_augmentor_end_to_end(device):
    from craft_amd.augment import FlowAugmentor
    r = np.random.RandomState(6)
    H, W, crop = 120, 160, (64, 96)
    img1 = torch.from_numpy(r.randint(0, 256, size=(H, W, 3)).astype(np.float32)).to(device)
    img2 = torch.from_numpy(r.randint(0, 256, size=(H, W, 3)).astype(np.float32)).to(device)
    flow = torch.from_numpy((r.standard_normal((H, W, 2)) * 3).astype(np.float32)).to(device)
    aug = FlowAugmentor("chairs", crop, min_scale=-0.1, max_scale=1.0, do_flip=True, shift_prob=0.5, shift_sigmas=(16, 10))
    shifted = 0
    for sd in range(12):
        random.seed(sd); np.random.seed(sd)
        a, b, f, v = aug(img1, img2, flow)
        assert a.shape == (crop[0], crop[1], 3) and b.shape == a.shape and f.shape == (crop[0], crop[1], 2)
        assert float(a.min()) >= 0 and float(a.max()) <= 255 and torch.equal(a, a.round())
        assert torch.isfinite(f).all()
        if v is not None:
            shifted += 1
            assert v.shape == crop and set(np.unique(v.cpu().numpy())) <= {0.0, 1.0}
            assert float((f.abs().sum(-1) * (1 - v)).max()) == 0.0            # padded area: zero flow
    assert 0 < shifted < 12
    # same seed, same result (all randomness comes from the seeded module-level generators)
    random.seed(3); np.random.seed(3)
    x = aug(img1, img2, flow)
    random.seed(3); np.random.seed(3)
    y = aug(img1, img2, flow)
    assert all(torch.equal(p, q) for p, q in zip(x[:3], y[:3]))


@pytest.mark.gpu
@pytest.mark.parametrize("fx,hflip", [(1.0, False), (1.31, False), (0.77, True)])
def test_sparse_flow_resize(device, fx, hflip):
    """craft_aug_sparse vs a numpy restatement of SparseFlowAugmentor.resize_sparse_flow_map + flip + crop (augmentor.py:249-316),
    including numpy's last-writer-wins order on contested targets (fx < 1 makes many collisions)."""
    from craft_amd.augment import sparse_resize_crop
    r = np.random.RandomState(8)
    H, W = 50, 70
    flow = (r.standard_normal((H, W, 2)) * 6).astype(np.float32)
    valid = (r.random_sample((H, W)) > 0.5).astype(np.float32)
    fy = fx
    coords = np.stack(np.meshgrid(np.arange(W), np.arange(H)), axis=-1).reshape(-1, 2).astype(np.float32)
    fl, va = flow.reshape(-1, 2), valid.reshape(-1)
    c0, f0 = coords[va >= 1], fl[va >= 1]
    ht1, wd1 = int(round(H * fy)), int(round(W * fx))
    c1, f1 = c0 * [fx, fy], f0 * [fx, fy]
    xx, yy = np.round(c1[:, 0]).astype(np.int32), np.round(c1[:, 1]).astype(np.int32)
    v = (xx > 0) & (xx < wd1) & (yy > 0) & (yy < ht1)
    fimg, vimg = np.zeros([ht1, wd1, 2], np.float32), np.zeros([ht1, wd1], np.int32)
    fimg[yy[v], xx[v]] = f1[v]
    vimg[yy[v], xx[v]] = 1
    if hflip:
        fimg, vimg = fimg[:, ::-1] * [-1.0, 1.0], vimg[:, ::-1]
    crop = (min(30, ht1 - 3), min(40, wd1 - 4))
    y0, x0 = 2, 3
    ref_f, ref_v = fimg[y0:y0 + crop[0], x0:x0 + crop[1]], vimg[y0:y0 + crop[0], x0:x0 + crop[1]]
    gf, gv = sparse_resize_crop(torch.from_numpy(flow).to(device), torch.from_numpy(valid).to(device), crop, y0, x0, fx, fy, hflip)
    assert np.array_equal(gv.cpu().numpy(), ref_v.astype(np.float32))
    assert np.allclose(gf.cpu().numpy(), ref_f, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_sparse_augmentor_end_to_end(device):
    from craft_amd.augment import SparseFlowAugmentor
    r = np.random.RandomState(9)
    H, W, crop = 150, 400, (96, 256)
    img1 = torch.from_numpy(r.randint(0, 256, size=(H, W, 3)).astype(np.float32)).to(device)
    img2 = torch.from_numpy(r.randint(0, 256, size=(H, W, 3)).astype(np.float32)).to(device)
    flow = torch.from_numpy((r.standard_normal((H, W, 2)) * 3).astype(np.float32)).to(device)
    valid = torch.from_numpy((r.random_sample((H, W)) > 0.6).astype(np.float32)).to(device)
    aug = SparseFlowAugmentor("kitti", crop, min_scale=-0.2, max_scale=0.4, do_flip=False)
    for sd in range(6):
        random.seed(sd); np.random.seed(sd)
        a, b, f, v = aug(img1, img2, flow, valid)
        assert a.shape == (crop[0], crop[1], 3) and f.shape == (crop[0], crop[1], 2) and v.shape == crop
        assert 0.05 < float(v.mean()) < 0.6 and float((f.abs().sum(-1) * (1 - v)).max()) == 0.0


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
