"""TEST INFRASTRUCTURE -- CPU oracle of the photometric augmentation.  Only tests/ may import this.

The reference's ``FlowAugmentor.color_transform`` (core/utils/augmentor.py:104, :111-123) is ``torchvision.transforms.ColorJitter`` applied
to ``PIL.Image.fromarray(uint8 array)``.  torchvision is a requirements.txt dependency of the reference (unpinned, absent from this image);
on PIL images its four operations are thin wrappers over Pillow (torchvision/transforms/_functional_pil.py):
  adjust_brightness = ImageEnhance.Brightness(img).enhance(f)      adjust_contrast   = ImageEnhance.Contrast(img).enhance(f)
  adjust_saturation = ImageEnhance.Color(img).enhance(f)           adjust_hue: h, s, v = img.convert("HSV").split();
                                                                     h += uint8(int32(f * 255)) (wrap-around); merge -> convert("RGB")
Pillow IS in this image (12.2.0), so the restatement below -- Pillow's 8-bit arithmetic written out in numpy (libImaging/Blend.c,
Convert.c: rgb2hsv_row / hsv2rgb, the "L" conversion) -- is PINNED: tools/make_golden_photo.py produces tests/golden/photo_pil.npz with
Pillow itself, tests/test_augment.py::test_photo_oracle_matches_pillow_fixture holds these functions to it bit for bit (and to Pillow
directly on larger random images when it is importable), and the HIP kernel k_aug_photo is held to both.
"""
import numpy as np


def gray_L(a: np.ndarray) -> np.ndarray:
    """Pillow's RGB -> "L" (Convert.c rgb2l): (R*19595 + G*38470 + B*7471 + 0x8000) >> 16."""
    a = a.astype(np.int64)
    return ((a[..., 0] * 19595 + a[..., 1] * 38470 + a[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def contrast_mean(a: np.ndarray) -> int:
    """ImageEnhance.Contrast.__init__: int(ImageStat.Stat(image.convert("L")).mean[0] + 0.5)."""
    return int(gray_L(a).astype(np.float64).sum() / gray_L(a).size + 0.5)


def blend(deg: np.ndarray, img: np.ndarray, alpha: float) -> np.ndarray:
    """ImagingBlend(degenerate, image, alpha) (Blend.c): single-precision in1 + alpha * (in2 - in1); inside [0, 1] the result is
    truncated to uint8, outside it is clipped to [0, 255] first."""
    alpha = np.float32(alpha)
    d, i = deg.astype(np.int32), img.astype(np.int32)
    t = d.astype(np.float32) + (alpha * (i - d).astype(np.float32)).astype(np.float32)
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def adjust_brightness(a, f):
    return blend(np.zeros_like(a), a, f)


def adjust_contrast(a, f):
    return blend(np.full_like(a, contrast_mean(a)), a, f)


def adjust_saturation(a, f):
    return blend(np.repeat(gray_L(a)[..., None], 3, -1), a, f)


def hue_shift(f: float) -> int:
    """torchvision's adjust_hue on PIL images: np_h += np.int32(hue_factor * 255).astype(np.uint8)  (truncation toward zero, mod 256)."""
    return int(np.int32(f * 255).astype(np.uint8))


def rgb_to_hsv8(a: np.ndarray):
    """Convert.c rgb2hsv_row: float quotients, the sums with the double literals 2.0 / 4.0 in double, h = fmod(h / 6.0 + 1.0, 1.0) in double
    stored to float, (int)(x * 255.0) truncation."""
    f64 = np.float64
    r, g, b = (a[..., k].astype(np.int32) for k in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = cr / maxc.astype(np.float32)
        rc, gc, bc = ((maxc - c).astype(np.float32) / cr for c in (r, g, b))
        h = np.where(r == maxc, (bc - gc).astype(f64), np.where(g == maxc, 2.0 + rc.astype(f64) - bc.astype(f64),
                                                                 4.0 + gc.astype(f64) - rc.astype(f64))).astype(np.float32)
        h = np.fmod(h.astype(f64) / 6.0 + 1.0, 1.0).astype(np.float32)
        uh = np.clip((h.astype(f64) * 255.0).astype(np.int64), 0, 255)
        us = np.clip((s.astype(f64) * 255.0).astype(np.int64), 0, 255)
    grey = maxc == minc
    return np.where(grey, 0, uh), np.where(grey, 0, us), maxc.astype(np.int64)


def hsv8_to_rgb(uh, us, uv) -> np.ndarray:
    """Convert.c hsv2rgb: i = floor(h * 6.0 / 255.0), f and fs stored to float, p / q / t = round(...) (half away from zero) in double."""
    f64 = np.float64
    hh = uh.astype(np.float32).astype(f64) * 6.0 / 255.0
    i = np.floor(hh).astype(np.int64)
    ff = (hh - i.astype(np.float32).astype(f64)).astype(np.float32).astype(f64)
    fs = (us.astype(np.float32).astype(f64) / 255.0).astype(np.float32).astype(f64)
    v = uv.astype(f64)
    rnd = lambda x: np.floor(x + 0.5)                     # (arguments are >= 0)
    p = np.clip(rnd(v * (1.0 - fs)), 0, 255).astype(np.int64)
    q = np.clip(rnd(v * (1.0 - fs * ff)), 0, 255).astype(np.int64)
    t = np.clip(rnd(v * (1.0 - fs * (1.0 - ff))), 0, 255).astype(np.int64)
    sect = i % 6
    R = np.choose(sect, [uv, q, p, p, t, uv])
    G = np.choose(sect, [t, uv, uv, q, p, p])
    B = np.choose(sect, [p, p, t, uv, uv, q])
    z = us == 0
    return np.stack([np.where(z, uv, R), np.where(z, uv, G), np.where(z, uv, B)], -1).astype(np.uint8)


def adjust_hue(a, f):
    uh, us, uv = rgb_to_hsv8(a)
    return hsv8_to_rgb((uh + hue_shift(f)) % 256, us, uv)


OPS = (adjust_brightness, adjust_contrast, adjust_saturation, adjust_hue)      # ColorJitter's fn_id order (transforms.py: 0 b, 1 c, 2 s, 3 h)


def color_jitter(a: np.ndarray, order, factors) -> np.ndarray:
    """ColorJitter.forward for a drawn permutation ``order`` of (0, 1, 2, 3) and the four drawn factors."""
    for op in order:
        a = OPS[int(op)](a, factors[int(op)])
    return a
