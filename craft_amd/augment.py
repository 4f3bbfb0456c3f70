"""Training-time augmentation on the GPU: the reference's ``core/utils/augmentor.py`` with the
pixel work as HIP kernels (``csrc/kernels_augment.hip``) over images that already live in HBM.

``FlowAugmentor`` keeps the reference's constructor arguments, probabilities and ORDER of random draws (``np.random`` /
``random`` module-level generators, augmentor.py:80-204), so a seeded run draws the same scale / stretch / flips / crop /
eraser rectangles / shift as the reference would; the photometric factors are drawn the way torchvision's ``ColorJitter``
does (uniform factors, random order of the four operations) from ``np.random`` as well (torchvision uses torch's generator:
those draws cannot be replayed).  Pixel semantics: ``random_shift`` is pinned bit-for-bit by a fixture produced by the
reference's own function (tests/golden/harness.npz); the four ColorJitter operations are Pillow's 8-bit arithmetic bit for bit, pinned
by a fixture Pillow itself produced (tests/golden/photo_pil.npz; torchvision's PIL path is a thin wrapper over Pillow); resize / blur
restate cv2.INTER_LINEAR / cv2.GaussianBlur in float and are checked against numpy restatements in the tests -- cv2 is absent from
this image, so its 8-bit fixed-point rounding is NOT pinned (documented gap).  ``SparseFlowAugmentor`` (KITTI): the sparse flow map is moved, not interpolated (``craft_aug_sparse``; exact integer logic, checked against a numpy restatement of resize_sparse_flow_map).
"""
from __future__ import annotations

import random
from typing import Optional, Tuple

import numpy as np
import torch

from .hip import call


def _f(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("craft_amd.augment works on GPU tensors (no CPU fallback)")
    return t.float().contiguous()


def draw_shift(shift_sigmas=(16, 10)) -> Tuple[int, int]:
    """The (dx, dy) draw of random_shift (augmentor.py:17-30), same generator calls in the same order."""
    u, v = shift_sigmas
    if random.random() > 0.5:
        dx, dy = np.random.laplace(0, u / 4), np.random.laplace(0, v)
    else:
        dx, dy = np.random.laplace(0, u), np.random.laplace(0, v / 4)
    return (int(dx) // 2) * 2, (int(dy) // 2) * 2


def random_shift(img1: torch.Tensor, img2: torch.Tensor, flow: torch.Tensor, dx: int, dy: int):
    """augmentor.py:16-78 for a given even (dx, dy): HWC float images / flow on the GPU -> (img1a, img2a, flowa, valid[H, W])."""
    img1, img2, flow = _f(img1), _f(img2), _f(flow)
    H, W, _ = img1.shape
    o1, o2, of = torch.empty_like(img1), torch.empty_like(img2), torch.empty_like(flow)
    valid = torch.empty(H, W, device=img1.device, dtype=torch.float32)
    call("craft_aug_shift", img1, img2, flow, H, W, int(dx), int(dy), o1, o2, of, valid)
    return o1, o2, of, valid


def spatial(src: torch.Tensor, crop, y0: int, x0: int, fx: float = 1.0, fy: float = 1.0, do_resize: bool = False, hflip: bool = False,
            vflip: bool = False, is_flow: bool = False) -> torch.Tensor:
    src = _f(src)
    H, W, C = src.shape
    out = torch.empty(crop[0], crop[1], C, device=src.device, dtype=torch.float32)
    call("craft_aug_spatial", src, H, W, C, int(do_resize), float(fx), float(fy), int(hflip), int(vflip), int(y0), int(x0), crop[0], crop[1],
         int(is_flow), out)
    return out


def hue_shift(factor: float) -> int:
    """torchvision's adjust_hue on PIL images: ``np_h += np.int32(hue_factor * 255).astype(np.uint8)`` -- the integer added (mod 256) to the
    uint8 hue plane."""
    return int(np.int32(factor * 255).astype(np.uint8))


def photo_step(img: torch.Tensor, op: int, factor: float) -> torch.Tensor:
    """One ColorJitter operation in place on a float HWC image of integer levels (0 brightness, 1 contrast, 2 saturation, 3 hue): Pillow's
    8-bit arithmetic, bit for bit (tests/golden/photo_pil.npz)."""
    aux = 0.0
    if op == 1:       # ImageEnhance.Contrast: int(mean of the "L" image + 0.5), "L" = (R*19595 + G*38470 + B*7471 + 0x8000) >> 16
        q = img.round().to(torch.int64)
        L = (q[..., 0] * 19595 + q[..., 1] * 38470 + q[..., 2] * 7471 + 0x8000) >> 16
        aux = float(int(int(L.sum().item()) / L.numel() + 0.5))
    elif op == 3:
        aux = float(hue_shift(factor))
    call("craft_aug_photo", img, img.numel() // 3, int(op), float(factor), aux)
    return img


def erase(img: torch.Tensor, rects, mean_color) -> torch.Tensor:
    if not rects:
        return img
    H, W, _ = img.shape
    r = torch.tensor(rects, dtype=torch.int32, device=img.device).reshape(-1, 4)
    call("craft_aug_erase", img, H, W, r, r.shape[0], float(mean_color[0]), float(mean_color[1]), float(mean_color[2]))
    return img


def gaussian_blur(img: torch.Tensor, ksize: int, sigma: float) -> torch.Tensor:
    """cv2.GaussianBlur(img, (ksize, ksize), sigma) of an HWC image in 0..255 (augmentor.py:195-198): separable Gaussian weights of
    cv2.getGaussianKernel for sigma > 0, BORDER_REFLECT_101, result rounded to integer levels (the reference blurs uint8 arrays; cv2's
    fixed-point 8-bit path is not pinned: cv2 is absent from this image)."""
    img = _f(img)
    H, W, C = img.shape
    out = torch.empty_like(img)
    call("craft_aug_blur", img, H, W, C, int(ksize), float(sigma), out)
    return out


class FlowAugmentor:
    """Dense-flow augmentation (augmentor.py:80-204) on GPU tensors: ``__call__(img1, img2, flow)`` with HWC float (or uint8)
    images in 0..255 and flow [H, W, 2] -> (img1, img2, flow, valid or None), all cropped to ``crop_size``."""

    def __init__(self, ds_name, crop_size, min_scale=-0.2, max_scale=0.5, spatial_aug_prob=0.8, blur_kernel=5, blur_sigma=-1, do_flip=True,
                 shift_prob=0, shift_sigmas=(16, 10)):
        self.ds_name, self.crop_size = ds_name, tuple(crop_size)
        self.min_scale, self.max_scale, self.spatial_aug_prob = min_scale, max_scale, spatial_aug_prob
        self.stretch_prob, self.max_stretch = 0.8, 0.2
        self.do_flip, self.h_flip_prob, self.v_flip_prob = do_flip, 0.5, 0.1
        self.shift_prob, self.shift_sigmas = shift_prob, shift_sigmas
        self.jitter = dict(brightness=0.4, contrast=0.4, saturation=0.4, hue=0.5 / 3.14)
        self.asymmetric_color_aug_prob, self.eraser_aug_prob = 0.2, 0.5
        self.blur_kernel, self.blur_sigma = int(blur_kernel), float(blur_sigma)
        if self.blur_sigma > 0 and (self.blur_kernel < 1 or self.blur_kernel % 2 == 0 or self.blur_kernel > 31):
            raise ValueError(f"blur_kernel must be odd and in 1..31 (cv2.GaussianBlur needs an odd size), got {blur_kernel}")

    # -- photometric (augmentor.py:106-123) -----------------------------------------------------------------------------
    def _jitter_params(self):
        j = self.jitter
        order = list(np.random.permutation(4))
        fac = [np.random.uniform(max(0, 1 - j["brightness"]), 1 + j["brightness"]), np.random.uniform(max(0, 1 - j["contrast"]), 1 + j["contrast"]),
               np.random.uniform(max(0, 1 - j["saturation"]), 1 + j["saturation"]), np.random.uniform(-j["hue"], j["hue"])]
        return order, fac

    def color_transform(self, img1, img2):
        if np.random.rand() < self.asymmetric_color_aug_prob:
            for img in (img1, img2):
                order, fac = self._jitter_params()
                for op in order:
                    photo_step(img, int(op), fac[op])
        else:
            stack = torch.cat([img1, img2], dim=0)                 # one draw, one mean for both frames (image_stack)
            order, fac = self._jitter_params()
            for op in order:
                photo_step(stack, int(op), fac[op])
            H = img1.shape[0]
            img1, img2 = stack[:H].contiguous(), stack[H:].contiguous()
        return img1, img2

    def eraser_transform(self, img1, img2, bounds=(50, 100)):
        ht, wd = img1.shape[:2]
        if np.random.rand() < self.eraser_aug_prob:
            # np.mean(img2.reshape(-1, 3), axis=0) assigned into a uint8 array (augmentor.py:129-137): the float64 mean is truncated
            mean_color = img2.reshape(-1, 3).double().mean(dim=0).floor().tolist()
            rects = []
            for _ in range(np.random.randint(1, 3)):
                x0, y0 = np.random.randint(0, wd), np.random.randint(0, ht)
                dx, dy = np.random.randint(bounds[0], bounds[1]), np.random.randint(bounds[0], bounds[1])
                rects.append((x0, y0, dx, dy))
            erase(img2, rects, mean_color)
        return img1, img2

    # -- spatial (augmentor.py:141-193) ---------------------------------------------------------------------------------
    def spatial_params(self, ht, wd):
        min_scale = np.maximum((self.crop_size[0] + 8) / float(ht), (self.crop_size[1] + 8) / float(wd))
        scale = 2 ** np.random.uniform(self.min_scale, self.max_scale)
        scale_x = scale_y = scale
        if np.random.rand() < self.stretch_prob:
            scale_x *= 2 ** np.random.uniform(-self.max_stretch, self.max_stretch)
            scale_y *= 2 ** np.random.uniform(-self.max_stretch, self.max_stretch)
        scale_x, scale_y = np.clip(scale_x, min_scale, None), np.clip(scale_y, min_scale, None)
        do_resize = np.random.rand() < self.spatial_aug_prob
        hs, ws = (int(round(ht * scale_y)), int(round(wd * scale_x))) if do_resize else (ht, wd)
        hflip = vflip = False
        if self.do_flip:
            hflip = np.random.rand() < self.h_flip_prob
            vflip = np.random.rand() < self.v_flip_prob
        y0 = np.random.randint(0, hs - self.crop_size[0])
        x0 = np.random.randint(0, ws - self.crop_size[1])
        return dict(fx=float(scale_x), fy=float(scale_y), do_resize=bool(do_resize), hflip=bool(hflip), vflip=bool(vflip), y0=int(y0), x0=int(x0))

    def spatial_transform(self, img1, img2, flow):
        p = self.spatial_params(img1.shape[0], img1.shape[1])
        return (spatial(img1, self.crop_size, **p), spatial(img2, self.crop_size, **p), spatial(flow, self.crop_size, is_flow=True, **p))

    def __call__(self, img1, img2, flow):
        img1, img2, flow = _f(img1).clone(), _f(img2).clone(), _f(flow)
        img1, img2 = self.color_transform(img1, img2)
        img1, img2 = self.eraser_transform(img1, img2)
        img1, img2, flow = self.spatial_transform(img1, img2, flow)
        valid: Optional[torch.Tensor] = None
        if self.shift_prob > 0 and random.random() < self.shift_prob:
            dx, dy = draw_shift(self.shift_sigmas)
            img1, img2, flow, valid = random_shift(img1, img2, flow, dx, dy)
        if self.blur_sigma > 0:                                   # augmentor.py:195-198
            img1, img2 = gaussian_blur(img1, self.blur_kernel, self.blur_sigma), gaussian_blur(img2, self.blur_kernel, self.blur_sigma)
        return img1, img2, flow, valid


def sparse_resize_crop(flow: torch.Tensor, valid: torch.Tensor, crop, y0: int, x0: int, fx: float = 1.0, fy: float = 1.0, hflip: bool = False):
    """resize_sparse_flow_map (augmentor.py:249-281) + h-flip + crop on the GPU -> (flow [ch, cw, 2], valid [ch, cw])."""
    flow, valid = _f(flow), _f(valid)
    H, W, _ = flow.shape
    Hs, Ws = int(round(H * fy)), int(round(W * fx))
    owner = torch.empty(Hs * Ws, device=flow.device, dtype=torch.int32)
    of = torch.empty(crop[0], crop[1], 2, device=flow.device, dtype=torch.float32)
    ov = torch.empty(crop[0], crop[1], device=flow.device, dtype=torch.float32)
    call("craft_aug_sparse", flow, valid, H, W, float(fx), float(fy), int(hflip), int(y0), int(x0), crop[0], crop[1], owner, of, ov)
    return of, ov


class SparseFlowAugmentor(FlowAugmentor):
    """Sparse-flow augmentation (KITTI / HD1K; augmentor.py:207-328): symmetric photometric jitter only, eraser, one isotropic
    scale (no stretch), h-flip (if enabled), crop with margins; the flow map is moved pixel by pixel instead of interpolated.
    ``__call__(img1, img2, flow, valid)`` -> (img1, img2, flow, valid)."""

    def __init__(self, ds_name, crop_size, min_scale=-0.2, max_scale=0.5, spatial_aug_prob=0.8, do_flip=False, shift_prob=0, shift_sigmas=(16, 10)):
        super().__init__(ds_name, crop_size, min_scale, max_scale, spatial_aug_prob, do_flip=do_flip, shift_prob=shift_prob, shift_sigmas=shift_sigmas)
        self.jitter = dict(brightness=0.3, contrast=0.3, saturation=0.3, hue=0.3 / 3.14)

    def color_transform(self, img1, img2):
        stack = torch.cat([img1, img2], dim=0)
        order, fac = self._jitter_params()
        for op in order:
            photo_step(stack, int(op), fac[op])
        H = img1.shape[0]
        return stack[:H].contiguous(), stack[H:].contiguous()

    def eraser_transform(self, img1, img2):
        return super().eraser_transform(img1, img2, bounds=(50, 100))

    def spatial_transform(self, img1, img2, flow, valid):
        """augmentor.py:290-330: one isotropic scale (clipped so that the crop fits), resize with probability spatial_aug_prob, h-flip, a crop
        drawn with margins and clipped back into the frame; same ``np.random`` calls in the same order."""
        ht, wd = img1.shape[:2]
        min_scale = np.maximum((self.crop_size[0] + 1) / float(ht), (self.crop_size[1] + 1) / float(wd))
        scale = 2 ** np.random.uniform(self.min_scale, self.max_scale)
        s = float(np.clip(scale, min_scale, None))
        do_resize = np.random.rand() < self.spatial_aug_prob
        fx = fy = s if do_resize else 1.0
        hs, ws = (int(round(ht * fy)), int(round(wd * fx)))
        hflip = bool(self.do_flip and np.random.rand() < 0.5)
        margin_y, margin_x = 20, 50
        y0 = np.random.randint(0, hs - self.crop_size[0] + margin_y)
        x0 = np.random.randint(-margin_x, ws - self.crop_size[1] + margin_x)
        y0 = int(np.clip(y0, 0, hs - self.crop_size[0]))
        x0 = int(np.clip(x0, 0, ws - self.crop_size[1]))
        a = spatial(img1, self.crop_size, y0, x0, fx, fy, do_resize, hflip, False)
        b = spatial(img2, self.crop_size, y0, x0, fx, fy, do_resize, hflip, False)
        if do_resize:
            f, v = sparse_resize_crop(flow, valid, self.crop_size, y0, x0, fx, fy, hflip)
        else:
            # augmentor.py:301-305: without a resize the flow map is NOT rebuilt pixel by pixel (no strict x > 0 / y > 0 test, flow at
            # invalid pixels kept as it is): h-flip (augmentor.py:307-312) and crop (:324-325) of the maps as they are
            ch, cw = self.crop_size
            if hflip:
                flow, valid = flow.flip(1) * torch.tensor([-1.0, 1.0], device=flow.device), valid.flip(1)
            f, v = flow[y0:y0 + ch, x0:x0 + cw].contiguous(), valid[y0:y0 + ch, x0:x0 + cw].contiguous()
        return a, b, f, v

    def __call__(self, img1, img2, flow, valid):
        img1, img2, flow, valid = _f(img1).clone(), _f(img2).clone(), _f(flow), _f(valid)
        img1, img2 = self.color_transform(img1, img2)
        img1, img2 = self.eraser_transform(img1, img2)
        return self.spatial_transform(img1, img2, flow, valid)
