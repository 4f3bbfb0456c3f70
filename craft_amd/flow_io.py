"""Flow / image file formats of the evaluation harness.

Behaviour follows ``core/utils/frame_utils.py`` of the reference (cited per function); OpenCV is not available here,
so the 16-bit KITTI PNGs go through a small self-contained PNG codec (``_png_read`` / ``_png_write``: zlib + the five
scan-line filters, non-interlaced grey / RGB(A), 8 or 16 bits per sample).

* ``.flo``  Middlebury: float32 tag 202021.25, int32 width, int32 height, then H*W*(u, v) float32, row-major,
  little-endian (frame_utils.py:12-31, :70-99).
* ``.pfm``  "PF"/"Pf" header, dims, scale (sign = endianness), rows stored bottom-up (frame_utils.py:33-68).
* KITTI flow PNG: 16-bit RGB = (u*64 + 2^15, v*64 + 2^15, valid) (frame_utils.py:102-108, :116-120);
  KITTI disparity PNG: 16-bit grey / 256 (frame_utils.py:110-114).
"""
from __future__ import annotations

import os
import re
import struct
import zlib
from typing import Tuple

import numpy as np

FLO_TAG = 202021.25


# ------------------------------------------------------------------------------------------------
# .flo
# ------------------------------------------------------------------------------------------------
def read_flo(path: str) -> np.ndarray:
    """-> float32 [H, W, 2]; raises on a bad tag or a truncated file (the reference prints and returns None)."""
    with open(path, "rb") as f:
        head = f.read(12)
        if len(head) != 12:
            raise ValueError(f"{path}: truncated .flo header")
        tag, w, h = struct.unpack("<fii", head)
        if tag != FLO_TAG:
            raise ValueError(f"{path}: magic number {tag!r} is not {FLO_TAG} (invalid .flo file)")
        if w <= 0 or h <= 0:
            raise ValueError(f"{path}: bad dimensions {w} x {h}")
        data = np.frombuffer(f.read(8 * w * h), dtype="<f4")
    if data.size != 2 * w * h:
        raise ValueError(f"{path}: expected {2 * w * h} floats, found {data.size}")
    return data.reshape(h, w, 2).astype(np.float32)


def write_flo(path: str, uv: np.ndarray, v: np.ndarray | None = None) -> None:
    """uv [H, W, 2] (or u, v as two [H, W] arrays) -> .flo (frame_utils.py:70-99)."""
    if v is None:
        uv = np.asarray(uv)
        if uv.ndim != 3 or uv.shape[2] != 2:
            raise ValueError("write_flo expects [H, W, 2]")
        u, v = uv[:, :, 0], uv[:, :, 1]
    else:
        u, v = np.asarray(uv), np.asarray(v)
    if u.shape != v.shape or u.ndim != 2:
        raise ValueError("u and v must be two [H, W] arrays of the same shape")
    h, w = u.shape
    inter = np.empty((h, w, 2), dtype="<f4")
    inter[..., 0], inter[..., 1] = u, v
    with open(path, "wb") as f:
        f.write(struct.pack("<fii", FLO_TAG, w, h))
        f.write(inter.tobytes())


# ------------------------------------------------------------------------------------------------
# .pfm
# ------------------------------------------------------------------------------------------------
def read_pfm(path: str) -> np.ndarray:
    """-> float32 [H, W, 3] ("PF") or [H, W] ("Pf"), top row first (frame_utils.py:33-68)."""
    with open(path, "rb") as f:
        header = f.readline().rstrip()
        if header == b"PF":
            color = True
        elif header == b"Pf":
            color = False
        else:
            raise ValueError(f"{path}: not a PFM file")
        m = re.match(rb"^(\d+)\s(\d+)\s$", f.readline())
        if not m:
            raise ValueError(f"{path}: malformed PFM header")
        width, height = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        data = np.frombuffer(f.read(), dtype=endian + "f4")
    shape = (height, width, 3) if color else (height, width)
    if data.size != int(np.prod(shape)):
        raise ValueError(f"{path}: PFM payload has {data.size} floats, expected {int(np.prod(shape))}")
    return np.flipud(data.reshape(shape)).astype(np.float32)


def write_pfm(path: str, img: np.ndarray, little_endian: bool = True) -> None:
    img = np.asarray(img, dtype=np.float32)
    color = img.ndim == 3
    if color and img.shape[2] != 3:
        raise ValueError("colour PFM needs 3 channels")
    h, w = img.shape[:2]
    with open(path, "wb") as f:
        f.write(b"PF\n" if color else b"Pf\n")
        f.write(f"{w} {h}\n".encode())
        f.write(b"-1.0\n" if little_endian else b"1.0\n")
        f.write(np.flipud(img).astype("<f4" if little_endian else ">f4").tobytes())


# ------------------------------------------------------------------------------------------------
# PNG (8 / 16 bit, grey / RGB / RGBA, non-interlaced)
# ------------------------------------------------------------------------------------------------
_PNG_SIG = b"\x89PNG\r\n\x1a\n"
_CHANNELS = {0: 1, 2: 3, 4: 2, 6: 4}


def _png_read(path: str) -> np.ndarray:
    """-> uint8 / uint16 array [H, W] or [H, W, C] in file channel order."""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:8] != _PNG_SIG:
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, ihdr = 8, [], None
    while pos + 8 <= len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
    if ihdr is None:
        raise ValueError(f"{path}: PNG without IHDR")
    w, h, depth, ctype, _, _, interlace = ihdr
    if interlace or depth not in (8, 16) or ctype not in _CHANNELS:
        raise ValueError(f"{path}: unsupported PNG (depth {depth}, colour type {ctype}, interlace {interlace})")
    ch = _CHANNELS[ctype]
    bpp = ch * depth // 8                      # bytes per pixel = filter distance
    stride = w * bpp
    data = zlib.decompress(b"".join(idat))
    if len(data) != h * (stride + 1):
        raise ValueError(f"{path}: PNG payload size mismatch")
    rows = np.frombuffer(data, dtype=np.uint8).reshape(h, stride + 1)
    out = np.empty((h, stride), dtype=np.uint8)
    # unfiltering is sequential along a scan line (Average / Paeth): done by the extension's host helper, not per byte in Python
    import ctypes
    from .hip import load
    fn = load().craft_png_unfilter
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    if fn(rows.ctypes.data, h, stride, bpp, out.ctypes.data) != 0:
        raise ValueError(f"{path}: bad PNG filter type")
    if depth == 16:
        arr = out.reshape(h, w, ch, 2)
        arr = (arr[..., 0].astype(np.uint16) << 8) | arr[..., 1].astype(np.uint16)     # big-endian samples
    else:
        arr = out.reshape(h, w, ch)
    return arr[..., 0] if ch == 1 else arr


def _png_write(path: str, arr: np.ndarray) -> None:
    """uint8 / uint16 [H, W] or [H, W, C] (C = 1, 3, 4) -> PNG, filter 0, zlib level 6."""
    arr = np.asarray(arr)
    if arr.dtype not in (np.uint8, np.uint16):
        raise ValueError("PNG samples must be uint8 or uint16")
    if arr.ndim == 2:
        arr = arr[..., None]
    h, w, ch = arr.shape
    ctype = {1: 0, 3: 2, 4: 6}.get(ch)
    if ctype is None:
        raise ValueError("PNG needs 1, 3 or 4 channels")
    depth = 16 if arr.dtype == np.uint16 else 8
    payload = arr.astype(">u2").tobytes() if depth == 16 else arr.tobytes()
    stride = w * ch * depth // 8
    lines = np.frombuffer(payload, dtype=np.uint8).reshape(h, stride)
    raw = np.concatenate([np.zeros((h, 1), dtype=np.uint8), lines], axis=1).tobytes()

    def chunk(typ: bytes, body: bytes) -> bytes:
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(_PNG_SIG)
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))


# ------------------------------------------------------------------------------------------------
# KITTI
# ------------------------------------------------------------------------------------------------
def read_flow_kitti(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """-> (flow float32 [H, W, 2], valid float32 [H, W])   (frame_utils.py:102-108; file channels R, G, B =
    u, v, valid -- the reference reads BGR with OpenCV and reverses)."""
    png = _png_read(path)
    if png.ndim != 3 or png.shape[2] < 3:
        raise ValueError(f"{path}: KITTI flow must be a 3-channel PNG")
    png = png[..., :3].astype(np.float32)
    return (png[..., :2] - 2 ** 15) / 64.0, png[..., 2]


def write_flow_kitti(path: str, uv: np.ndarray) -> None:
    """flow [H, W, 2] -> 16-bit RGB PNG (u*64 + 2^15, v*64 + 2^15, 1), values truncated as ``astype(uint16)`` does
    (frame_utils.py:116-120)."""
    uv = 64.0 * np.asarray(uv, dtype=np.float64) + 2 ** 15
    valid = np.ones(uv.shape[:2] + (1,))
    _png_write(path, np.concatenate([uv, valid], axis=-1).astype(np.uint16))


def read_disp_kitti(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """-> (flow [H, W, 2] = (-disp, 0), valid bool [H, W])   (frame_utils.py:110-114)."""
    disp = _png_read(path).astype(np.float64) / 256.0
    if disp.ndim == 3:
        disp = disp[..., 0]
    return np.stack([-disp, np.zeros_like(disp)], -1).astype(np.float32), disp > 0.0


# ------------------------------------------------------------------------------------------------
# images and the dispatching reader
# ------------------------------------------------------------------------------------------------
def read_image(path: str) -> np.ndarray:
    """png / jpg / ppm -> uint8 [H, W, 3] (grey images are tiled, alpha dropped: datasets.py:112-119)."""
    ext = os.path.splitext(path)[-1].lower()
    if ext == ".png":
        try:
            img = _png_read(path)
            if img.dtype == np.uint16:
                img = (img >> 8).astype(np.uint8)
        except ValueError as e:
            if "unsupported PNG" not in str(e):
                raise
            from PIL import Image                      # palette / 1-2-4 bit / interlaced files: Pillow expands them
            img = np.array(Image.open(path).convert("RGB"))
    else:
        from PIL import Image
        img = np.array(Image.open(path))
    img = img.astype(np.uint8)
    if img.ndim == 3 and img.shape[2] == 2:            # grey + alpha
        img = img[..., 0]
    if img.ndim == 2:
        img = np.tile(img[..., None], (1, 1, 3))
    return img[..., :3]


def write_image(path: str, img: np.ndarray) -> None:
    """uint8 [H, W, 3] -> .png (own codec) or .ppm (P6)."""
    img = np.asarray(img, dtype=np.uint8)
    ext = os.path.splitext(path)[-1].lower()
    if ext == ".png":
        _png_write(path, img)
    elif ext == ".ppm":
        h, w = img.shape[:2]
        with open(path, "wb") as f:
            f.write(f"P6\n{w} {h}\n255\n".encode())
            f.write(img[..., :3].tobytes())
    else:
        raise ValueError(f"unsupported image extension {ext}")


def read_gen(path: str):
    """The reference's dispatching reader (frame_utils.py:123-137): images by extension, ``.flo``, ``.pfm`` (colour PFM
    drops its last channel), ``.bin``/``.raw`` via ``np.load``; anything else -> []."""
    ext = os.path.splitext(path)[-1].lower()
    if ext in (".png", ".jpeg", ".ppm", ".jpg"):
        return read_image(path)
    if ext in (".bin", ".raw"):
        return np.load(path)
    if ext == ".flo":
        return read_flo(path)
    if ext == ".pfm":
        flow = read_pfm(path)
        return flow if flow.ndim == 2 else flow[:, :, :-1]
    return []
