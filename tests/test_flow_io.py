"""Formats, dataset walkers and the metric definitions of the evaluation harness -- CPU."""
import os
import struct
import zlib

import numpy as np
import pytest
import torch

from craft_amd import flow_io
from craft_amd.flow_datasets import KITTI, FlyingChairs, MpiSintel


def test_flo_golden_bytes_and_roundtrip(tmp_path):
    uv = np.arange(2 * 3 * 2, dtype=np.float32).reshape(2, 3, 2) - 2.5
    p = str(tmp_path / "a.flo")
    flow_io.write_flo(p, uv)
    raw = open(p, "rb").read()
    # Middlebury layout (frame_utils.py:12-31, :70-99): "PIEH" tag, int32 w, int32 h, interleaved (u, v) rows
    assert raw[:4] == b"PIEH" and struct.unpack("<f", raw[:4])[0] == 202021.25
    assert struct.unpack("<ii", raw[4:12]) == (3, 2)
    assert np.array_equal(np.frombuffer(raw[12:], "<f4"), uv.reshape(-1))
    assert np.array_equal(flow_io.read_flo(p), uv)
    flow_io.write_flo(p, uv[..., 0], uv[..., 1])          # (u, v) call form
    assert np.array_equal(flow_io.read_flo(p), uv)
    assert np.array_equal(flow_io.read_gen(p), uv)
    open(p, "wb").write(struct.pack("<fii", 1.0, 3, 2) + uv.tobytes())
    with pytest.raises(ValueError):
        flow_io.read_flo(p)
    open(p, "wb").write(raw[:-4])
    with pytest.raises(ValueError):
        flow_io.read_flo(p)


@pytest.mark.parametrize("little", [True, False])
@pytest.mark.parametrize("color", [True, False])
def test_pfm_roundtrip(tmp_path, little, color):
    rng = np.random.default_rng(0)
    img = rng.standard_normal((5, 7, 3) if color else (5, 7)).astype(np.float32)
    p = str(tmp_path / "a.pfm")
    flow_io.write_pfm(p, img, little_endian=little)
    assert np.array_equal(flow_io.read_pfm(p), img)
    # rows are stored bottom-up: the first stored row is the LAST image row
    raw = open(p, "rb").read().split(b"\n", 3)[3]
    first = np.frombuffer(raw[: img[-1].size * 4], "<f4" if little else ">f4").reshape(img[-1].shape)
    assert np.array_equal(first, img[-1])
    if color:
        assert np.array_equal(flow_io.read_gen(p), img[..., :2])      # frame_utils.py:131-136 drops the last channel


def _png_with_filters(arr, filters):
    """Reference PNG encoder for the tests: one given filter type per scan line (PNG spec section 9)."""
    h, w, ch = arr.shape
    depth = 16 if arr.dtype == np.uint16 else 8
    bpp = ch * depth // 8
    lines = np.frombuffer(arr.astype(">u2").tobytes() if depth == 16 else arr.tobytes(), np.uint8).reshape(h, -1).astype(np.int32)
    out = bytearray()
    prev = np.zeros(lines.shape[1], np.int32)
    for y in range(h):
        ft, cur = filters[y % len(filters)], lines[y]
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - a
        elif ft == 2:
            f = cur - prev
        elif ft == 3:
            f = cur - ((a + prev) >> 1)
        else:
            p = a + prev - c
            pa, pb, pc = abs(p - a), abs(p - prev), abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
            f = cur - pred
        out += bytes([ft]) + (f & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)
    ctype = {1: 0, 3: 2, 4: 6}[ch]
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(bytes(out))) + chunk(b"IEND", b""))


@pytest.mark.parametrize("dtype,ch", [(np.uint8, 3), (np.uint16, 3), (np.uint16, 1), (np.uint8, 4)])
def test_png_codec_all_filters(tmp_path, dtype, ch):
    rng = np.random.default_rng(1)
    arr = rng.integers(0, np.iinfo(dtype).max + 1, size=(11, 9, ch)).astype(dtype)
    p = str(tmp_path / "a.png")
    open(p, "wb").write(_png_with_filters(arr, [0, 1, 2, 3, 4]))
    got = flow_io._png_read(p)
    assert got.dtype == dtype and np.array_equal(got.reshape(arr.shape), arr)
    flow_io._png_write(p, arr)                          # own writer -> own reader
    assert np.array_equal(flow_io._png_read(p).reshape(arr.shape), arr)


def test_png_against_pillow(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(2)
    rgb = rng.integers(0, 256, size=(13, 17, 3)).astype(np.uint8)
    p = str(tmp_path / "pil.png")
    Image.fromarray(rgb).save(p)                         # Pillow picks its own filters
    assert np.array_equal(flow_io.read_image(p), rgb)
    flow_io.write_image(p, rgb)                          # our writer, Pillow's reader
    assert np.array_equal(np.array(Image.open(p)), rgb)
    grey = rng.integers(0, 256, size=(6, 5)).astype(np.uint8)
    Image.fromarray(grey).save(p)
    assert np.array_equal(flow_io.read_image(p), np.tile(grey[..., None], (1, 1, 3)))    # datasets.py:112-115
    ppm = str(tmp_path / "x.ppm")
    flow_io.write_image(ppm, rgb)
    assert np.array_equal(flow_io.read_image(ppm), rgb)


def test_kitti_flow_png(tmp_path):
    rng = np.random.default_rng(3)
    uv = (rng.standard_normal((7, 10, 2)) * 40).astype(np.float32)
    p = str(tmp_path / "000000_10.png")
    flow_io.write_flow_kitti(p, uv)
    png = flow_io._png_read(p)
    assert png.dtype == np.uint16 and png.shape == (7, 10, 3)
    # file channel order R, G, B = u, v, valid; value = trunc(64 * x + 2^15)   (frame_utils.py:116-120)
    assert np.array_equal(png[..., 0], (64.0 * uv[..., 0].astype(np.float64) + 2 ** 15).astype(np.uint16))
    assert np.array_equal(png[..., 2], np.ones((7, 10), np.uint16))
    flow, valid = flow_io.read_flow_kitti(p)
    assert flow.dtype == np.float32 and np.all(valid == 1.0)
    assert np.abs(flow - uv).max() <= 1.0 / 64 + 1e-6
    # a sparse file: invalid pixels carry valid = 0
    png[2:4, :, 2] = 0
    flow_io._png_write(p, png)
    assert flow_io.read_flow_kitti(p)[1][2:4].sum() == 0
    d = (rng.integers(0, 2, size=(4, 6)) * rng.integers(1, 40000, size=(4, 6))).astype(np.uint16)
    flow_io._png_write(p, d)
    fl, va = flow_io.read_disp_kitti(p)
    assert np.array_equal(va, d > 0) and np.allclose(fl[..., 0], -(d.astype(np.float64) / 256.0)) and np.all(fl[..., 1] == 0)


def _fake_pair(rng, h, w):
    return rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8), (rng.standard_normal((h, w, 2)) * 3).astype(np.float32)


def test_dataset_walkers(tmp_path):
    rng = np.random.default_rng(4)
    # ---- Sintel: <root>/training/{clean,flow}/<scene>/frame_XXXX.{png,flo}; N frames -> N-1 pairs per scene
    root = tmp_path / "Sintel"
    for scene, n in (("alley_1", 3), ("bamboo_2", 2)):
        (root / "training" / "clean" / scene).mkdir(parents=True)
        (root / "training" / "flow" / scene).mkdir(parents=True)
        for i in range(n):
            img, fl = _fake_pair(rng, 8, 12)
            flow_io.write_image(str(root / "training" / "clean" / scene / f"frame_{i + 1:04d}.png"), img)
            if i < n - 1:
                flow_io.write_flo(str(root / "training" / "flow" / scene / f"frame_{i + 1:04d}.flo"), fl)
    ds = MpiSintel(split="training", root=str(root), dstype="clean")
    assert len(ds) == 3 and ds.extra_info == [("alley_1", 0), ("alley_1", 1), ("bamboo_2", 0)]
    im1, im2, fl, va, extra = ds[1]
    assert im1.shape == (3, 8, 12) and im1.dtype == torch.float32 and fl.shape == (2, 8, 12) and va.shape == (8, 12)
    assert ds.image_list[1][0].endswith("frame_0002.png") and ds.image_list[1][1].endswith("frame_0003.png")
    assert torch.all(va == 1) and extra == ("alley_1", 1)
    ds.is_test = True
    assert len(ds[0]) == 3
    # ---- KITTI: image_2/*_10.png + *_11.png, flow_occ/*_10.png
    kroot = tmp_path / "KITTI"
    (kroot / "training" / "image_2").mkdir(parents=True)
    (kroot / "training" / "flow_occ").mkdir(parents=True)
    for i in range(2):
        img, fl = _fake_pair(rng, 6, 10)
        flow_io.write_image(str(kroot / "training" / "image_2" / f"{i:06d}_10.png"), img)
        flow_io.write_image(str(kroot / "training" / "image_2" / f"{i:06d}_11.png"), img[::-1].copy())
        flow_io.write_flow_kitti(str(kroot / "training" / "flow_occ" / f"{i:06d}_10.png"), fl)
    kd = KITTI(split="training", root=str(kroot))
    assert len(kd) == 2 and kd.sparse and kd.extra_info[1] == ["000001_10.png"]
    assert kd[0][2].shape == (2, 6, 10) and kd[0][3].min() == 1
    # ---- FlyingChairs: *.ppm pairs + *.flo + split file (2 = validation)
    croot = tmp_path / "FlyingChairs_release" / "data"
    croot.mkdir(parents=True)
    for i in range(3):
        img, fl = _fake_pair(rng, 4, 6)
        flow_io.write_image(str(croot / f"{i + 1:05d}_img1.ppm"), img)
        flow_io.write_image(str(croot / f"{i + 1:05d}_img2.ppm"), img)
        flow_io.write_flo(str(croot / f"{i + 1:05d}_flow.flo"), fl)
    (tmp_path / "FlyingChairs_release" / "FlyingChairs_train_val.txt").write_text("1\n2\n2\n")
    assert len(FlyingChairs(split="validation", root=str(croot))) == 2
    assert len(FlyingChairs(split="training", root=str(croot))) == 1


def test_read_image_palette_and_grey_alpha(tmp_path):
    """Palette PNGs and grey+alpha PNGs: read_image must hand back uint8 [H, W, 3] like the reference's reader."""
    from PIL import Image
    from craft_amd import flow_io
    r = np.random.RandomState(2)
    rgb = r.randint(0, 256, size=(9, 13, 3)).astype(np.uint8)
    Image.fromarray(rgb).convert("P", palette=Image.ADAPTIVE, colors=16).save(str(tmp_path / "pal.png"))
    ref = np.array(Image.open(str(tmp_path / "pal.png")).convert("RGB"))
    got = flow_io.read_image(str(tmp_path / "pal.png"))
    assert got.shape == (9, 13, 3) and got.dtype == np.uint8 and np.array_equal(got, ref)
    la = np.stack([r.randint(0, 256, size=(9, 13)), r.randint(0, 256, size=(9, 13))], -1).astype(np.uint8)
    flow_io._png_write(str(tmp_path / "la.png"), la[..., 0])
    Image.fromarray(la, mode="LA").save(str(tmp_path / "la.png"))
    got = flow_io.read_image(str(tmp_path / "la.png"))
    assert got.shape == (9, 13, 3) and np.array_equal(got[..., 0], la[..., 0]) and np.array_equal(got[..., 2], la[..., 0])


FLOWIO = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flowio_ref.npz"))


def test_flo_and_pfm_against_the_references_own_functions(tmp_path):
    """tests/golden/flowio_ref.npz (tools/make_golden_flowio.py): readFlow / writeFlow / readPFM compiled out of the reference's
    core/utils/frame_utils.py and run on seeded arrays.  write_flo produces the reference's BYTES (both call forms, float64 input included:
    one cast to float32), read_flo returns what readFlow returns for them, read_pfm what readPFM returns for colour / grey, little- / big-endian
    files with non-unit scales (the reference, like this reader, ignores the scale's magnitude)."""
    p = str(tmp_path / "x.flo")
    for k in range(3):
        uv, raw = FLOWIO[f"flo.{k}.uv"], FLOWIO[f"flo.{k}.bytes"].tobytes()
        flow_io.write_flo(p, uv)
        assert open(p, "rb").read() == raw
        flow_io.write_flo(p, uv[..., 0], uv[..., 1])
        assert open(p, "rb").read() == raw
        open(p, "wb").write(raw)
        got = flow_io.read_flo(p)
        assert got.dtype == np.float32 and np.array_equal(got, FLOWIO[f"flo.{k}.read"]) and np.array_equal(got, uv)
    flow_io.write_flo(p, FLOWIO["flo.f64.uv"])
    assert open(p, "rb").read() == FLOWIO["flo.f64.bytes"].tobytes()
    q = str(tmp_path / "x.pfm")
    for k in range(int(FLOWIO["pfm.n"])):
        open(q, "wb").write(FLOWIO[f"pfm.{k}.bytes"].tobytes())
        got = flow_io.read_pfm(q)
        assert got.dtype == np.float32 and np.array_equal(got, FLOWIO[f"pfm.{k}.read"]), k
