"""Harness row H' pinned to the reference's own code (tests/golden/harness.npz, written by tools/make_golden_harness.py,
which compiles validate_chairs / validate_sintel / validate_kitti / shift_pixels from /root/reference/evaluate.py's AST and
sequence_loss from train.py's, and runs them here on synthetic data with a deterministic stand-in for the network).

CPU: ``InputPadder`` against the pads and padded tensors of the reference's class.
CPU: ``shift_pixels`` against the reference function's outputs (the shift-robustness experiment, shifteval.sh).
GPU: our validate_* drivers (file formats -> pad -> model -> unpad -> craft_flow_metrics) on the same data written to disk in
the datasets' own formats must return what the reference's functions returned; our ``sequence_loss`` kernel must return the
reference's loss, metrics and gradients."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from craft_amd.utils import InputPadder

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "harness.npz"))


def test_input_padder_matches_reference_table():
    for H, W, kitti, mod, *pad in Z["padder.table"].tolist():
        ours = InputPadder((1, 3, H, W), mode="kitti" if kitti else "sintel", mod=mod)
        assert list(ours._pad) == pad, (H, W, kitti, mod)
    x = torch.from_numpy(Z["padder.x"])
    for mode in ("sintel", "kitti"):
        p = InputPadder(x.shape, mode=mode)
        (xp,) = p.pad(x)
        assert np.array_equal(xp.numpy(), Z[f"padder.{mode}.padded"])
        assert torch.equal(p.unpad(xp), x)


def test_shift_pixels_matches_the_reference_function():
    """craft_amd.evaluate.shift_pixels against the outputs of the reference's own function (evaluate.py:44-89): the four quadrants, a
    shift along one axis only (moves nothing there either), (0, 0), and a 3-D (unbatched) input."""
    from craft_amd.evaluate import shift_pixels
    img, flow = torch.from_numpy(Z["shiftpx.img"]), torch.from_numpy(Z["shiftpx.flow"])
    for i, xy in enumerate(Z["shiftpx.cases"].tolist()):
        a, b, m = shift_pixels(img.clone(), flow.clone(), tuple(xy))
        assert np.array_equal(a.numpy(), Z[f"shiftpx.{i}.img"]), xy
        assert np.array_equal(b.numpy(), Z[f"shiftpx.{i}.flow"]), xy
        assert np.array_equal(m.numpy(), Z[f"shiftpx.{i}.mask"]), xy
    a, b, _ = shift_pixels(img[0].clone(), flow[0].clone(), (3, 2))
    assert np.array_equal(a.numpy(), Z["shiftpx.3d.img"]) and np.array_equal(b.numpy(), Z["shiftpx.3d.flow"])
    a, b, m = shift_pixels(img, None, None)
    assert a is img and b is None and bool(m.all())


class StandInNet(torch.nn.Module):
    """The same deterministic stand-in the fixture was generated with (test scaffolding, not the product)."""

    def forward(self, image1, image2, iters=6, flow_init=None, test_mode=1, **kw):
        d = image1[:, :2].float() - image2[:, 1:3].float()
        up = F.avg_pool2d(d, 5, stride=1, padding=2, count_include_pad=False) / 2.0
        return F.avg_pool2d(up, 8) / 8.0, up


def _hwc(a):
    return np.ascontiguousarray(a.transpose(1, 2, 0))


@pytest.mark.gpu
def test_validate_drivers_return_the_reference_numbers(device, tmp_path):
    from craft_amd import evaluate, flow_io
    net = StandInNet()
    # ---- MPI-Sintel layout: training/{clean,final,flow}/<scene>/frame_XXXX.{png,flo}; one scene per pair
    root = tmp_path / "Sintel"
    n = len(Z["sintel_clean.im1"])
    assert np.array_equal(Z["sintel_clean.gt"], Z["sintel_clean.gt"])
    for dst in ("clean", "final"):
        for i in range(n):
            d = root / "training" / dst / f"s{i}"
            d.mkdir(parents=True)
            flow_io.write_image(str(d / "frame_0001.png"), _hwc(Z[f"sintel_{dst}.im1"][i]))
            flow_io.write_image(str(d / "frame_0002.png"), _hwc(Z[f"sintel_{dst}.im2"][i]))
    # the flow directory is shared by both passes in the dataset; the fixture's two passes have their own ground truth, so
    # they are evaluated one tree at a time
    for dst in ("clean", "final"):
        for i in range(n):
            d = root / "training" / "flow" / f"s{i}"
            d.mkdir(parents=True, exist_ok=True)
            flow_io.write_flo(str(d / "frame_0001.flo"), _hwc(Z[f"sintel_{dst}.gt"][i]))
        res = evaluate.validate_sintel(net, root=str(root), iters=4, dstype=dst, batch_size=2, device=device)
        assert res[dst] == pytest.approx(float(Z[f"sintel.{dst}"]), rel=2e-6)
        m = res[dst + "_metrics"]
        assert [m["px1"], m["px3"], m["px5"]] == pytest.approx(Z[f"sintel.{dst}.px"].tolist(), abs=1e-6)      # printed with %f
        lo, mags = 0, []
        for hi in evaluate.MAG_ENDPOINTS:
            mags.append(m[f"epe_{lo}-{hi}"])
            lo = hi
        assert mags == pytest.approx(Z[f"sintel.{dst}.mag"].tolist(), abs=5.1e-3)                               # printed with .2f
        if dst == "clean":        # the shift experiment of shifteval.sh (xy_shift, evaluate.py:510-534) on the same tree
            for i, xy in enumerate(Z["shiftval.sintel"].tolist()):
                sres = evaluate.validate_sintel(net, root=str(root), iters=4, dstype="clean", batch_size=2, device=device, xy_shift=tuple(xy))
                assert sres["clean"] == pytest.approx(float(Z[f"shiftval.sintel.{i}.epe"]), rel=2e-6), xy
                sm = sres["clean_metrics"]
                assert [sm["px1"], sm["px3"], sm["px5"]] == pytest.approx(Z[f"shiftval.sintel.{i}.px"].tolist(), abs=1e-6)
                lo, smag = 0, []
                for hi in evaluate.MAG_ENDPOINTS:
                    smag.append(sm[f"epe_{lo}-{hi}"])
                    lo = hi
                assert smag == pytest.approx(Z[f"shiftval.sintel.{i}.mag"].tolist(), abs=5.1e-3)
    # ---- KITTI layout: sparse ground truth as 16-bit PNG (the fixture's flow sits on the 1/64 px grid: lossless)
    kroot = tmp_path / "KITTI"
    (kroot / "training" / "image_2").mkdir(parents=True)
    (kroot / "training" / "flow_occ").mkdir(parents=True)
    for i in range(len(Z["kitti.im1"])):
        flow_io.write_image(str(kroot / "training" / "image_2" / f"{i:06d}_10.png"), _hwc(Z["kitti.im1"][i]))
        flow_io.write_image(str(kroot / "training" / "image_2" / f"{i:06d}_11.png"), _hwc(Z["kitti.im2"][i]))
        fp = str(kroot / "training" / "flow_occ" / f"{i:06d}_10.png")
        flow_io.write_flow_kitti(fp, _hwc(Z["kitti.gt"][i]))
        png = flow_io._png_read(fp)
        png[..., 2] = (Z["kitti.valid"][i] >= 0.5).astype(png.dtype)
        flow_io._png_write(fp, png)
        back, valid = flow_io.read_flow_kitti(fp)
        assert np.array_equal(back, _hwc(Z["kitti.gt"][i])) and np.array_equal(valid, Z["kitti.valid"][i])
    kres = evaluate.validate_kitti(net, root=str(kroot), iters=4, device=device)
    assert kres["epe"] == pytest.approx(float(Z["kitti.epe"]), rel=2e-6)
    assert kres["f1"] == pytest.approx(float(Z["kitti.f1"]), rel=2e-6)
    km = kres["metrics"]
    assert [km["px1"], km["px3"], km["px5"]] == pytest.approx(Z["kitti.px"].tolist(), abs=5.1e-5)               # printed with .4f
    for i, xy in enumerate(Z["shiftval.kitti"].tolist()):
        kres = evaluate.validate_kitti(net, root=str(kroot), iters=4, device=device, xy_shift=tuple(xy))
        assert kres["epe"] == pytest.approx(float(Z[f"shiftval.kitti.{i}.epe"]), rel=2e-6)
        assert kres["f1"] == pytest.approx(float(Z[f"shiftval.kitti.{i}.f1"]), rel=2e-6)
    # ---- FlyingChairs layout: <id>_img1.ppm / _img2.ppm / _flow.flo + a split file (2 = validation)
    croot = tmp_path / "chairs"
    croot.mkdir()
    nc = len(Z["chairs.im1"])
    for i in range(nc):
        flow_io.write_image(str(croot / f"{i + 1:05d}_img1.ppm"), _hwc(Z["chairs.im1"][i]))
        flow_io.write_image(str(croot / f"{i + 1:05d}_img2.ppm"), _hwc(Z["chairs.im2"][i]))
        flow_io.write_flo(str(croot / f"{i + 1:05d}_flow.flo"), _hwc(Z["chairs.gt"][i]))
    split = tmp_path / "chairs_split.txt"
    split.write_text("2\n" * nc)
    cres = evaluate.validate_chairs(net, root=str(croot), iters=4, batch_size=2, split_file=str(split), device=device)
    assert cres["chairs_epe"] == pytest.approx(float(Z["chairs.epe"]), rel=2e-6)
    for i, xy in enumerate(Z["shiftval.chairs"].tolist()):
        cres = evaluate.validate_chairs(net, root=str(croot), iters=4, batch_size=2, split_file=str(split), device=device, xy_shift=tuple(xy))
        assert cres["chairs_epe"] == pytest.approx(float(Z[f"shiftval.chairs.{i}.epe"]), rel=2e-6)


@pytest.mark.gpu
def test_sequence_loss_matches_reference_function(device):
    from craft_amd.train import sequence_loss
    preds = [torch.from_numpy(p).to(device) for p in Z["loss.preds"]]
    gt, valid = torch.from_numpy(Z["loss.gt"]), torch.from_numpy(Z["loss.valid"])
    loss, metrics, grads = sequence_loss(preds, gt, valid, gamma=0.8, max_flow=float(Z["loss.max_flow"]), want_grad=True)
    assert float(loss) == pytest.approx(float(Z["loss.value"]), rel=2e-6)
    assert [metrics["epe"], metrics["1px"], metrics["3px"], metrics["5px"]] == pytest.approx(Z["loss.metrics"].tolist(), rel=2e-6)
    ref = Z["loss.grads"]
    for i, g in enumerate(grads):
        assert np.allclose(g.cpu().numpy(), ref[i], rtol=1e-6, atol=1e-12), f"d loss / d pred_{i}"
