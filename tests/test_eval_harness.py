"""Evaluation harness on the GPU: craft_flow_metrics vs the reference's numpy formulas and the
validate_* drivers on miniature Sintel / KITTI / Chairs trees written with our own format writers."""
import numpy as np
import pytest
import torch

from craft_amd import CRAFT, default_args, evaluate, flow_io
from craft_amd.synth import synth_pair, synth_state_dict

pytestmark = pytest.mark.gpu


def _ref_metrics(pr, gt, valid, off=(0.0, 0.0)):
    """evaluate.py:529 (EPE), :578-598 (px rates, magnitude ranges), :833-841 (KITTI outliers), in numpy."""
    epe = np.sqrt(((pr - gt) ** 2).sum(1)).reshape(-1)
    mag = np.sqrt(((gt + np.array(off).reshape(1, 2, 1, 1)) ** 2).sum(1)).reshape(-1)
    v = np.ones_like(epe, bool) if valid is None else valid.reshape(-1) >= 0.5
    e, m = epe[v], mag[v]
    out = {"epe": e.mean(), "px1": (e < 1).mean(), "px3": (e < 3).mean(), "px5": (e < 5).mean(),
           "f1": 100 * ((e > 3) & (e / m > 0.05)).mean(), "count": v.sum()}
    lo = 0
    for hi in evaluate.MAG_ENDPOINTS:
        sel = (m >= lo) & (m < hi)
        out[f"epe_{lo}-{hi}"] = e[sel].mean() if sel.sum() else 0.0
        lo = hi
    return out


@pytest.mark.parametrize("sparse", [False, True])
def test_flow_metrics_kernel(device, sparse):
    rng = np.random.default_rng(7)
    B, H, W = 3, 37, 53
    gt = (rng.standard_normal((B, 2, H, W)) * rng.choice([0.3, 4, 15, 40], size=(B, 1, H, W))).astype(np.float32)
    pr = gt + (rng.standard_normal((B, 2, H, W)) * rng.choice([0.2, 2, 6], size=(B, 1, H, W))).astype(np.float32)
    valid = (rng.random((B, H, W)) > 0.4).astype(np.float32) if sparse else None
    m = evaluate.FlowMetrics(device)
    # two updates accumulate like two batches
    m.update(torch.from_numpy(pr[:2]).to(device), torch.from_numpy(gt[:2]), None if valid is None else torch.from_numpy(valid[:2]), (1.5, -2.0))
    m.update(torch.from_numpy(pr[2:]).to(device), torch.from_numpy(gt[2:]), None if valid is None else torch.from_numpy(valid[2:]), (1.5, -2.0))
    got, ref = m.result(), _ref_metrics(pr, gt, valid, (1.5, -2.0))
    for k, v in ref.items():
        assert got[k] == pytest.approx(v, rel=2e-5, abs=1e-6), k


def _tiny_model(device):
    model = CRAFT(default_args(hip_precision="fp32"))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
    return model.to(device).eval()


def test_validate_drivers_on_miniature_datasets(device, tmp_path):
    model = _tiny_model(device)
    H, W = 132, 250                      # not multiples of 8: exercises both padding modes
    im1, im2, flow = synth_pair(3, H, W, seed=11, max_flow=6)
    gt = flow.permute(0, 2, 3, 1).numpy()
    u8 = lambda t: t.permute(1, 2, 0).numpy().astype(np.uint8)
    # ---- Sintel tree (two scenes; frames are independent pairs here, so each pair gets its own scene directory)
    root = tmp_path / "Sintel"
    for i in range(2):
        for sub in ("clean", "final", "flow"):
            (root / "training" / sub / f"s{i}").mkdir(parents=True)
        for dst in ("clean", "final"):
            flow_io.write_image(str(root / "training" / dst / f"s{i}" / "frame_0001.png"), u8(im1[i]))
            flow_io.write_image(str(root / "training" / dst / f"s{i}" / "frame_0002.png"), u8(im2[i]))
        flow_io.write_flo(str(root / "training" / "flow" / f"s{i}" / "frame_0001.flo"), gt[i])
    res = evaluate.validate_sintel(model, root=str(root), iters=3, dstype="both", batch_size=2, device=device)
    # the same numbers computed by hand: pad (sintel mode) -> forward -> unpad -> EPE over all pixels
    with torch.no_grad():
        _, up = evaluate._predict(model, im1[:2], im2[:2], 3, "sintel", device)
    ref = _ref_metrics(up.cpu().numpy(), flow[:2].numpy(), None)
    assert up.shape[-2:] == (H, W)
    assert res["clean"] == pytest.approx(ref["epe"], rel=1e-5) and res["final"] == pytest.approx(ref["epe"], rel=1e-5)
    assert res["clean_metrics"]["px3"] == pytest.approx(ref["px3"], abs=1e-6)
    # ---- KITTI tree: sparse ground truth in the 16-bit PNG encoding (quantised to 1/64 px), bottom padding
    kroot = tmp_path / "KITTI"
    (kroot / "training" / "image_2").mkdir(parents=True)
    (kroot / "training" / "flow_occ").mkdir(parents=True)
    flow_io.write_image(str(kroot / "training" / "image_2" / "000000_10.png"), u8(im1[2]))
    flow_io.write_image(str(kroot / "training" / "image_2" / "000000_11.png"), u8(im2[2]))
    flow_io.write_flow_kitti(str(kroot / "training" / "flow_occ" / "000000_10.png"), gt[2])
    png = flow_io._png_read(str(kroot / "training" / "flow_occ" / "000000_10.png"))
    png[::3, :, 2] = 0                                              # every third row invalid
    flow_io._png_write(str(kroot / "training" / "flow_occ" / "000000_10.png"), png)
    kres = evaluate.validate_kitti(model, root=str(kroot), iters=3, device=device)
    gtq, valid = flow_io.read_flow_kitti(str(kroot / "training" / "flow_occ" / "000000_10.png"))
    with torch.no_grad():
        _, upk = evaluate._predict(model, im1[2:], im2[2:], 3, "kitti", device)
    kref = _ref_metrics(upk.cpu().numpy(), gtq.transpose(2, 0, 1)[None], valid[None])
    assert kres["epe"] == pytest.approx(kref["epe"], rel=1e-5) and kres["f1"] == pytest.approx(kref["f1"], abs=1e-4)
    assert kres["metrics"]["count"] == valid.sum()
    # ---- submissions: .flo files that read back as the prediction
    (root / "test").mkdir()
    for dst in ("clean", "final"):
        (root / "test" / dst / "s0").mkdir(parents=True)
        flow_io.write_image(str(root / "test" / dst / "s0" / "frame_0001.png"), u8(im1[0]))
        flow_io.write_image(str(root / "test" / dst / "s0" / "frame_0002.png"), u8(im2[0]))
    out = tmp_path / "sub"
    evaluate.create_sintel_submission(model, root=str(root), output_path=str(out), iters=3, device=device)
    sub = flow_io.read_flo(str(out / "clean" / "s0" / "frame0001.flo"))
    with torch.no_grad():
        _, up0 = evaluate._predict(model, im1[:1], im2[:1], 3, "sintel", device)
    assert np.abs(sub - up0[0].permute(1, 2, 0).cpu().numpy()).max() < 1e-4
    # ---- warm start: the second pair of a scene starts from the forward-interpolated low-res flow of the first
    from craft_amd.utils import forward_interpolate
    flow_io.write_image(str(root / "test" / "clean" / "s0" / "frame_0003.png"), u8(im1[0]))
    flow_io.write_image(str(root / "test" / "final" / "s0" / "frame_0003.png"), u8(im1[0]))
    out2 = tmp_path / "sub_ws"
    evaluate.create_sintel_submission(model, root=str(root), output_path=str(out2), iters=3, device=device, warm_start=True)
    with torch.no_grad():
        lo0, _ = evaluate._predict(model, im1[:1], im2[:1], 3, "sintel", device)
        _, up1 = evaluate._predict(model, im2[:1], im1[:1], 3, "sintel", device, flow_init=forward_interpolate(lo0[0])[None])
    sub1 = flow_io.read_flo(str(out2 / "clean" / "s0" / "frame0002.flo"))
    assert np.abs(sub1 - up1[0].permute(1, 2, 0).cpu().numpy()).max() < 1e-4
    assert np.abs(flow_io.read_flo(str(out2 / "clean" / "s0" / "frame0001.flo")) - sub).max() < 1e-4   # first frame: no init
