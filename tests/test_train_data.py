"""Training input feed (craft_amd/train_data.py = datasets.py FlowDataset.__getitem__ + fetch_dataloader): miniature Sintel and KITTI
trees on disk -> GPU augmentation -> batches -> one Trainer.step."""
import numpy as np
import pytest
import torch

from craft_amd import CRAFT, default_args, flow_io
from craft_amd.flow_datasets import KITTI, MpiSintel
from craft_amd.synth import synth_pair, synth_state_dict
from craft_amd.train import Trainer
from craft_amd.train_data import STAGE_AUG, TrainSource, make_augmentor, seed_workers, train_batches, train_batches_async

pytestmark = pytest.mark.gpu


def _trees(tmp_path, n=4, H=160, W=240):
    im1, im2, flow = synth_pair(n + 1, H, W, seed=21, max_flow=5)
    gt = flow.permute(0, 2, 3, 1).numpy()
    u8 = lambda t: t.permute(1, 2, 0).numpy().astype(np.uint8)
    root = tmp_path / "Sintel"
    for i in range(n):
        for sub in ("clean", "final", "flow"):
            (root / "training" / sub / f"s{i}").mkdir(parents=True)
        for dst in ("clean", "final"):
            flow_io.write_image(str(root / "training" / dst / f"s{i}" / "frame_0001.png"), u8(im1[i]))
            flow_io.write_image(str(root / "training" / dst / f"s{i}" / "frame_0002.png"), u8(im2[i]))
        flow_io.write_flo(str(root / "training" / "flow" / f"s{i}" / "frame_0001.flo"), gt[i])
    kroot = tmp_path / "KITTI"
    (kroot / "training" / "image_2").mkdir(parents=True)
    (kroot / "training" / "flow_occ").mkdir(parents=True)
    flow_io.write_image(str(kroot / "training" / "image_2" / "000000_10.png"), u8(im1[n]))
    flow_io.write_image(str(kroot / "training" / "image_2" / "000000_11.png"), u8(im2[n]))
    flow_io.write_flow_kitti(str(kroot / "training" / "flow_occ" / "000000_10.png"), gt[n])
    return str(root), str(kroot)


def test_stage_parameters_match_fetch_dataloader():
    assert STAGE_AUG["chairs"] == dict(min_scale=-0.1, max_scale=1.0, do_flip=True)                      # datasets.py:513-515
    assert STAGE_AUG["sintel"] == dict(min_scale=-0.2, max_scale=0.6, do_flip=True)                      # :537
    assert STAGE_AUG["kitti"] == dict(min_scale=-0.2, max_scale=0.4, do_flip=False)                      # :554
    assert STAGE_AUG["viper"]["spatial_aug_prob"] == 1 and STAGE_AUG["autoflow"]["spatial_aug_prob"] == 1


def test_batches_from_files_and_one_training_step(device, tmp_path):
    root, kroot = _trees(tmp_path)
    crop = (96, 128)
    clean, final, kitti = MpiSintel("training", root, "clean"), MpiSintel("training", root, "final"), KITTI("training", kroot)
    # the sintel stage's mix (datasets.py:548: replication factors in front of each dataset), miniature factors here
    sources = [TrainSource(clean, make_augmentor(clean, "sintel", crop), repeat=2),
               TrainSource(final, make_augmentor(final, "sintel", crop, shift_prob=0.5), repeat=2),
               TrainSource(kitti, make_augmentor(kitti, "sintel/kitti", crop), repeat=3)]
    assert sum(len(s) for s in sources) == 2 * 4 + 2 * 4 + 3
    seed_workers(3)
    it = train_batches(sources, batch_size=2, device=device, seed=1, epochs=1)
    batches = list(it)
    assert len(batches) == 19 // 2                                   # drop_last
    for im1, im2, flow, valid in batches:
        assert im1.shape == (2, 3, *crop) and im2.shape == im1.shape and flow.shape == (2, 2, *crop) and valid.shape == (2, *crop)
        assert im1.is_cuda and torch.isfinite(im1).all() and torch.isfinite(flow).all()
        assert float(im1.min()) >= 0 and float(im1.max()) <= 255
        assert set(torch.unique(valid).tolist()) <= {0.0, 1.0} and float(valid.mean()) > 0.2
    # two ranks see disjoint halves of the epoch's permutation
    seed_workers(3)
    r0 = list(train_batches(sources, 2, device, seed=1, rank=0, world=2, epochs=1))
    r1 = list(train_batches(sources, 2, device, seed=1, rank=1, world=2, epochs=1))
    assert len(r0) == len(r1) == 5                                   # 19 samples padded to 20: 10 per rank, batches of 2
    # and the batches train
    model = CRAFT(default_args(hip_precision="fp32"))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=2), strict=True)
    tr = Trainer(model.to(device), lr=1e-4, num_steps=20, iters=2, freeze_bn=True)
    m = tr.step(*batches[0])
    assert m["loss"] == m["loss"] and m["loss"] < 1e4


def test_async_feed_yields_the_same_batches(device, tmp_path):
    """train_batches_async (decode workers + pinned staging + augmentation on a side stream, datasets.py:569-580 / train.py:337's
    DataLoader(num_workers=4)) reproduces train_batches batch for batch under the same seeds, over two epochs, and stops cleanly when
    the consumer walks away early."""
    root, kroot = _trees(tmp_path)
    crop = (96, 128)
    clean, kitti = MpiSintel("training", root, "clean"), KITTI("training", kroot)

    def sources():
        return [TrainSource(clean, make_augmentor(clean, "sintel", crop, shift_prob=0.5), repeat=2),
                TrainSource(kitti, make_augmentor(kitti, "sintel/kitti", crop), repeat=3)]
    seed_workers(5)
    ref = [tuple(t.clone() for t in b) for b in train_batches(sources(), 2, device, seed=4, epochs=2)]
    seed_workers(5)
    got = []
    for b in train_batches_async(sources(), 2, device, seed=4, epochs=2, workers=3, prefetch=2):
        got.append(tuple(t.clone() for t in b))
    assert len(got) == len(ref) == 2 * (11 // 2)
    for a, b in zip(got, ref):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    it = train_batches_async(sources(), 2, device, seed=4, epochs=None, workers=2, prefetch=2)      # endless: take 3 batches, then close
    for _ in range(3):
        next(it)
    it.close()


def test_training_driver_runs_checkpoints_validates_and_resumes(device, tmp_path):
    """craft_amd.train_main (train.py:176-262): the sintel stage on miniature trees -- steps, periodic checkpoint + validation through the
    evaluation harness, the final checkpoint in the reference's layout, and a resumed run that continues the schedule."""
    from craft_amd import train_main
    from craft_amd.utils import read_checkpoint
    root, kroot = _trees(tmp_path)
    out = tmp_path / "ckpt"
    common = ["--stage", "sintel", "--validation", "sintel", "kitti", "--output", str(out), "--batch_size", "2", "--image_size", "96", "128",
              "--iters", "2", "--print_freq", "2", "--lr", "1e-4", "--craft", "--f2", "full", "--setrans", "--sintel_root", root, "--kitti_root", kroot,
              "--workers", "2"]
    path = train_main.main(["--name", "mini", "--num_steps", "4", "--val_freq", "2"] + common)
    assert path == str(out / "mini.pth") and (out / "2_mini.pth").exists() and (out / "4_mini.pth").exists()
    ck = read_checkpoint(path)
    assert set(ck) == {"model", "optimizer", "lr_scheduler", "logger"} and all(k.startswith("module.") for k in ck["model"])
    assert ck["logger"]["total_steps"] == 4 and ck["logger"]["val_steps"] == [2, 4]
    assert set(ck["logger"]["val_results"]) >= {"clean", "final", "epe", "f1"} and all(len(v) == 2 for v in ck["logger"]["val_results"].values())
    # resume with optimizer + scheduler state under the reference's own precision recipe: two more steps of a 6-step schedule
    path2 = train_main.main(["--name", "mini2", "--num_steps", "6", "--val_freq", "100", "--restore_ckpt", str(out / "2_mini.pth"), "--loadopt", "--loadsched",
                             "--mixed_precision"] + common)
    ck2 = read_checkpoint(path2)
    assert ck2["logger"]["total_steps"] == 6 and ck2["lr_scheduler"]["last_epoch"] == 6


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_training_driver_two_ranks(device, tmp_path, backend):
    """The driver under torch.distributed.run: two ranks (gloo: sharing the box's GPU; nccl: one GPU each, skipped on a 1-GPU box), each on
    its strided share of the epoch, one gradient all-reduce per step, rank 0 checkpoints and validates while rank 1 waits at the barrier."""
    import os
    import socket
    import subprocess
    import sys
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    root, kroot = _trees(tmp_path)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CRAFT_BENCH_BACKEND"] = backend
    out = tmp_path / "ckpt2"
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           "-m", "craft_amd.train_main", "--name", "dp", "--stage", "sintel", "--validation", "sintel", "--output", str(out), "--batch_size", "1",
           "--image_size", "96", "128", "--iters", "2", "--num_steps", "3", "--val_freq", "2", "--print_freq", "1", "--craft", "--setrans",
           "--sintel_root", root, "--kitti_root", kroot, "--workers", "2"]
    r = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    assert (out / "dp.pth").exists() and (out / "2_dp.pth").exists()
    assert "2 rank(s) x batch 1" in r.stdout and r.stdout.count("[train_main] step") == 2
