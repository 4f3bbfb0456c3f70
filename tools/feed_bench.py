#!/usr/bin/env python3
"""Is training from files input-bound?  Steps per second of Trainer.step at BASELINE configs[4] (Sintel crops 368x768, batch 4, frozen
BatchNorm, bf16 MFMA attention) fed (a) from tensors resident in HBM, (b) by train_batches (decode + upload + augmentation inline, between
steps), (c) by train_batches_async (decode workers, pinned staging, augmentation on a side stream, 2 batches ahead) from a synthetic MPI-Sintel
tree on local disk (1024x436 PNG frames + .flo, the real dataset's sizes).      usage: python tools/feed_bench.py [--steps 16] [--workers 4]"""
import argparse, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from craft_amd import CRAFT, default_args, flow_io
from craft_amd.flow_datasets import MpiSintel
from craft_amd.synth import synth_pair, synth_state_dict
from craft_amd.train import Trainer
from craft_amd.train_data import TrainSource, make_augmentor, seed_workers, train_batches, train_batches_async

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=16); ap.add_argument("--workers", type=int, default=12); ap.add_argument("--frames", type=int, default=24)
a = ap.parse_args()
dev = torch.device("cuda")
tmp = tempfile.mkdtemp(prefix="craft_sintel_")
im1, im2, flow = synth_pair(a.frames, 436, 1024, seed=5, max_flow=20)
rng = np.random.RandomState(0)
for i in range(a.frames):
    for sub in ("clean", "flow"):
        os.makedirs(os.path.join(tmp, "training", sub, f"s{i:02d}"), exist_ok=True)
    for j, im in enumerate((im1[i], im2[i])):
        u8 = im.permute(1, 2, 0).numpy().astype(np.int16) + rng.randint(-6, 7, size=(436, 1024, 3))        # sensor-like noise: realistic PNG entropy
        flow_io.write_image(os.path.join(tmp, "training", "clean", f"s{i:02d}", f"frame_{j + 1:04d}.png"), np.clip(u8, 0, 255).astype(np.uint8))
    flow_io.write_flo(os.path.join(tmp, "training", "flow", f"s{i:02d}", "frame_0001.flo"), flow[i].permute(1, 2, 0).numpy())
ds = MpiSintel("training", tmp, "clean")
png_kb = os.path.getsize(os.path.join(tmp, "training", "clean", "s00", "frame_0001.png")) / 1024
crop = (368, 768)
model = CRAFT(default_args(hip_precision="train_bf16attn"))
model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
tr = Trainer(model.to(dev), lr=1.25e-4, wdecay=1e-5, num_steps=100000, iters=12, clip=1.0, freeze_bn=True)


def run(feed, n):
    it = iter(feed)
    for _ in range(4):
        tr.step(*next(it))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(*next(it))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    if hasattr(it, "close"):
        it.close()
    return dt


def resident():
    b1, b2, f = synth_pair(4, *crop, seed=9)
    batch = (b1.to(dev), b2.to(dev), f.to(dev), torch.ones(4, *crop, device=dev))
    while True:
        yield batch


src = lambda: [TrainSource(ds, make_augmentor(ds, "sintel", crop))]      # noqa: E731
t0 = time.perf_counter(); [ds[i] for i in range(4)]; dec_ms = 1e3 * (time.perf_counter() - t0) / 4
out = {"workload": "configs[4]: Sintel crops 368x768, batch 4, 12 iters, train_bf16attn, frozen BatchNorm", "frames": a.frames,
       "png_kb": round(png_kb, 1), "decode_ms_per_sample_one_thread": round(dec_ms, 1)}
out["resident_ms_per_step"] = round(1e3 * run(resident(), a.steps), 2)
seed_workers(1)
out["sync_feed_ms_per_step"] = round(1e3 * run(train_batches(src(), 4, dev, seed=1), a.steps), 2)
seed_workers(1)
out["async_feed_ms_per_step"] = round(1e3 * run(train_batches_async(src(), 4, dev, seed=1, workers=a.workers, prefetch=2), a.steps), 2)
out["workers"] = a.workers
out["async_over_resident"] = round(out["async_feed_ms_per_step"] / out["resident_ms_per_step"], 3)
print(json.dumps(out))
