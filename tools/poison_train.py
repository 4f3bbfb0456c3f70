#!/usr/bin/env python3
"""Uninitialised-read detector for the training step: fill the caching allocator's free pool with NaN (or a finite garbage value) before
a step, so any kernel that reads memory it never wrote shows up as a NaN / changed loss.  usage: python tools/poison_train.py [--policy mixed]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from craft_amd import CRAFT, default_args
from craft_amd.synth import synth_pair, synth_state_dict
from craft_amd.train import Trainer

ap = argparse.ArgumentParser()
ap.add_argument("--policy", default="mixed"); ap.add_argument("--B", type=int, default=2); ap.add_argument("--iters", type=int, default=2)
ap.add_argument("--H", type=int, default=368); ap.add_argument("--W", type=int, default=496); ap.add_argument("--gb", type=float, default=30.0)
a = ap.parse_args()
dev = torch.device("cuda")


def poison(value):
    n = int(a.gb * 2 ** 30 // 4)
    chunks = [torch.full((n // 8,), value, device=dev) for _ in range(8)]      # several block sizes end up in the pool
    small = [torch.full((m,), value, device=dev) for m in (1 << 8, 1 << 12, 1 << 16, 1 << 20, 1 << 22) for _ in range(64)]
    del chunks, small


def run(tag, value):
    torch.cuda.empty_cache()
    if value is not None:
        poison(value)
    model = CRAFT(default_args(hip_precision=a.policy))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
    model = model.to(dev)
    tr = Trainer(model, lr=4e-4, num_steps=1000, iters=a.iters, clip=1.0)
    im1, im2, flow = synth_pair(a.B, a.H, a.W, seed=100)
    valid = torch.ones(a.B, a.H, a.W)
    out = []
    for i in range(3):
        if value is not None:
            poison(value)
        m = tr.step(im1, im2, flow, valid)
        out.append(m["loss"])
    print(f"{tag:28s} losses {out}", flush=True)
    del tr, model


run("fresh", None)
run("fresh again", None)
run("pool poisoned with 3.0", 3.0)
run("pool poisoned with NaN", float("nan"))
