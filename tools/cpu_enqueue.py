#!/usr/bin/env python3
"""Host time to ENQUEUE one forward (no sync) vs its GPU time: tells whether a batch / image size is launch-bound."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from craft_amd import CRAFT, default_args
from craft_amd.synth import synth_pair, synth_state_dict
B, H, W, T = (int(x) for x in sys.argv[1:5])
dev = torch.device("cuda")
m = CRAFT(default_args(hip_precision="mixed")); m.load_state_dict(synth_state_dict(m.state_dict(), seed=1234)); m = m.to(dev).eval()
im1, im2, _ = synth_pair(B, H, W, seed=1); im1, im2 = im1.to(dev), im2.to(dev)
with torch.no_grad():
    for _ in range(3): m(im1, im2, iters=T, test_mode=1)
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(10):
        t0 = time.perf_counter(); m(im1, im2, iters=T, test_mode=1); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print(f"B{B} {H}x{W} T{T}: enqueue {sum(enq)/10:.2f} ms, enqueue+drain {sum(tot)/10:.2f} ms per forward (one at a time)")
