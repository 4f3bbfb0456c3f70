#!/usr/bin/env python3
"""Experiment: capture CRAFT.forward in a HIP graph (torch.cuda.CUDAGraph) and compare replay with eager launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from craft_amd import CRAFT, default_args
from craft_amd.synth import synth_pair, synth_state_dict

def main():
    B, H, W, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dev = torch.device("cuda")
    m = CRAFT(default_args(hip_precision="mixed"))
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=1234)); m = m.to(dev).eval()
    im1, im2, _ = synth_pair(B, H, W, seed=1); im1, im2 = im1.to(dev), im2.to(dev)
    with torch.no_grad():
        for _ in range(3): lo, up = m(im1, im2, iters=T, test_mode=1)
    torch.cuda.synchronize()
    def bench(fn, n=20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    eager = bench(lambda: m(im1, im2, iters=T, test_mode=1))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        m(im1, im2, iters=T, test_mode=1)
    torch.cuda.current_stream().wait_stream(s)
    with torch.no_grad(), torch.cuda.graph(g):
        lo_g, up_g = m(im1, im2, iters=T, test_mode=1)
    g.replay(); torch.cuda.synchronize()
    print("max |graph - eager| =", float((up_g - up).abs().max()))
    graph = bench(g.replay)
    print(f"B{B} {H}x{W} T{T}: eager {eager:.3f} ms  graph replay {graph:.3f} ms")
main()
