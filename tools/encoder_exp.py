#!/usr/bin/env python3
"""GPU experiment: time the two CNN encoders (PyTorch-ROCm / MIOpen) under different settings."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from craft_amd import CRAFT, default_args
from craft_amd.synth import synth_pair, synth_state_dict

def bench(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3

m = CRAFT(default_args()); m.load_state_dict(synth_state_dict(m.state_dict(), seed=1234)); m = m.cuda().eval()
im1, im2, _ = synth_pair(4, 448, 1024, seed=0); im1 = (2*(im1/255)-1).cuda(); im2 = (2*(im2/255)-1).cuda()
def enc():
    with torch.no_grad():
        a = m.fnet([im1, im2]); b = m.cnet(im1)
    return a, b
ref = enc()
print("fp32 default           %.2f ms" % bench(enc))
torch.backends.cudnn.benchmark = True
print("fp32 cudnn.benchmark   %.2f ms" % bench(enc))
m2 = m.to(memory_format=torch.channels_last)
i1, i2 = im1.contiguous(memory_format=torch.channels_last), im2.contiguous(memory_format=torch.channels_last)
def enc_cl():
    with torch.no_grad():
        a = m2.fnet([i1, i2]); b = m2.cnet(i1)
    return a, b
print("fp32 channels_last     %.2f ms" % bench(enc_cl))
out = enc_cl()
print("  max|diff| fmap1 vs default: %.3e" % (out[0][0] - ref[0][0]).abs().max().item())
def enc16():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        a = m2.fnet([i1, i2]); b = m2.cnet(i1)
    return a, b
print("fp16 autocast (ch_last) %.2f ms" % bench(enc16))
o16 = enc16()
print("  max|diff| fmap1 fp16 vs fp32: %.3e (|fmap| max %.2f)" % ((o16[0][0].float() - ref[0][0]).abs().max().item(), ref[0][0].abs().max().item()))
