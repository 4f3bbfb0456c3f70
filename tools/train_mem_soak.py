"""Full-size soak: N training steps at BASELINE configs[3] / [4] on a few fixed synthetic batches -- loss falls, step time and peak memory
stay flat (no leak through the per-step pools / packs).  usage: python tools/train_mem_soak.py [3|4] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from craft_amd import CRAFT, default_args
from craft_amd.synth import synth_pair, synth_state_dict
from craft_amd.train import Trainer

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
H, W, B, policy = {3: (368, 496, 8, "mixed"), 4: (368, 768, 4, "train_bf16attn")}[cfg]
dev = torch.device("cuda:0")
model = CRAFT(default_args(hip_precision=policy))
model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
tr = Trainer(model.to(dev), lr=2e-4, wdecay=1e-5, num_steps=steps, iters=12, clip=1.0, freeze_bn=cfg != 3)
data = []
for s in range(3):
    im1, im2, flow = synth_pair(B, H, W, seed=500 + s, max_flow=8)
    data.append((im1.to(dev), im2.to(dev), flow.to(dev), torch.ones(B, H, W, device=dev)))
t0 = time.perf_counter()
for i in range(steps):
    m = tr.step(*data[i % len(data)])
    if i % 10 == 9 or i == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print(f"step {i + 1:4d}: loss {m['loss']:9.4f} epe {m['epe']:7.4f}  {1e3 * (t1 - t0) / (10 if i else 1):7.2f} ms/step  allocated {torch.cuda.memory_allocated() / 1e9:6.2f} GB  "
              f"reserved {torch.cuda.memory_reserved() / 1e9:6.2f} GB  peak {torch.cuda.max_memory_allocated() / 1e9:6.2f} GB", flush=True)
        t0 = t1
