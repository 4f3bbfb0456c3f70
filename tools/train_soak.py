"""Soak: a few hundred training steps on a small fixed set of synthetic pairs -- the loss must fall and stay finite.
usage (GPU box): python tools/train_soak.py [policy] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from craft_amd import CRAFT, default_args
from craft_amd.synth import synth_pair, synth_state_dict
from craft_amd.train import Trainer

pol = sys.argv[1] if len(sys.argv) > 1 else "train_f16x3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = CRAFT(default_args(hip_precision=pol))
model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
tr = Trainer(model.to(dev), lr=2e-4, wdecay=1e-5, num_steps=steps, iters=6, clip=1.0)
data = []
for s in range(4):
    im1, im2, flow = synth_pair(4, 128, 192, seed=300 + s, max_flow=6)
    data.append((im1.to(dev), im2.to(dev), flow.to(dev), torch.ones(4, 128, 192, device=dev)))
hist = []
for i in range(steps):
    m = tr.step(*data[i % len(data)])
    hist.append(m["loss"])
    if i % 25 == 0 or i == steps - 1:
        print(f"{pol} step {i:4d}: loss {m['loss']:.4f}  epe {m['epe']:.4f}", flush=True)
assert all(h == h and h < 1e5 for h in hist), "non-finite loss"
first, last = sum(hist[:8]) / 8, sum(hist[-8:]) / 8
print(f"{pol}: mean loss of the first 8 steps {first:.4f} -> last 8 steps {last:.4f}")
assert last < 0.7 * first, "the loss did not fall"
