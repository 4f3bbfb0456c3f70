"""Developer tool: which call sites create zero-filled tensors / pad copies in one training step (each is a ~5 us kernel)."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from craft_amd import CRAFT, default_args
from craft_amd import autograd as AG
from craft_amd.synth import synth_pair, synth_state_dict

dev = torch.device("cuda:0")
model = CRAFT(default_args(hip_precision="train_f16x3"))
model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
model = model.to(dev).train()
im1, im2, flow = synth_pair(2, 128, 192, seed=100)
im1, im2, flow = im1.to(dev), im2.to(dev), flow.to(dev)
valid = torch.ones(2, 128, 192, device=dev)
hist = collections.Counter()
orig = {n: getattr(torch, n) for n in ("zeros", "zeros_like", "cat")}


def wrap(name):
    def f(*a, **k):
        st = traceback.extract_stack(limit=4)
        site = [s for s in st[:-1] if "craft_amd" in s.filename]
        key = f"{name} " + (f"{os.path.basename(site[-1].filename)}:{site[-1].lineno} {site[-1].name}" if site else "other")
        hist[key] += 1
        return orig[name](*a, **k)
    return f


preds = model(im1, im2, iters=12)          # warm
for n in orig:
    setattr(torch, n, wrap(n))
preds = model(im1, im2, iters=12)
loss, _ = AG.sequence_loss(preds, flow, valid, 0.8)
loss.backward()
for n in orig:
    setattr(torch, n, orig[n])
for k, v in hist.most_common(40):
    print(f"{v:5d}  {k}")
