"""Loss per step and gradient-buffer health of a few Trainer steps (diagnostics).  usage: python tools/debug_train_steps.py [loss_scale] [steps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from craft_amd import CRAFT, default_args
from craft_amd.synth import synth_pair, synth_state_dict
from craft_amd.train import Trainer
ls = sys.argv[1] if len(sys.argv) > 1 else "auto"
ls = ls if ls == "auto" else float(ls)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda")
model = CRAFT(default_args(hip_precision="train_f16x3"))
model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
model = model.to(dev)
tr = Trainer(model, lr=4e-4, wdecay=1e-4, num_steps=100000, iters=12, clip=1.0, loss_scale=ls)
im1, im2, flow = synth_pair(8, 368, 496, seed=100)
valid = torch.ones(8, 368, 496)
for i in range(steps):
    w0 = tr.optimizer.flat.clone()
    m = tr.step(im1, im2, flow, valid)
    g = tr.optimizer.flat_grad
    bad = int((~torch.isfinite(g)).sum())
    print(f"step {i}: loss {m['loss']:.4f} scale {tr.last_loss_scale:g} |g|/scale {float(g[torch.isfinite(g)].double().norm()) / tr.last_loss_scale:.4e} non-finite {bad} "
          f"max|dw| {float((tr.optimizer.flat - w0).abs().max()):.3e}", flush=True)
    if bad:
        off = 0
        for (k, p), o in zip(model.named_parameters(), tr.optimizer.offsets):
            n = p.numel()
            nb = int((~torch.isfinite(g[o:o + n])).sum())
            if nb:
                print(f"    {k}: {nb} of {n} non-finite")
