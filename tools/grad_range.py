"""Magnitudes of the gradients that enter the backward operators of one training step (diagnostics for the loss scale):
per autograd.Function the largest |g| and the smallest / largest RMS over its calls, UNSCALED and relative to the initial loss gradient
1 / (B*2*H*W).  usage: python tools/grad_range.py [--B 8 --H 368 --W 496 --iters 12 --policy train_f16x3]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from craft_amd import CRAFT, default_args
from craft_amd import autograd as AG
from craft_amd import train_encoder as TE
from craft_amd.synth import synth_pair, synth_state_dict
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8); ap.add_argument("--H", type=int, default=368); ap.add_argument("--W", type=int, default=496)
ap.add_argument("--iters", type=int, default=12); ap.add_argument("--policy", default="fp32"); ap.add_argument("--seed", type=int, default=1234)
a = ap.parse_args()
stats = {}


def wrap(cls):
    orig = cls.backward

    def bw(ctx, *gs):
        for g in gs:
            if isinstance(g, torch.Tensor) and g.numel() > 1 and g.is_floating_point():
                st = stats.setdefault(cls.__name__, [0, 0.0, float("inf"), 0.0])
                st[0] += 1
                st[1] = max(st[1], float(g.abs().max()))
                r = float(g.float().pow(2).mean().sqrt())
                st[2] = min(st[2], r); st[3] = max(st[3], r)
        return orig(ctx, *gs)
    cls.backward = staticmethod(bw)


for mod in (AG, TE):
    for name in dir(mod):
        c = getattr(mod, name)
        if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function and c.__module__ == mod.__name__:
            wrap(c)
dev = torch.device("cuda")
m = CRAFT(default_args(hip_precision=a.policy))
m.load_state_dict(synth_state_dict(m.state_dict(), seed=a.seed), strict=True)
m = m.to(dev).train()
im1, im2, flow = synth_pair(a.B, a.H, a.W, seed=100)
preds = m(im1.to(dev), im2.to(dev), iters=a.iters)
loss, _ = AG.sequence_loss(preds, flow, torch.ones(a.B, a.H, a.W), 0.8)
loss.backward()
g0 = 1.0 / (a.B * 2 * a.H * a.W)
print(f"loss {float(loss):.4f}; initial gradient per element g0 = {g0:.3e}; fp32 policy {a.policy}")
print(f"{'Function':18s} {'calls':>6s} {'max|g|':>11s} {'max/g0':>10s} {'rms min':>11s} {'rms min/g0':>11s} {'rms max/g0':>11s}")
for k, (n, mx, rmin, rmax) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:18s} {n:6d} {mx:11.3e} {mx / g0:10.3g} {rmin:11.3e} {rmin / g0:11.3g} {rmax / g0:11.3g}")
