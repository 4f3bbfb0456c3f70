"""Per-parameter comparison of the HIP training step with a committed reference capture (diagnostics)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from craft_amd import CRAFT, default_args
from craft_amd import autograd as AG
from craft_amd.synth import synth_state_dict
from golden_util import sample_idx
case = sys.argv[1] if len(sys.argv) > 1 else "train_b2_128x192_T3"
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
z = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
meta = json.loads(str(z["meta"]))
dev = torch.device("cuda")
model = CRAFT(default_args(hip_precision=prec, dropout_prob=0.0))
model.load_state_dict(synth_state_dict(model.state_dict(), seed=meta["seed"], qk_gain=meta["qk_gain"]), strict=True)
model = model.to(dev).train()
if meta["freeze_bn"]:
    model.freeze_bn()
im1 = torch.from_numpy(z["image1"].astype(np.float32)).to(dev); im2 = torch.from_numpy(z["image2"].astype(np.float32)).to(dev)
preds = model(im1, im2, iters=meta["iters"])
for it, p in enumerate(preds):
    a = p.detach().cpu().numpy().reshape(-1)[sample_idx(p.numel())]
    print(f"up{it}: max|d| {np.abs(a - z[f'up{it}.v']).max():.3e}")
loss, m = AG.sequence_loss(preds, torch.from_numpy(z["flow_gt"]), torch.from_numpy(z["valid"]), meta["gamma"])
loss.backward()
print("loss", float(loss), float(z["loss"]))
seen = set()
for k, p in model.named_parameters():
    if id(p) in seen: continue
    seen.add(id(p))
    if f"grad.{k}.v" not in z.files or p.grad is None:
        print(f"{k:60s} grad {'None' if p.grad is None else 'set'} ref {'yes' if f'grad.{k}.v' in z.files else 'no'}"); continue
    a = p.grad.detach().float().cpu().numpy().reshape(-1)
    ref = z[f"grad.{k}.v"]; rms = np.sqrt(z[f"grad.{k}.s"][1] / a.size)
    d = a[sample_idx(a.size)] - ref
    err = np.abs(d).max()
    print(f"{k:60s} rms {rms:.3e} max|d| {err:.3e} ratio {err / max(rms, 1e-30):.2e}  relL2 {np.linalg.norm(d) / max(np.linalg.norm(ref), 1e-30):.2e}")
