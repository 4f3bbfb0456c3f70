"""Per-parameter gradient comparison of HIP training steps against torch autograd over the CPU oracle at a given shape (diagnostics).
usage: python tools/train_parity_scan.py [--H 368 --W 496 --B 2 --iters 12 --policies fp32,train_f16x3 --f64]
--f64 also evaluates the oracle in float64 (how far two correct fp32 evaluations may differ: conditioning of each gradient)."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from craft_amd import CRAFT, default_args
from craft_amd import autograd as AG
from craft_amd.synth import synth_pair, synth_state_dict
from oracle import craft_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--H", type=int, default=368); ap.add_argument("--W", type=int, default=496); ap.add_argument("--B", type=int, default=2)
ap.add_argument("--iters", type=int, default=12); ap.add_argument("--policies", default="fp32,train_f16x3")
ap.add_argument("--f64", action="store_true"); ap.add_argument("--loss-scale", default="auto"); ap.add_argument("--freeze-bn", action="store_true"); ap.add_argument("--seed", type=int, default=77)
a = ap.parse_args()
dev = torch.device("cuda")
model = CRAFT(default_args(dropout_prob=0.0))
sd0 = synth_state_dict(model.state_dict(), seed=a.seed)
names = [k for k, _ in model.named_parameters()]
im1, im2, flow = synth_pair(a.B, a.H, a.W, seed=31)
valid = (torch.rand(a.B, a.H, a.W, generator=torch.Generator().manual_seed(1)) > 0.15).float()
torch.set_num_threads(min(32, torch.get_num_threads()))


def oracle(dtype):
    sd = {k: (v.clone().to(dtype).requires_grad_(True) if k in names else (v.clone().to(dtype) if v.is_floating_point() else v.clone())) for k, v in sd0.items()}
    sd["corr_fn.setrans.key.weight"], sd["corr_fn.setrans.key.bias"] = sd["corr_fn.setrans.query.weight"], sd["corr_fn.setrans.query.bias"]
    t0 = time.time()
    preds, _ = O.craft_train_forward(sd, O.OracleConfig(), im1.to(dtype), im2.to(dtype), iters=a.iters, freeze_bn=a.freeze_bn)
    loss, _ = O.sequence_loss(preds, flow.to(dtype), valid.to(dtype), 0.8)
    loss.backward()
    print(f"[oracle {dtype}] loss {float(loss):.6f}  {time.time() - t0:.1f} s", flush=True)
    return float(loss), {k: sd[k].grad.float() for k in names if sd[k].grad is not None}, [p.detach().float() for p in preds]


def report(tag, grads, ref, top=12):
    rows = []
    rms_all = sorted(float(g.pow(2).mean().sqrt()) for g in ref.values())
    scale = rms_all[len(rms_all) // 2]
    for k, g in ref.items():
        if k.startswith("corr_fn.setrans.key.") or k not in grads or float(g.pow(2).mean().sqrt()) < 1e-4 * scale:
            continue                      # (mathematically zero gradients: biases in front of a normalisation layer)
        n = float(g.norm())
        rows.append((float((grads[k] - g).norm()) / max(n, 1e-30), k, n / g.numel() ** 0.5))
    rows.sort(reverse=True)
    print(f"== {tag}: worst relative L2 of {len(rows)} parameter gradients")
    for l2, k, rms in rows[:top]:
        print(f"   {l2:9.3e}  rms {rms:9.3e}  {k}")


l32, g32, p32 = oracle(torch.float32)
if a.f64:
    l64, g64, p64 = oracle(torch.float64)
    report("oracle fp32 vs oracle fp64", g32, g64)
for pol in a.policies.split(","):
    m = CRAFT(default_args(hip_precision=pol, dropout_prob=0.0, hip_loss_scaled=a.loss_scale != "1"))   # (fp16 roles run as fp16 under a loss scale)
    m.load_state_dict(sd0, strict=True)
    m = m.to(dev).train()
    if a.freeze_bn:
        m.freeze_bn()
    preds = m(im1.to(dev), im2.to(dev), iters=a.iters)
    loss, _ = AG.sequence_loss(preds, flow, valid, 0.8)
    from craft_amd.train import auto_loss_scale
    ls = auto_loss_scale(flow.numel()) if a.loss_scale == "auto" else float(a.loss_scale)
    loss.backward(torch.full((), ls, device=loss.device))
    g = {k: p.grad.detach().cpu() / ls for k, p in m.named_parameters() if p.grad is not None}
    dmax = max(float((x.detach().cpu() - y).abs().max()) for x, y in zip(preds, p32))
    print(f"[hip {pol}, loss scale {ls:g}] loss {float(loss):.6f} (oracle {l32:.6f}); max |pred - oracle| {dmax:.3e} px")
    report(f"hip {pol} vs oracle fp32", g, g32)
    if a.f64:
        report(f"hip {pol} vs oracle fp64", g, g64)
    del m, preds, loss
