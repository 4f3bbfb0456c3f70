#!/usr/bin/env python3
"""Random-shape parity of the TRAINING step (GPU): loss and every parameter gradient of the HIP forward + backward against torch
autograd over the CPU oracle (pinned to the reference by tests/test_oracle_train_golden.py), on seeded random draws of image size,
batch, iteration count, BatchNorm mode and model variant (--setrans / GMA attention, cross-attention / plain correlation).

    python tools/fuzz_train_parity.py [n_draws] [seed0]
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from craft_amd import CRAFT, default_args  # noqa: E402
from craft_amd import autograd as AG  # noqa: E402
from craft_amd.synth import synth_pair, synth_state_dict  # noqa: E402
from oracle import craft_oracle as O  # noqa: E402

VARIANTS = [dict(), dict(use_setrans=False), dict(craft=False), dict(craft=False, use_setrans=False),
            dict(inter_pos_code_type="lsinu", intra_pos_code_type="lsinu"), dict(use_setrans=False, num_heads=2)]
H_MIN, H_MAX = int(os.environ.get("FUZZ_H_MIN", 64)), int(os.environ.get("FUZZ_H_MAX", 136))
W_MIN, W_MAX = int(os.environ.get("FUZZ_W_MIN", 64)), int(os.environ.get("FUZZ_W_MAX", 200))


def draw(seed):
    r = random.Random(seed)
    while True:
        H, W = 8 * r.randint(H_MIN // 8, H_MAX // 8), 8 * r.randint(W_MIN // 8, W_MAX // 8)
        if ((H // 8) * (W // 8)) % 4 == 0 and min(H, W) >= 64:
            break
    return dict(H=H, W=W, B=r.randint(1, 2), iters=r.randint(1, 3), freeze_bn=r.random() < 0.4, over=r.choice(VARIANTS), seed=seed,
                gamma=r.choice([0.8, 0.85]))


def run_one(c, dev, l2_tol=3e-2, verbose=True):       # fp32 policy; observed worst over 24 draws: 2.5e-2 (a BatchNorm gamma at 96x64), typically <= 1e-2
    over = c["over"]
    model = CRAFT(default_args(hip_precision="fp32", dropout_prob=0.0, **over))
    sd0 = synth_state_dict(model.state_dict(), seed=c["seed"], qk_gain=2.5)
    model.load_state_dict(sd0, strict=True)
    model = model.to(dev).train()
    if c["freeze_bn"]:
        model.freeze_bn()
    im1, im2, flow = synth_pair(c["B"], c["H"], c["W"], seed=c["seed"] + 1)
    valid = (torch.rand(c["B"], c["H"], c["W"], generator=torch.Generator().manual_seed(c["seed"])) > 0.15).float()
    preds = model(im1.to(dev), im2.to(dev), iters=c["iters"])
    loss, _ = AG.sequence_loss(preds, flow, valid, c["gamma"])
    loss.backward()
    names = [k for k, _ in model.named_parameters()]
    sd = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in sd0.items()}
    if "corr_fn.setrans.key.weight" in sd:
        sd["corr_fn.setrans.key.weight"], sd["corr_fn.setrans.key.bias"] = sd["corr_fn.setrans.query.weight"], sd["corr_fn.setrans.query.bias"]
    preds_r, _ = O.craft_train_forward(sd, O.OracleConfig(**over), im1, im2, iters=c["iters"], freeze_bn=c["freeze_bn"])
    loss_r, _ = O.sequence_loss(preds_r, flow, valid, c["gamma"])
    loss_r.backward()
    err_loss = abs(float(loss.detach()) - float(loss_r.detach())) / abs(float(loss_r.detach()))
    rms = sorted(float(sd[k].grad.pow(2).mean().sqrt()) for k in names if sd[k].grad is not None)
    scale = rms[len(rms) // 2]
    worst, worst_k, bad = 0.0, "", []
    seen = set()
    for k, p in model.named_parameters():
        if id(p) in seen or sd[k].grad is None or k.startswith("corr_fn.setrans.key."):
            continue
        seen.add(id(p))
        ref = sd[k].grad
        if p.grad is None:
            bad.append(f"{k}: no gradient")
            continue
        if float(ref.pow(2).mean().sqrt()) < 1e-4 * scale:                 # mathematically zero: only has to be small
            if float(p.grad.pow(2).mean().sqrt()) > 1e-3 * scale:
                bad.append(f"{k}: should vanish")
            continue
        l2 = ((p.grad.cpu() - ref).norm() / ref.norm()).item()
        tol = 10 * l2_tol if p.numel() == 1 else l2_tol                    # (ill-conditioned scalars: tests/test_train_backward.py)
        # a scalar parameter's gradient is ONE number; the inter-frame pooling weight's is a sum over N^2 x modes scores that cancels to
        # ~1e-3 of its absolute mass (tests/test_train_backward.py, tools/scalar_grad_noise.py).  When it cancels to far below the typical
        # parameter gradient, its RELATIVE error is noise (draw 11727: -7.0e-6 against -2.4e-4 at a median gradient rms of 4.3e-2): such a
        # scalar is held to an ABSOLUTE error of 1 % of the median parameter-gradient rms instead
        if p.numel() == 1 and abs(float(p.grad.reshape(-1)[0]) - float(ref.reshape(-1)[0])) <= 1e-2 * scale:
            continue
        if l2 > tol or l2 != l2:
            extra = f" (value {float(p.grad.reshape(-1)[0]):.4e} vs {float(ref.reshape(-1)[0]):.4e}; median parameter-gradient rms {scale:.2e})" if p.numel() == 1 else ""
            bad.append(f"{k}: relative L2 {l2:.2e}{extra}")
        if p.numel() > 1 and l2 > worst:
            worst, worst_k = l2, k
    if err_loss > 1e-4:
        bad.append(f"loss {float(loss.detach()):.6f} vs {float(loss_r.detach()):.6f}")
    note = ""
    if bad and err_loss <= 1e-4 and not any("no gradient" in b_ or "should vanish" in b_ for b_ in bad):
        # Before a draw counts as a failure: is it CONDITIONED well enough for the bound to mean anything?  The oracle's own fp32 gradients are
        # compared with a float64 evaluation of the same oracle; when THEY differ by more than a third of the bound the draw measures summation
        # order, not the kernels (draw 12618, a 12 x 8-token image: fp32 oracle vs float64 oracle 1.6e-2 .. 2.3e-2 on every fnet weight).  Such a
        # draw is then judged against the float64 gradients with a bound of max(l2_tol, 4 x the fp32 oracle's own worst deviation).
        sd64 = {k: (v.clone().double().requires_grad_(True) if k in names else (v.clone().double() if v.is_floating_point() else v.clone())) for k, v in sd0.items()}
        if "corr_fn.setrans.key.weight" in sd64:
            sd64["corr_fn.setrans.key.weight"], sd64["corr_fn.setrans.key.bias"] = sd64["corr_fn.setrans.query.weight"], sd64["corr_fn.setrans.query.bias"]
        p64, _ = O.craft_train_forward(sd64, O.OracleConfig(**over), im1.double(), im2.double(), iters=c["iters"], freeze_bn=c["freeze_bn"])
        l64, _ = O.sequence_loss(p64, flow.double(), valid.double(), c["gamma"])
        l64.backward()
        big = [k for k in names if sd64[k].grad is not None and sd[k].grad is not None and sd64[k].numel() > 1
               and float(sd64[k].grad.pow(2).mean().sqrt()) >= 1e-4 * scale]
        own = max(float((sd[k].grad.double() - sd64[k].grad).norm() / sd64[k].grad.norm()) for k in big)
        if own > l2_tol / 3:
            tol64 = max(l2_tol, 4 * own)
            params = dict(model.named_parameters())
            hip64 = max(float((params[k].grad.cpu().double() - sd64[k].grad).norm() / sd64[k].grad.norm()) for k in big if params[k].grad is not None)
            note = f"  [ill-conditioned draw: fp32 oracle vs float64 oracle {own:.1e}; HIP vs float64 {hip64:.1e}, bound {tol64:.1e}]"
            if hip64 <= tol64:
                bad = []
    if verbose:
        print(f"draw {c['seed']}: {c['H']}x{c['W']} B={c['B']} T={c['iters']} freeze_bn={c['freeze_bn']} {over or 'canonical'}: loss err {err_loss:.1e}, "
              f"worst gradient L2 {worst:.1e} ({worst_k}){'  FAIL ' + '; '.join(bad[:4]) if bad else ''}{note}", flush=True)
    return bad, worst


def sweep(n, seed0=5000, verbose=True):
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(32, torch.get_num_threads()))
    nbad, worst = 0, 0.0
    for i in range(n):
        bad, w = run_one(draw(seed0 + i), dev, verbose=verbose)
        nbad += bool(bad)
        worst = max(worst, w)
    return nbad, worst


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    nbad, worst = sweep(n, s0)
    print(f"{n} draws, {nbad} failed, worst gradient relative L2 {worst:.2e}")
    sys.exit(1 if nbad else 0)
