#!/usr/bin/env python3
"""tests/golden/train_traj_b2_128x160_T2.npz: THREE consecutive iterations of the REFERENCE's training loop (train.py:215-236: zero_grad,
forward in model.train(), sequence_loss, backward, clip_grad_norm_, AdamW step, OneCycleLR step) on the imported core.network.CRAFT with the
optimizer and scheduler train.py's own fetch_optimizer builds (compiled from its AST) -- dropout ON, the masks of pass s handed in as data
(tests/dropout_hash.py, base = pass_base(torch_seed, s), as in tools/make_golden_train_dropout.py), a different synthetic batch per step.
Recorded: the three losses and learning rates, a strided sample + (sum, sum^2) of every parameter AFTER the third update and of its change
from the initial value, cnet's BatchNorm running statistics.  craft_amd.train.Trainer.step is held to it on the GPU
(tests/test_trainer_gpu.py::test_three_steps_follow_the_reference_training_loop).

Only runs in the build container.      python tools/make_golden_train_traj.py [--check]
"""
import ast
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference/core")

from craft_amd.synth import synth_pair, synth_state_dict  # noqa: E402
from dropout_hash import pass_base, probs_mask, token_mask  # noqa: E402
from make_golden import ref_args, sample  # noqa: E402
from make_golden_train import ref_sequence_loss  # noqa: E402
from make_golden_train_dropout import SITES  # noqa: E402

CASE = dict(name="train_traj_b2_128x160_T2", B=2, H=128, W=160, iters=2, seed=1234, qk_gain=2.5, gamma=0.8, torch_seed=20260931, steps=3,
            lr=4e-4, wdecay=1e-4, epsilon=1e-8, num_steps=1000, clip=1.0)


def ref_fetch_optimizer():
    path = "/root/reference/train.py"
    tree = ast.parse(open(path).read(), path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "fetch_optimizer"]
    ns = {"optim": torch.optim}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, "exec"), ns)
    return ns["fetch_optimizer"]


def batch(c, s):
    im1, im2, flow = synth_pair(c["B"], c["H"], c["W"], seed=c["seed"] + 10 * s)
    g = torch.Generator().manual_seed(c["seed"] + 77 + s)
    gt = (flow + 0.5 * torch.randn(c["B"], 2, c["H"], c["W"], generator=g)).float()
    valid = (torch.rand(c["B"], c["H"], c["W"], generator=g) > 0.2).float()
    return im1, im2, gt, valid


def main():
    from network import CRAFT  # the reference
    c = CASE
    torch.manual_seed(0)
    m = CRAFT(ref_args())
    sd0 = synth_state_dict(m.state_dict(), seed=c["seed"], qk_gain=c["qk_gain"])
    m.load_state_dict(sd0, strict=True)
    m.train()
    args = SimpleNamespace(lr=c["lr"], wdecay=c["wdecay"], epsilon=c["epsilon"], num_steps=c["num_steps"])
    optimizer, scheduler = ref_fetch_optimizer()(args, m)
    seq_loss = ref_sequence_loss()
    names = {id(mod): n for n, mod in m.named_modules()}
    state = {"base": 0, "count": {}}

    def fwd(self, x):
        n = names[id(self)]
        k = state["count"].get(n, 0)
        state["count"][n] = k + 1
        assert self.training and n in SITES and k < len(SITES[n]), f"unexpected dropout call: {n} #{k}"
        seed = state["base"] + SITES[n][k]
        mask = probs_mask(seed, x.shape[0], x.shape[1], x.shape[2], self.p) if x.dim() == 4 else token_mask(seed, tuple(x.shape), self.p)
        return x * mask

    orig = torch.nn.Dropout.forward
    torch.nn.Dropout.forward = fwd
    losses, lrs = [], []
    try:
        for s in range(c["steps"]):                       # train.py:215-236 (GradScaler disabled: fp32)
            state["base"], state["count"] = pass_base(c["torch_seed"], s), {}
            im1, im2, gt, valid = batch(c, s)
            optimizer.zero_grad()
            preds = m(im1, im2, iters=c["iters"])
            loss, _ = seq_loss(preds, gt, valid, c["gamma"])
            loss.backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), c["clip"])
            lrs.append(optimizer.param_groups[0]["lr"])
            optimizer.step()
            scheduler.step()
            losses.append(float(loss))
    finally:
        torch.nn.Dropout.forward = orig
    out = {"meta": json.dumps(dict(c, dropout=True, over={}, freeze_bn=False, torch=torch.__version__)),
           "losses": np.array(losses, dtype=np.float64), "lrs": np.array(lrs, dtype=np.float64)}
    for s in range(c["steps"]):
        im1, im2, gt, valid = batch(c, s)
        out[f"image1.{s}"], out[f"image2.{s}"] = im1.numpy().astype(np.uint8), im2.numpy().astype(np.uint8)
        out[f"flow_gt.{s}"], out[f"valid.{s}"] = gt.numpy(), valid.numpy()
    seen = set()
    for k, p in m.named_parameters():
        if id(p) in seen:
            continue
        seen.add(id(p))
        for kk, v in sample(p.detach()).items():
            out[f"w.{k}.{kk}"] = v
        for kk, v in sample(p.detach() - sd0[k]).items():
            out[f"dw.{k}.{kk}"] = v
    for k, v in m.state_dict().items():
        if k.startswith("cnet.") and (k.endswith("running_mean") or k.endswith("running_var")):
            out[f"bn.{k}"] = v.numpy()
    path = os.path.join(ROOT, "tests", "golden", c["name"] + ".npz")
    if "--check" in sys.argv:                      # the generator is a no-op on the committed tree
        z = np.load(path)
        bad = [k for k in out if k != "meta" and not np.array_equal(z[k], out[k])]
        print("differs:", bad if bad else "nothing")
        raise SystemExit(1 if bad else 0)
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes; losses", losses, "lrs", lrs)


if __name__ == "__main__":
    main()
