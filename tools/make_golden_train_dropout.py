#!/usr/bin/env python3
"""tests/golden/train_dropout_b2_128x160_T2.npz: loss and every parameter gradient of the REFERENCE's training step WITH ITS DROPOUT ON
(core.network.CRAFT imported from /root/reference, model.train(), hidden_dropout_prob 0.1 / attention_probs_dropout_prob 0.2), the masks
being DATA instead of draws from torch's generator: nn.Dropout.forward of the imported model is replaced by `x * mask`, the mask of each call
coming from the counter-based hash of the HIP kernels (tests/dropout_hash.py) with the seed craft_amd/train_forward.py gives that site.  The
script also RECORDS which Dropout modules ran, in order -- the six sites the oracle and the HIP step implement -- and fails on any other.

Only runs in the build container (same layout as tools/make_golden_train.py's fixtures, + the torch seed the pass's seeds derive from).

    python tools/make_golden_train_dropout.py [--check]
"""
import json
import os
import sys

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

CASE = dict(name="train_dropout_b2_128x160_T2", B=2, H=128, W=160, iters=2, seed=1234, qk_gain=2.5, freeze_bn=False, gamma=0.8, torch_seed=20260930)
# module name -> seed offsets of its calls, in call order (craft_amd/train_forward.py)
SITES = {"f2_trans.vispos_encoder.dropout": [1], "f2_trans.setrans.att_dropout": [2], "corr_fn.vispos_encoder.dropout": [3, 4],
         "att.vispos_encoder.dropout": [5], "att.setrans.att_dropout": [6]}


def main():
    from network import CRAFT  # the reference
    c = CASE
    torch.manual_seed(0)
    m = CRAFT(ref_args())
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=c["seed"], qk_gain=c["qk_gain"]), strict=True)
    m.train()
    base = pass_base(c["torch_seed"])
    names = {id(mod): n for n, mod in m.named_modules()}
    calls, count = [], {}

    def fwd(self, x):
        n = names[id(self)]
        k = count.get(n, 0)
        count[n] = k + 1
        assert self.training and n in SITES and k < len(SITES[n]), f"unexpected dropout call: {n} #{k}"
        calls.append([n, list(x.shape), self.p])
        seed = base + SITES[n][k]
        mask = probs_mask(seed, x.shape[0], x.shape[1], x.shape[2], self.p) if x.dim() == 4 else token_mask(seed, tuple(x.shape), self.p)
        return x * mask

    orig = torch.nn.Dropout.forward
    torch.nn.Dropout.forward = fwd
    try:
        B, H, W = c["B"], c["H"], c["W"]
        im1, im2, flow = synth_pair(B, H, W, seed=c["seed"])
        g = torch.Generator().manual_seed(c["seed"] + 77)
        gt = (flow + 0.5 * torch.randn(B, 2, H, W, generator=g)).float()
        valid = (torch.rand(B, H, W, generator=g) > 0.2).float()
        preds = m(im1, im2, iters=c["iters"])
        loss, metrics = ref_sequence_loss()(preds, gt, valid, c["gamma"])
        loss.backward()
    finally:
        torch.nn.Dropout.forward = orig
    assert sum(len(v) for v in SITES.values()) == len(calls) == 6, calls
    out = {"meta": json.dumps(dict(name=c["name"], B=B, H=H, W=W, iters=c["iters"], seed=c["seed"], qk_gain=c["qk_gain"], freeze_bn=c["freeze_bn"],
                                   gamma=c["gamma"], dropout=True, torch_seed=c["torch_seed"], dropout_base=base, dropout_calls=calls, over={},
                                   torch=torch.__version__)),
           "image1": im1.numpy().astype(np.uint8), "image2": im2.numpy().astype(np.uint8), "flow_gt": gt.numpy(), "valid": valid.numpy(),
           "loss": np.float64(loss.item()), "metrics": np.array([metrics["epe"], metrics["1px"], metrics["3px"], metrics["5px"]])}
    unused, seen = [], set()
    for k, p in m.named_parameters():
        if id(p) in seen:
            continue
        seen.add(id(p))
        if p.grad is None:
            unused.append(k)
            continue
        for kk, v in sample(p.grad).items():
            out[f"grad.{k}.{kk}"] = v
    out["unused"] = np.array(json.dumps(unused))
    for k, v in m.state_dict().items():
        if k.startswith("cnet.") and (k.endswith("running_mean") or k.endswith("running_var")):
            out[f"bn.{k}"] = v.numpy()
    for it, p in enumerate(preds):
        for kk, v in sample(p).items():
            out[f"up{it}.{kk}"] = v
    path = os.path.join(ROOT, "tests", "golden", c["name"] + ".npz")
    if "--check" in sys.argv:                      # the generator is a no-op on the committed tree
        z = np.load(path)
        bad = [k for k in out if k != "meta" and not np.array_equal(z[k], out[k])]
        print("differs:", bad if bad else "nothing")
        raise SystemExit(1 if bad else 0)
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes; loss", float(loss), "; dropout calls:", calls, "; unused:", len(unused))


if __name__ == "__main__":
    main()
