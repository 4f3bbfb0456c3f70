"""Developer check: loss / gradient errors of the training policies against the reference capture (tests/golden/train_*.npz).
usage (GPU box): python tools/train_policy_err.py [policy ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from craft_amd import CRAFT, default_args
from craft_amd import autograd as AG
from craft_amd.synth import synth_state_dict
from golden_util import GOLDEN_DIR, sample_idx
TRAIN_CASES = ["train_b2_128x192_T3", "train_freezebn_b2_128x160_T2"]


def main():
    dev = torch.device("cuda:0")
    for pol in (sys.argv[1:] or ["fp32", "train_f16x3", "train_bf16attn", "train_bf16"]):
        for case in TRAIN_CASES:
            z = np.load(os.path.join(GOLDEN_DIR, case + ".npz"))
            meta = json.loads(str(z["meta"]))
            model = CRAFT(default_args(hip_precision=pol, dropout_prob=0.0))
            model.load_state_dict(synth_state_dict(model.state_dict(), seed=meta["seed"], qk_gain=meta["qk_gain"]), strict=True)
            model = model.to(dev).train()
            if meta["freeze_bn"]:
                model.freeze_bn()
            im1 = torch.from_numpy(z["image1"].astype(np.float32)).to(dev)
            im2 = torch.from_numpy(z["image2"].astype(np.float32)).to(dev)
            preds = model(im1, im2, iters=meta["iters"])
            loss, _ = AG.sequence_loss(preds, torch.from_numpy(z["flow_gt"]), torch.from_numpy(z["valid"]), meta["gamma"])
            loss.backward()
            errs = {}
            for k, p in model.named_parameters():
                if p.grad is None or f"grad.{k}.v" not in z.files:
                    continue
                g = p.grad.detach().float().cpu().numpy().ravel()
                idx = sample_idx(g.size)
                ref = z[f"grad.{k}.v"].astype(np.float64)
                got = g[idx].astype(np.float64)
                den = np.linalg.norm(ref)
                if den > 1e-12 and g.size > 1:
                    errs[k] = float(np.linalg.norm(got - ref) / den)
            worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
            groups = {}
            for k, e in errs.items():
                groups.setdefault(k.split(".")[0], []).append(e)
            print(f"{pol:16s} {case:34s} loss rel err {abs(float(loss) - float(z['loss'])) / abs(float(z['loss'])):.2e}  worst {worst[0][1]:.2e} ({worst[0][0]})")
            print("    per module max: " + ", ".join(f"{m} {max(v):.1e}" for m, v in groups.items()), flush=True)


if __name__ == "__main__":
    main()
