"""Developer check: run-to-run spread of the ill-conditioned scalar gradient d loss / d (inter-frame pooling weight)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from craft_amd import CRAFT, default_args
from craft_amd import autograd as AG
from craft_amd.synth import synth_state_dict
key = "corr_fn.setrans.attn_softaggr.feat2score.weight"
dev = torch.device("cuda:0")
for case in ("train_b2_128x192_T3", "train_gma_b2_128x160_T2"):
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", case + ".npz"))
    meta = json.loads(str(z["meta"])); over = meta.get("over", {})
    for henc in (True, False):
        vals = []
        for rep in range(4):
            model = CRAFT(default_args(hip_precision="fp32", dropout_prob=0.0, hip_encoders=henc, **over))
            model.load_state_dict(synth_state_dict(model.state_dict(), seed=meta["seed"], qk_gain=meta["qk_gain"]), strict=True)
            model = model.to(dev).train()
            im1 = torch.from_numpy(z["image1"].astype(np.float32)).to(dev); im2 = torch.from_numpy(z["image2"].astype(np.float32)).to(dev)
            preds = model(im1, im2, iters=meta["iters"])
            loss, _ = AG.sequence_loss(preds, torch.from_numpy(z["flow_gt"]), torch.from_numpy(z["valid"]), meta["gamma"])
            loss.backward()
            vals.append(float(dict(model.named_parameters())[key].grad.reshape(-1)[0]))
        print(case, "hip_encoders", henc, "grads", ["%.6e" % v for v in vals], "reference %.6e" % float(z[f"grad.{key}.v"].reshape(-1)[0]), flush=True)
