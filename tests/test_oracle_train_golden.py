"""Pin the oracle's TRAINING step (oracle.craft_train_forward + oracle.sequence_loss under torch autograd) to the reference:
loss, the gradient of every parameter, the unused-parameter set and the BatchNorm running statistics after the step, against
tests/golden/train_*.npz captured from the imported reference in model.train() with dropout forced to 0
(tools/make_golden_train.py).  CPU; the GPU tests then compare the HIP backward with the same fixtures and with this oracle."""
import json
import os

import numpy as np
import pytest
import torch

from craft_amd import CRAFT, default_args
from craft_amd.synth import synth_state_dict
from golden_util import GOLDEN_DIR, sample_idx
from oracle import craft_oracle as O

TRAIN_CASES = ["train_b2_128x192_T3", "train_freezebn_b2_128x160_T2",
               # the reference's other shipped training configurations (train-craft-f2full-gma.sh, plain correlation, train-gma.sh)
               "train_gma_b2_128x160_T2", "train_nocraft_b2_128x160_T2", "train_plaingma_b2_128x160_T2",
               # what CRAFT.forward accepts under model.train() beyond the shipped scripts: the two-way correlation of --f1 shared | private
               # (corr.py:164-171) and GMA's relative-position scores (gma.py:34-50, :84-98)
               "train_f1shared_b2_128x160_T2", "train_f1private_b2_128x160_T2", "train_gmapos_b2_128x160_T2", "train_gmaposonly_b2_128x160_T2",
               # --interpos / --intrapos lsinu: the learned sinusoidal embedding and its pos_fc gradients
               "train_lsinu_b2_128x160_T2",
               # --num_heads 2 with GMA's attention: head merge + the aggregator's `project` (gma.py:123-126, :133-138)
               "train_gmaheads2_b2_128x160_T2",
               # the canonical configuration WITH the reference's dropout on (0.1 / 0.2), the masks handed in as data: nn.Dropout.forward of the
               # imported reference replaced by x * mask (tools/make_golden_train_dropout.py), the oracle fed the same masks (DROPOUT_MASKS)
               "train_dropout_b2_128x160_T2"]


def grad_scale(z):
    """Median RMS over all parameter gradients of a fixture: the yardstick for "this gradient is zero up to rounding"."""
    r = [np.sqrt(z[k][1] / max(1, int(np.prod(z[k[:-2] + ".shape"])))) for k in z.files if k.startswith("grad.") and k.endswith(".s")]
    return float(np.median(r))


def grad_check(z, key, g, rtol, atol_rel):
    """g against the strided sample and the second moment of the reference gradient `key`; atol scales with the gradient's RMS.
    Gradients that are mathematically zero (a conv bias in front of a normalisation layer: rounding noise ~1e-7 of the
    typical gradient in the reference) only have to be as small here."""
    ref_v, ref_s = z[f"grad.{key}.v"], z[f"grad.{key}.s"]
    a = g.detach().float().cpu().contiguous().numpy().reshape(-1)
    assert tuple(z[f"grad.{key}.shape"]) == tuple(g.shape), key
    rms = float(np.sqrt(ref_s[1] / a.size))
    scale = grad_scale(z)
    if rms < 1e-4 * scale:
        assert float(np.sqrt((a.astype(np.float64) ** 2).mean())) < 1e-3 * scale, f"{key}: should vanish"
        return
    got = a[sample_idx(a.size)]
    err = np.abs(got - ref_v)
    tol = atol_rel * rms + rtol * np.abs(ref_v) + 1e-12
    assert np.all(err <= tol), f"{key}: max|d|={err.max():.3e} (rms {rms:.3e}) at {int(np.argmax(err - tol))}"
    s2 = float((a.astype(np.float64) ** 2).sum())
    assert abs(s2 - ref_s[1]) <= 20 * (rtol + atol_rel) * ref_s[1] + 1e-20, f"{key}: sum of squares {s2:.6e} vs {ref_s[1]:.6e}"


@pytest.mark.parametrize("case", TRAIN_CASES)
def test_oracle_training_step_matches_reference(case):
    z = np.load(os.path.join(GOLDEN_DIR, case + ".npz"))
    meta = json.loads(str(z["meta"]))
    over = meta.get("over", {})
    model = CRAFT(default_args(**over))
    sd = synth_state_dict(model.state_dict(), seed=meta["seed"], qk_gain=meta["qk_gain"])
    names = [k for k, _ in model.named_parameters()]
    sd = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    if "corr_fn.setrans.key.weight" in sd:
        sd["corr_fn.setrans.key.weight"], sd["corr_fn.setrans.key.bias"] = sd["corr_fn.setrans.query.weight"], sd["corr_fn.setrans.query.bias"]
    if over.get("f1trans") == "shared":               # --f1 shared: f1_trans IS f2_trans (network.py:94-97), one set of Parameters under two names
        for k in [k for k in sd if k.startswith("f1_trans.")]:
            sd[k] = sd["f2_trans." + k[len("f1_trans."):]]
    im1, im2 = torch.from_numpy(z["image1"].astype(np.float32)), torch.from_numpy(z["image2"].astype(np.float32))
    if meta.get("dropout"):
        from dropout_hash import pass_masks
        O.DROPOUT_MASKS = pass_masks(meta["dropout_base"], meta["B"], (meta["H"] // 8) * (meta["W"] // 8))
    try:
        preds, bn = O.craft_train_forward(sd, O.OracleConfig(**over), im1, im2, iters=meta["iters"], freeze_bn=meta["freeze_bn"])
    finally:
        O.DROPOUT_MASKS = None
    loss, metrics = O.sequence_loss(preds, torch.from_numpy(z["flow_gt"]), torch.from_numpy(z["valid"]), meta["gamma"])
    loss.backward()
    assert float(loss) == pytest.approx(float(z["loss"]), rel=2e-5)
    assert [metrics["epe"], metrics["1px"], metrics["3px"], metrics["5px"]] == pytest.approx(z["metrics"].tolist(), rel=1e-4, abs=1e-5)
    unused = json.loads(str(z["unused"]))
    checked = 0
    for k in names:
        if k.startswith("corr_fn.setrans.key."):
            continue                                  # the same Parameter as .query (tied): the reference lists it once
        if k in unused:
            assert sd[k].grad is None, f"{k}: the reference leaves this parameter without a gradient"
            continue
        if sd[k].grad is None:
            # feat2score biases cancel inside their softmax: the oracle never reads them; the reference computes a gradient
            # that is zero up to rounding
            assert k.endswith("feat2score.bias") and z[f"grad.{k}.s"][1] < 1e-10, k
        else:
            grad_check(z, k, sd[k].grad, rtol=2e-3, atol_rel=2e-3)
        checked += 1
    assert checked == len([k for k in names if not k.startswith("corr_fn.setrans.key.")]) - len(unused) and checked >= 130      # (143 in the canonical model)
    if meta["freeze_bn"]:
        assert not bn
    for k in [f for f in z.files if f.startswith("bn.")]:
        name = k[3:].replace(".downsample.1.", ".norm3.")      # one module registered under two names (extractor.py:21-47)
        ref = z[k]
        got = bn[name].numpy() if not meta["freeze_bn"] else sd[name].numpy()
        assert np.allclose(got, ref, rtol=1e-4, atol=1e-6), name
