"""The fused update iteration (craft_amd/train_update.py: one autograd node per refinement iteration, hand-written backward, inference
entry points in the forward, weight gradients of all iterations in one launch per layer) against the operator-by-operator training path
of round 2 (train_forward.py with args.hip_fused_update = False), which the golden / oracle tests of test_train_backward.py pin to
the reference (network.py:230-260, update.py:137-162): same predictions, loss and parameter gradients up to summation order."""
import pytest
import torch

from craft_amd import CRAFT, default_args
from craft_amd import autograd as AG
from craft_amd.synth import synth_pair, synth_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda")


def _step(device, fused, over, B, H, W, iters, policy, seed=3):
    model = CRAFT(default_args(hip_precision=policy, dropout_prob=0.0, hip_fused_update=fused, **over))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=seed), strict=True)
    model = model.to(device).train()
    im1, im2, flow = synth_pair(B, H, W, seed=11)
    preds = model(im1.to(device), im2.to(device), iters=iters)
    loss, _ = AG.sequence_loss(preds, flow, torch.ones(B, H, W), 0.8)
    loss.backward(torch.full((), 1024.0, device=device))
    return float(loss.detach()), [p.detach() for p in preds], {k: p.grad.detach().clone() / 1024.0 for k, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("over,policy", [({}, "train_f16x3"), ({}, "train_amp_bf16"), (dict(use_setrans=False), "train_f16x3"),
                                         (dict(craft=False, use_setrans=False), "train_f16x3")])
@pytest.mark.parametrize("B,H,W,iters", [(2, 128, 160, 3), (1, 136, 200 - 8, 2)])
def test_fused_iteration_equals_operator_path(device, over, policy, B, H, W, iters):
    la, pa, ga = _step(device, True, over, B, H, W, iters, policy)
    lb, pb, gb = _step(device, False, over, B, H, W, iters, policy)
    bf = "bf16" in policy
    assert la == pytest.approx(lb, rel=5e-3 if bf else 2e-5)
    for a, b in zip(pa, pb):
        assert (a - b).abs().max().item() < (0.25 if bf else 2e-3)        # (bf16: two evaluation orders of an 8-bit-mantissa pipeline)
    assert set(ga) == set(gb)
    rms = sorted(float(g.pow(2).mean().sqrt()) for g in gb.values())
    scale = rms[len(rms) // 2]
    worst, wk = 0.0, None
    for k, g in gb.items():
        if float(g.pow(2).mean().sqrt()) < 1e-4 * scale:
            assert float(ga[k].pow(2).mean().sqrt()) < 1e-3 * scale, k
            continue
        assert ga[k].shape == g.shape, k
        l2 = float((ga[k] - g).norm() / g.norm())
        if g.numel() == 1:
            l2 /= 10.0                              # (the ill-conditioned scalar pooling weights: see test_train_backward.py)
        if l2 > worst:
            worst, wk = l2, k
    print(f"[fused update] {over} {policy} {B}x{H}x{W} T={iters}: loss {la:.6f} / {lb:.6f}, worst relative L2 {worst:.2e} ({wk})")
    assert worst < (0.25 if bf else 1e-2), (wk, worst)
