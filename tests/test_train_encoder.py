"""The CNN encoders in training mode on the HIP kernels (craft_amd/train_encoder.py) against the PyTorch modules under torch
autograd: output tokens, the gradient of every parameter, and BatchNorm's running statistics.  InstanceNorm (fnet), BatchNorm
with batch statistics (cnet in the chairs stage) and with running statistics (cnet under freeze_bn)."""
import copy

import numpy as np
import pytest
import torch

from craft_amd import ops
from craft_amd.extractor import BasicEncoder
from craft_amd.hip import PREC_F16X3, PREC_F32

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _enc(kind, device, seed, out_dim):
    torch.manual_seed(seed)
    enc = BasicEncoder(output_dim=out_dim, norm_fn=kind).to(device)
    with torch.no_grad():
        for m in enc.modules():
            if isinstance(m, torch.nn.Conv2d):
                m.bias.normal_(0, 0.1)
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
                m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 2.0)
    return enc


@pytest.mark.parametrize("prec,tol", [(PREC_F32, 2e-3), (PREC_F16X3, 4e-3)])
@pytest.mark.parametrize("kind,train_stats", [("instance", True), ("batch", True), ("batch", False)])
def test_encoder_training_matches_torch(device, kind, train_stats, prec, tol):
    from craft_amd.train_encoder import encoder_forward_train
    B, H, W, out_dim = 3, 64, 96, 128 if kind == "instance" else 256
    ref = _enc(kind, device, 3, out_dim)
    ref.train(train_stats)
    mine = copy.deepcopy(ref)
    g = torch.Generator(device="cpu").manual_seed(5)
    raw = (torch.rand(B, 3, H, W, generator=g) * 255).to(device)
    G = torch.randn(B, out_dim, H // 8, W // 8, generator=g).to(device)
    y_ref = ref((2 * (raw / 255.0) - 1.0).contiguous())
    (y_ref * G).sum().backward()
    tok = encoder_forward_train(mine, raw, prec)
    assert tok.shape == (B, (H // 8) * (W // 8), out_dim)
    Gt = ops.tokens_from_nchw(G)
    (tok * Gt).sum().backward()
    assert rel_l2(tok.detach(), ops.tokens_from_nchw(y_ref.detach())) < tol
    worst = 0.0
    for (k, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, k
        scale = float(q.grad.norm()) / max(1, q.grad.numel()) ** 0.5
        if kind == "instance" and k.endswith("bias") and k != "conv2.bias":
            # a conv bias in front of InstanceNorm has no effect on the output: both gradients are rounding noise
            assert float(p.grad.abs().max()) < 1e-3 * float(G.abs().max()) * (H * W) ** 0.5, k
            continue
        if kind == "batch" and train_stats and k.endswith("bias") and ("conv" in k or "downsample.0" in k) and k != "conv2.bias":
            continue                                  # same, in front of a training-mode BatchNorm
        e = rel_l2(p.grad, q.grad)
        worst = max(worst, e)
        assert e < 5 * tol, f"{k}: relative L2 {e:.2e} (rms {scale:.2e})"
    if kind == "batch":
        for (k, a), (_, b) in zip(mine.named_buffers(), ref.named_buffers()):
            assert torch.allclose(a.float(), b.float(), rtol=1e-4, atol=1e-5), k
    print(f"[enc train] {kind} train_stats={train_stats} prec={prec}: worst gradient relative L2 {worst:.2e}")


def test_norm_act_kernels_against_torch(device):
    """craft_norm_act_* alone: per-image statistics, residual tail, odd sizes."""
    from craft_amd.train_encoder import NormAct
    from craft_amd.hip import ACT_RELU
    torch.manual_seed(0)
    B, N, C = 2, 777, 96
    x = torch.randn(B, N, C, device=device, requires_grad=True)
    r = torch.randn(B, N, C, device=device, requires_grad=True)
    mean = x.detach().mean(1)
    var = x.detach().var(1, unbiased=False)
    mr = torch.stack([mean, torch.rsqrt(var + 1e-5)], dim=2).contiguous()
    out = NormAct.apply(x, mr, None, None, ACT_RELU, r, N)
    G = torch.randn_like(out)
    (out * G).sum().backward()
    x2 = x.detach().clone().requires_grad_(True)
    r2 = r.detach().clone().requires_grad_(True)
    xh = (x2 - x2.mean(1, keepdim=True)) * torch.rsqrt(x2.var(1, unbiased=False, keepdim=True) + 1e-5)
    ref = torch.relu(r2 + torch.relu(xh))
    (ref * G).sum().backward()
    assert torch.allclose(out, ref, atol=1e-5)
    assert rel_l2(x.grad, x2.grad) < 1e-4 and rel_l2(r.grad, r2.grad) < 1e-6
