"""Gradient parity of the training step WITH DROPOUT ON (VERDICT r5 "weak" 2: the timed training legs run ``model.train()`` with the
reference's dropout -- 0.1 on the LayerNorm-ed tokens of every input encoder, 0.2 on the attention probabilities -- while the oracle checks
ran with dropout off + mask statistics).

The HIP step draws its masks from a counter-based hash, keep(i) = mix32(seed * phi + i) >= p * 2^32 over the flat element index i of the
tensor (``craft_dropout`` / the fused softmax kernels: csrc/kernels_train.hip), with one seed per site derived from ``torch.initial_seed()``
and the model's pass counter (craft_amd/train_forward.py).  That hash is RESTATED in numpy (tests/dropout_hash.py: nothing of the product is
called to make a mask), the six masks of a pass are built from the seeds alone and handed to the oracle as data (``oracle.craft_oracle.DROPOUT_MASKS``), and
the oracle's loss, predictions and every parameter gradient under torch autograd are compared with the HIP step's.  The CPU part pins the
restated hash to the kernel's published constants and statistics; the GPU part is the parity test proper.
"""
import numpy as np
import pytest
import torch

from dropout_hash import MASK64, dropout_scale, mix32, pass_base, pass_masks, probs_mask, token_mask


def test_restated_hash_constants_and_statistics():
    """fmix64 known answers (computed from its definition in Python integers) and the keep rate of the restated mask."""
    def fmix_py(k):
        k ^= k >> 33
        k = (k * 0xff51afd7ed558ccd) & MASK64
        k ^= k >> 33
        k = (k * 0xc4ceb9fe1a85ec53) & MASK64
        k ^= k >> 33
        return k & 0xFFFFFFFF
    ks = [0, 1, 2, 0x9E3779B97F4A7C15, (1 << 64) - 1, 123456789012345678]
    got = mix32(np.array(ks, dtype=np.uint64))
    assert [int(g) for g in got] == [fmix_py(k) for k in ks]
    for p in (0.1, 0.2, 0.5):
        m = dropout_scale(77, np.arange(1 << 20, dtype=np.uint64), p)
        assert abs((m != 0).mean() - (1 - p)) < 3e-3 and abs(m.max() - 1 / (1 - p)) < 1e-6
    assert not np.array_equal(dropout_scale(77, np.arange(4096, dtype=np.uint64), 0.2), dropout_scale(78, np.arange(4096, dtype=np.uint64), 0.2))


@pytest.mark.gpu
def test_restated_hash_is_the_kernels_mask(device):
    """craft_dropout on a tensor of ones = the restated factor, bit for bit; the fused softmax's dropped copy = softmax x the restated mask."""
    from craft_amd import autograd as AG
    x = torch.ones(3, 70, 256, device=device)
    for p, seed in ((0.1, 5), (0.2, (1 << 58) + 12345), (0.5, 0)):
        y = AG.Dropout.apply(x, p, seed).cpu()
        assert torch.equal(y, token_mask(seed, (3, 70, 256), p)), (p, seed)
    B, M, H8, W8 = 2, 4, 5, 9
    N, ld = H8 * W8, 64
    S = (torch.randn(B, M, N, ld, generator=torch.Generator().manual_seed(1)) * 2).to(device)
    P = AG.AttnSoftmax.apply(S.clone(), None, 0.0, -1, None, (H8, W8)).cpu()[..., :N]
    Pd = AG.AttnSoftmax.apply(S.clone(), None, 0.0, -1, None, (H8, W8), 0.2, 991).cpu()[..., :N]
    assert torch.allclose(Pd, P * probs_mask(991, B, M, N, 0.2), rtol=1e-6, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("policy,B,H,W,iters,freeze_bn", [("fp32", 2, 128, 160, 2, False), ("mixed", 2, 128, 160, 2, False), ("fp32", 1, 136, 192, 3, True), ("train_f16x3", 1, 136, 192, 3, True),      # (frozen BatchNorm, one pair, ragged 17 x 24 grid)
                                                          # the TIMED workload itself: configs[3]'s shape, depth and policy (bench.py train_cfg3), batch 2
                                                          ("mixed", 2, 368, 496, 12, False),
                                                          # ... and configs[4]'s: 368x768, frozen BatchNorm, bf16 MFMA attention (bench.py train_cfg4), batch 1
                                                          ("train_bf16attn", 1, 368, 768, 12, True)])
def test_training_step_with_dropout_on_against_oracle(device, policy, B, H, W, iters, freeze_bn):
    """model.train() with the reference's dropout probabilities (0.1 hidden / 0.2 attention), HIP step against torch autograd over the CPU
    oracle fed the SAME masks: loss, every prediction, every parameter gradient, bounds of the dropout-off step tests."""
    from craft_amd import CRAFT, default_args
    from craft_amd import autograd as AG
    from craft_amd.synth import synth_pair, synth_state_dict
    from craft_amd.train import auto_loss_scale
    from oracle import craft_oracle as O
    model = CRAFT(default_args(hip_precision=policy, hip_loss_scaled=True))            # dropout_prob left at its default: the config's 0.1 / 0.2
    sd0 = synth_state_dict(model.state_dict(), seed=77)
    model.load_state_dict(sd0, strict=True)
    model = model.to(device).train()
    if freeze_bn:
        model.freeze_bn()
    ph, pa = model.f2_trans.config.hidden_dropout_prob, model.f2_trans.config.attention_probs_dropout_prob
    assert (ph, pa) == (0.1, 0.2) and model.att.config.hidden_dropout_prob == 0.1 and model.corr_fn.config.hidden_dropout_prob == 0.1
    im1, im2, flow = synth_pair(B, H, W, seed=31)
    valid = (torch.rand(B, H, W, generator=torch.Generator().manual_seed(1)) > 0.15).float()
    torch.manual_seed(20260930)
    model.__dict__["_train_calls"] = 0
    base = pass_base(torch.initial_seed())                                               # craft_amd/train_forward.py: the pass's seed schedule
    preds = model(im1.to(device), im2.to(device), iters=iters)
    loss, _ = AG.sequence_loss(preds, flow, valid, 0.8)
    ls = auto_loss_scale(flow.numel())
    loss.backward(torch.full((), ls, device=loss.device))
    for p_ in model.parameters():
        if p_.grad is not None:
            p_.grad.mul_(1.0 / ls)
    # ---- the oracle with the same six masks (sites and seeds: train_forward.py; mask = the restated hash, nothing of the product)
    masks = pass_masks(base, B, (H // 8) * (W // 8), 4, ph, pa)
    names = [k for k, _ in model.named_parameters()]
    sd = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in sd0.items()}
    sd["corr_fn.setrans.key.weight"], sd["corr_fn.setrans.key.bias"] = sd["corr_fn.setrans.query.weight"], sd["corr_fn.setrans.query.bias"]
    torch.set_num_threads(min(32, torch.get_num_threads()))
    O.DROPOUT_MASKS = masks
    try:
        preds_r, _ = O.craft_train_forward(sd, O.OracleConfig(), im1, im2, iters=iters, freeze_bn=freeze_bn)
        loss_r, _ = O.sequence_loss(preds_r, flow, valid, 0.8)
        loss_r.backward()
    finally:
        O.DROPOUT_MASKS = None
    # (the masks matter: the same oracle step without them lands somewhere else)
    with torch.no_grad():
        preds_0, _ = O.craft_train_forward({k: v.detach() for k, v in sd.items()}, O.OracleConfig(), im1, im2, iters=iters, freeze_bn=freeze_bn)
    assert (preds_0[-1] - preds_r[-1].detach()).abs().max().item() > 0.05, "dropout must change the prediction"
    tight = policy in ("fp32", "train_f16x3")           # fp32-class operands everywhere (measured: 5.8e-3 / 5.3e-3 worst gradient error)
    # (loss relative, predictions px, gradient relative L2): fp32-class = the dropout-off tests' bounds; mixed = 2 x the worst figures measured on
    # the MI355X with dropout on (4.0e-3 px at 368x496 x 12 iterations; 1.8e-2, corr_fn.setrans.query.weight at 128x160 -- the masks thin the
    # sums its gradient is made of, the dropout-off step measures 5e-3 there); train_bf16attn = tests/test_cfg_step_parity.py's oracle bounds
    loss_rel, pred_px, l2_bound = (3e-5, 2e-3, 1e-2) if tight else (2e-5, 0.13, 0.11) if policy == "train_bf16attn" else (1e-4, 8e-3, 4e-2)
    assert float(loss.detach()) == pytest.approx(float(loss_r.detach()), rel=loss_rel)
    pred_err = max((a.detach().cpu() - b.detach()).abs().max().item() for a, b in zip(preds, preds_r))
    assert pred_err < pred_px, f"prediction error {pred_err:.2e} px"
    grads_r = {k: sd[k].grad for k in names}
    rms_all = sorted(float(g.pow(2).mean().sqrt()) for g in grads_r.values() if g is not None)
    scale = rms_all[len(rms_all) // 2]
    worst, worst_k, checked, seen = 0.0, None, 0, set()
    for k, p in model.named_parameters():
        if id(p) in seen or grads_r[k] is None or k.startswith("corr_fn.setrans.key."):
            continue
        seen.add(id(p))
        ref = grads_r[k]
        if float(ref.pow(2).mean().sqrt()) < 1e-4 * scale:          # mathematically zero (a bias in front of a normalisation layer)
            assert p.grad is None or float(p.grad.pow(2).mean().sqrt()) < 1e-3 * scale, k
            continue
        assert p.grad is not None, k
        l2 = ((p.grad.cpu() - ref).norm() / ref.norm()).item()
        mul = 15.0 if p.numel() == 1 else 1.0                        # (one ill-conditioned number: tests/test_train_backward.py explains)
        if l2 / mul > worst:
            worst, worst_k = l2 / mul, k
        checked += 1
        if policy == "train_bf16attn" and p.numel() == 1:
            continue                      # (one number: its "L2" is its own relative error -- as in tests/test_cfg_step_parity.py)
        assert l2 < mul * l2_bound, f"{k}: relative L2 error {l2:.2e}"
    assert checked >= 100
    print(f"[dropout-on parity] {policy} B={B} {H}x{W} T={iters} freeze_bn={freeze_bn}: loss {float(loss.detach()):.6f} vs oracle {float(loss_r.detach()):.6f}, "
          f"predictions {pred_err:.2e} px, worst relative L2 gradient error {worst:.2e} ({worst_k}), {checked} parameters")
