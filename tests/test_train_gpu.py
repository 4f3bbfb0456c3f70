"""Training-step kernels on the GPU: sequence_loss + its gradient vs the reference formula under
torch autograd, fused AdamW + clip_grad_norm_ vs torch.optim.AdamW, checkpoint save / resume in the reference's layout."""
import numpy as np
import pytest
import torch

from craft_amd import train as T

pytestmark = pytest.mark.gpu


def _ref_sequence_loss(flow_preds, flow_gt, valid, gamma, max_flow=400.0):
    """train.py:44-73 verbatim in behaviour."""
    n = len(flow_preds)
    valid = (valid >= 0.5) & ((flow_gt ** 2).sum(dim=1).sqrt() < max_flow)
    loss = 0.0
    for i in range(n):
        loss = loss + gamma ** (n - i - 1) * (valid[:, None] * (flow_preds[i] - flow_gt).abs()).mean()
    epe = torch.sum((flow_preds[-1] - flow_gt) ** 2, dim=1).sqrt().view(-1)[valid.view(-1)]
    return loss, {"epe": epe.mean().item(), "1px": (epe < 1).float().mean().item(), "3px": (epe < 3).float().mean().item(),
                  "5px": (epe < 5).float().mean().item()}


def test_sequence_loss_and_gradient(device):
    g = torch.Generator().manual_seed(3)
    B, H, W, n = 2, 37, 45, 4
    gt = torch.randn(B, 2, H, W, generator=g) * 30
    gt[0, :, :5, :5] = 500.0                                           # beyond MAX_FLOW: excluded
    valid = (torch.rand(B, H, W, generator=g) > 0.2).float()
    preds = [(gt + torch.randn(B, 2, H, W, generator=g) * (n - i)).requires_grad_(True) for i in range(n)]
    ref_loss, ref_m = _ref_sequence_loss(preds, gt, valid, 0.8)
    ref_loss.backward()
    loss, m, grads = T.sequence_loss([p.detach().to(device) for p in preds], gt, valid, gamma=0.8, want_grad=True)
    assert float(loss) == pytest.approx(float(ref_loss.detach()), rel=1e-5)
    for k in ref_m:
        assert m[k] == pytest.approx(ref_m[k], rel=1e-5, abs=1e-7), k
    for gr, p in zip(grads, preds):
        assert torch.allclose(gr.cpu(), p.grad, rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("max_norm", [0.0, 1.0])
def test_fused_adamw_matches_torch(device, max_norm):
    torch.manual_seed(1)
    def make():
        torch.manual_seed(1)
        return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.ReLU(), torch.nn.Conv2d(8, 2, 3)).to(device)
    a, b = make(), make()
    ref = torch.optim.AdamW(a.parameters(), lr=2e-3, weight_decay=1e-2, eps=1e-8)
    ours = T.FlatAdamW(b.parameters(), lr=2e-3, weight_decay=1e-2, eps=1e-8)
    sched = T.OneCycleLR(2e-3, 50, pct_start=0.2)
    rs = torch.optim.lr_scheduler.OneCycleLR(ref, max_lr=2e-3, total_steps=50, pct_start=0.2, cycle_momentum=False,
                                             anneal_strategy="linear")
    x = torch.randn(4, 3, 12, 12, device=device)
    for it in range(12):
        ref.zero_grad(); ours.zero_grad()
        (a(x) ** 2).sum().mul(3.0).backward()
        (b(x) ** 2).sum().mul(3.0).backward()
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(a.parameters(), max_norm)
        ref.step(); rs.step()
        ours.step(lr=sched.get_last_lr()[0], max_norm=max_norm); sched.step()
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert torch.allclose(pa, pb, rtol=2e-5, atol=2e-7), f"iteration {it}"
    # a gradient summed over 2 ranks and averaged in the update == the single-rank gradient
    ours2 = T.FlatAdamW(make().parameters(), lr=2e-3, weight_decay=1e-2, eps=1e-8)
    ours3 = T.FlatAdamW(make().parameters(), lr=2e-3, weight_decay=1e-2, eps=1e-8)
    gsrc = torch.randn(ours2.numel, device=device)
    ours2.flat_grad.copy_(gsrc); ours3.flat_grad.copy_(gsrc * 2)
    ours2.step(max_norm=1.0); ours3.step(max_norm=1.0, grad_mul=0.5)
    assert torch.allclose(ours2.flat, ours3.flat, rtol=1e-6, atol=1e-8)


def test_checkpoint_save_resume(device, tmp_path):
    from craft_amd import CRAFT, default_args
    from craft_amd.synth import synth_state_dict
    m = CRAFT(default_args())
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=5), strict=True)
    m = m.to(device)
    opt, sched = T.fetch_optimizer(m, lr=1.25e-4, wdecay=1e-5, epsilon=1e-8, num_steps=200)
    assert opt.n_params == sum(p.numel() for p in m.parameters()) == 6307435          # one 25 MB buffer
    assert opt.n_params <= opt.numel < opt.n_params + 32 * len(opt.params)          # + alignment padding (< 0.1 %)
    opt.flat_grad.normal_()
    opt.step(lr=sched.get_last_lr()[0], max_norm=1.0); sched.step()
    path = str(tmp_path / "ck.pth")
    T.save_checkpoint(path, m, opt, sched, logger={"total_steps": 1})
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck.keys()) == {"model", "optimizer", "lr_scheduler", "logger"} and all(k.startswith("module.") for k in ck["model"])
    m2 = CRAFT(default_args()).to(device)
    opt2, sched2 = T.fetch_optimizer(m2, lr=1.25e-4, wdecay=1e-5, epsilon=1e-8, num_steps=200)
    _, logger = T.load_checkpoint(path, m2, opt2, sched2, load_optimizer_state=True, load_scheduler_state=True)
    assert logger == {"total_steps": 1} and sched2.last_epoch == 1 and opt2.step_count == 1
    for (k, v), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k == k2 and torch.equal(v.cpu(), v2.cpu()), k
    for i, st in opt.state_dict()["state"].items():                  # (the alignment padding of the flat buffers is not state)
        assert torch.equal(st["exp_avg"], opt2.state_dict()["state"][i]["exp_avg"])
    # the resumed model's parameters are still views of its flat buffer: a step moves them
    w0 = m2.update_block.flow_head.conv2.weight.detach().clone()
    opt2.flat_grad.fill_(1.0)
    opt2.step()
    assert not torch.equal(w0, m2.update_block.flow_head.conv2.weight.detach())


@pytest.mark.parametrize("max_norm", [0.0, 1.0])
def test_dynamic_loss_scale_skips_backs_off_and_recovers(device, max_norm):
    """torch.cuda.amp.GradScaler's contract (train.py:215, 231-238) on the device record: a non-finite gradient -- with or without
    clipping -- skips the update (weights, moments, the optimizer's step count stay), is counted, halves the scale; clean steps apply
    with the bias correction of the APPLIED count and double the scale after `growth_interval` of them."""
    def make():
        torch.manual_seed(5)
        return torch.nn.Sequential(torch.nn.Linear(7, 9), torch.nn.Tanh(), torch.nn.Linear(9, 3)).to(device)
    a, b = make(), make()
    ref = torch.optim.AdamW(a.parameters(), lr=3e-3, weight_decay=1e-2, eps=1e-8)
    ours = T.FlatAdamW(b.parameters(), lr=3e-3, weight_decay=1e-2, eps=1e-8)
    ours.set_loss_scale(1024.0, dynamic=True, growth_interval=3)
    x = torch.randn(16, 7, device=device)

    def grads(poison):
        ref.zero_grad(); ours.zero_grad()
        (a(x) ** 2).sum().backward()
        (b(x) ** 2).sum().backward(ours.scale_seed())               # scaled backward, the seed read on the device
        if poison:
            ours.flat_grad[5] = float("inf")

    def close():
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert torch.allclose(pa, pb, rtol=2e-5, atol=2e-7)

    grads(False)
    if max_norm > 0:
        torch.nn.utils.clip_grad_norm_(a.parameters(), max_norm)
    ref.step(); ours.step(max_norm=max_norm)
    close()
    s = ours.scaler_snapshot(wait=True)
    assert s == {"loss_scale": 1024.0, "applied_steps": 1, "skipped_steps": 0, "found_inf": 0}
    # overflow: nothing moves, the skip is counted, the scale halves, AdamW's step stays at 1
    before = [t.clone() for t in (ours.flat, ours.exp_avg, ours.exp_avg_sq)]
    grads(True)
    ours.step(max_norm=max_norm)
    for t0, t1 in zip(before, (ours.flat, ours.exp_avg, ours.exp_avg_sq)):
        assert torch.equal(t0, t1)
    s = ours.scaler_snapshot(wait=True)
    assert s == {"loss_scale": 512.0, "applied_steps": 1, "skipped_steps": 1, "found_inf": 1}
    assert float(ours.state_dict()["state"][0]["step"]) == 1.0 and ours.step_count == 1
    # three clean steps under the halved scale: applied like torch's steps 2..4 (bias correction did not drift), then the scale doubles
    for k in range(3):
        assert float(ours.scale_seed()) == 512.0
        grads(False)
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(a.parameters(), max_norm)
        ref.step(); ours.step(max_norm=max_norm)
        close()
    s = ours.scaler_snapshot(wait=True)
    assert s == {"loss_scale": 1024.0, "applied_steps": 4, "skipped_steps": 1, "found_inf": 0}
    assert float(ours.state_dict()["state"][0]["step"]) == float(ref.state_dict()["state"][0]["step"]) == 4.0
    # a NaN is an overflow too
    grads(False)
    ours.flat_grad[0] = float("nan")
    ours.step(max_norm=max_norm)
    assert ours.scaler_snapshot(wait=True)["skipped_steps"] == 2 and ours.step_count == 4


def test_flow_tokens_one_launch(device):
    """craft_flow_tokens (network.py:232-234, :247): flow = coords1 - coords0, its zero-padded 32-wide copy and the copy of coords1."""
    from craft_amd.hip import call
    g = torch.Generator().manual_seed(5)
    B, N = 3, 1001
    c1 = (torch.randn(B, N, 2, generator=g) * 50).to(device)
    c0 = (torch.randn(B, N, 2, generator=g) * 50).to(device)
    flow = torch.full((B, N, 2), 9.0, device=device)
    f32 = torch.full((B, N, 32), 9.0, device=device)
    cc = torch.full((B, N, 2), 9.0, device=device)
    call("craft_flow_tokens", c1, c0, B * N, flow, f32, cc)
    assert torch.equal(flow, c1 - c0) and torch.equal(cc, c1)
    assert torch.equal(f32[..., :2], c1 - c0) and float(f32[..., 2:].abs().max()) == 0.0
    flow2 = torch.empty_like(flow)
    call("craft_flow_tokens", c1, c0, B * N, flow2, None, None)          # the optional outputs
    assert torch.equal(flow2, flow)


def test_convex_upsample_backward_into_a_padded_flow_gradient(device):
    """craft_convex_upsample_bwd's flow-gradient row stride (ABI 3): accumulated straight into the flow head's 32-wide padded output
    gradient it equals the dense [B*N][2] form (atomics: compared to round-off), and the other 30 columns stay untouched."""
    from craft_amd.hip import call
    g = torch.Generator().manual_seed(6)
    B, H8, W8 = 2, 9, 13
    N = H8 * W8
    mask = torch.randn(B, N, 576, generator=g).to(device)
    flow = torch.randn(B, N, 2, generator=g).to(device)
    dup = torch.randn(B, 2, 8 * H8, 8 * W8, generator=g).to(device)
    dm2, d2 = torch.empty(B, N, 576, device=device), torch.zeros(B, N, 2, device=device)
    call("craft_convex_upsample_bwd", mask, 576, flow, dup, B, H8, W8, dm2, 576, d2, 2)
    dm32, d32 = torch.empty(B, N, 576, device=device), torch.zeros(B, N, 32, device=device)
    d32[..., 2:] = 3.0
    call("craft_convex_upsample_bwd", mask, 576, flow, dup, B, H8, W8, dm32, 576, d32, 32)
    assert torch.equal(dm2, dm32)
    assert torch.allclose(d32[..., :2], d2, rtol=1e-5, atol=1e-5) and float((d32[..., 2:] - 3.0).abs().max()) == 0.0
    # against torch autograd of the reference formula (network.py:151-162)
    fl = flow.clone().requires_grad_(True)
    mk = mask.clone().requires_grad_(True)
    m = torch.softmax(mk.view(B, H8, W8, 9, 64).permute(0, 3, 4, 1, 2).reshape(B, 1, 9, 8, 8, H8, W8), dim=2)
    uf = torch.nn.functional.unfold(8 * fl.view(B, H8, W8, 2).permute(0, 3, 1, 2), [3, 3], padding=1).view(B, 2, 9, 1, 1, H8, W8)
    up = torch.sum(m * uf, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(B, 2, 8 * H8, 8 * W8)
    up.backward(dup)
    assert torch.allclose(d2, fl.grad, rtol=1e-4, atol=1e-4) and torch.allclose(dm2, mk.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("r", [4, 3, 1])
def test_corr_lookup_backward_radii(device, r):
    """craft_corr_lookup_bwd vs torch autograd through the oracle's lookup (corr.py:47-71) on a normalised pyramid, for the reference's
    radius 4 (the kernel's compile-time form) and two run-time radii; coordinates that leave the image, accumulation over two calls."""
    from craft_amd.hip import call
    from oracle import craft_oracle as O
    g = torch.Generator().manual_seed(40 + r)
    B, H8, W8 = 2, 12, 20
    N = H8 * W8
    c = torch.randn(B, N, N, generator=g)
    pyr = [p.clone().requires_grad_(True) for p in O.build_pyramid(c, H8, W8, 4)]
    coords = O.coords_grid(B, H8, W8) + torch.randn(B, 2, H8, W8, generator=g) * 5.0
    win2 = (2 * r + 1) ** 2
    out = O.corr_lookup(pyr, coords, r)                                    # [B, 4 * win2, H8, W8]
    dout = torch.randn(B, N, 4 * win2, generator=g)
    out.backward(dout.transpose(1, 2).reshape(B, 4 * win2, H8, W8))
    G = [torch.zeros(B * N, *p.shape[-2:], device=device) for p in pyr]
    ct = coords.permute(0, 2, 3, 1).reshape(B, N, 2).contiguous().to(device)
    dd = dout.to(device)
    for _ in range(2):                                                     # (+=: two lookups at the same coordinates)
        call("craft_corr_lookup_bwd", dd, dd.stride(-2), ct, G[0], G[1], G[2], G[3], 4, B, H8, W8, r, 0, 0)
    for l in range(4):
        ref = 2.0 * pyr[l].grad.reshape(B * N, *pyr[l].shape[-2:])
        assert torch.allclose(G[l].cpu(), ref, rtol=1e-4, atol=1e-5), f"radius {r} level {l}"
