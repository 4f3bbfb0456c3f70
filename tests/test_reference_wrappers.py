"""The reference's own loop body wrapped around craft_amd.CRAFT (the class must be wrappable).

train.py:179-183, 215-236:   nn.DataParallel(model, device_ids=...), torch.optim.AdamW(model.parameters()), OneCycleLR,
                             GradScaler(enabled=True), scaler.scale(loss).backward(), scaler.unscale_, clip_grad_norm_, scaler.step
train_ddp.py:196-200:        DistributedDataParallel(model, device_ids=[local_rank], find_unused_parameters=True)

* the DataParallel + GradScaler + torch AdamW loop produces the gradients of craft_amd's native Trainer on the same batch and
  moves the weights like it;
* under DDP (gloo, two ranks sharing the box's GPU) every parameter's gradient hook fires exactly once per backward although the
  layers that run in all iterations hand their accumulated gradient to autograd only on the last use (autograd.py), the unused
  parameters are found, and the averaged gradient equals the one-process gradient over both pairs."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest
import torch

from craft_amd import CRAFT, default_args
from craft_amd.synth import synth_pair, synth_state_dict
from craft_amd.train import Trainer

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_sequence_loss(flow_preds, flow_gt, valid, gamma=0.8, max_flow=400.0):
    """train.py:44-61 in behaviour, on torch ops (what the reference's loop calls on the model's output list)."""
    n = len(flow_preds)
    mag = torch.sum(flow_gt ** 2, dim=1).sqrt()
    valid = (valid >= 0.5) & (mag < max_flow)
    loss = 0.0
    for i in range(n):
        loss = loss + gamma ** (n - i - 1) * (valid[:, None] * (flow_preds[i] - flow_gt).abs()).mean()
    return loss


def _model(device, seed=1234):
    model = CRAFT(default_args(hip_precision="fp32", dropout_prob=0.0))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=seed), strict=True)
    return model.to(device)


def test_dataparallel_gradscaler_adamw_loop_matches_trainer(device):
    im1, im2, flow = synth_pair(2, 128, 160, seed=11)
    valid = torch.ones(2, 128, 160)
    iters, lr, wd, clip = 3, 2e-4, 1e-4, 1.0
    # ---- the native step
    ours = _model(device)
    tr = Trainer(ours, lr=lr, wdecay=wd, num_steps=50, iters=iters, clip=clip, freeze_bn=True)
    m = tr.step(im1, im2, flow, valid)
    want_grad = {n: (p.grad / tr.last_loss_scale).clone() for n, p in ours.named_parameters()}
    want_w = {n: p.detach().clone() for n, p in ours.named_parameters()}
    # ---- the reference's loop body around the same class (train.py:179-183, 215-236)
    model = torch.nn.DataParallel(_model(device), device_ids=[0])
    model.train()
    model.module.freeze_bn()
    optimizer = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=wd, eps=1e-8)
    scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer, lr, 50 + 100, pct_start=0.05, cycle_momentum=False, anneal_strategy="linear")
    scaler = torch.amp.GradScaler("cuda", enabled=True)
    optimizer.zero_grad()
    preds = model(im1.to(device), im2.to(device), iters=iters)
    assert isinstance(preds, list) and len(preds) == iters and preds[0].shape == (2, 2, 128, 160) and preds[0].requires_grad
    loss = reference_sequence_loss(preds, flow.to(device), valid.to(device))
    assert float(loss) == pytest.approx(m["loss"], rel=1e-5)
    scaler.scale(loss).backward()
    scaler.unscale_(optimizer)
    got_grad = {n: (None if p.grad is None else p.grad.clone()) for n, p in model.module.named_parameters()}
    torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
    scaler.step(optimizer)
    scheduler.step()
    scaler.update()
    unused = {n for n, g in got_grad.items() if g is None}
    assert unused == {"att.setrans.attn_softaggr.feat2score.weight", "att.setrans.attn_softaggr.feat2score.bias"} or \
        all("attn_softaggr" in n and n.startswith("att.") for n in unused), unused
    num = sum(((got_grad[n] - want_grad[n]) ** 2).sum().item() for n in got_grad if got_grad[n] is not None)
    den = sum((want_grad[n] ** 2).sum().item() for n in got_grad if got_grad[n] is not None)
    assert (num / den) ** 0.5 < 1e-4, f"DataParallel + GradScaler gradients differ from the Trainer's: relative L2 {(num / den) ** 0.5:.2e}"
    for n, g in got_grad.items():
        if g is not None and want_grad[n].abs().max() > 0:
            rel = ((g - want_grad[n]).norm() / want_grad[n].norm()).item()
            assert rel < 5e-3, (n, rel)
    worst = max((p.detach() - want_w[n]).abs().max().item() for n, p in model.module.named_parameters())
    assert worst < 2e-5, f"torch AdamW under GradScaler and the fused step moved the weights differently: {worst:.2e}"
    # the state dict of the wrapped model carries the reference's 'module.' prefix and loads back into a bare model
    sd = model.state_dict()
    assert all(k.startswith("module.") for k in sd)
    bare = CRAFT(default_args(hip_precision="fp32"))
    bare.load_state_dict({k[len("module."):]: v.cpu() for k, v in sd.items()}, strict=True)


DDP_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from craft_amd import CRAFT, default_args
    from craft_amd.synth import synth_pair, synth_state_dict
    sys.path.insert(0, os.path.join(%r, "tests"))
    from test_reference_wrappers import reference_sequence_loss
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    if world > 1:
        dist.init_process_group("gloo", init_method="env://")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model = CRAFT(default_args(hip_precision="fp32", dropout_prob=0.0))
    # rank 1 starts from other weights: DDP's construction broadcast must bring rank 0's over
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234 + 5 * rank), strict=True)
    model = model.to(dev)
    model.train(); model.freeze_bn()
    net = DDP(model, device_ids=[0], find_unused_parameters=True) if world > 1 else model      # train_ddp.py:196-200
    fired = {}
    for n, p in model.named_parameters():
        p.register_hook(lambda g, n=n: fired.__setitem__(n, fired.get(n, 0) + 1))
    im1, im2, flow = synth_pair(2, 128, 160, seed=21)
    valid = torch.ones(2, 128, 160)
    sl = slice(rank, rank + 1) if world > 1 else slice(0, 2)
    out = {}
    for step in range(2):                         # two backward passes: the accumulators must re-arm, the hooks fire once per pass
        fired.clear()
        net.zero_grad()
        preds = net(im1[sl].to(dev), im2[sl].to(dev), iters=3)
        loss = reference_sequence_loss(preds, flow[sl].to(dev), valid[sl].to(dev))
        loss.backward()
        multi = {n: c for n, c in fired.items() if c != 1}
        assert not multi, "gradient hooks fired more than once: %%r" %% multi
        out[step] = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
        unused = [n for n, p in model.named_parameters() if p.grad is None or n not in fired]
        assert all("att.setrans.attn_softaggr" in n for n in unused), unused
    if rank == 0:
        torch.save({"grads": out, "loss": float(loss)}, sys.argv[1])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
""")


def test_ddp_find_unused_parameters_two_ranks_gloo(device, tmp_path):
    script = tmp_path / "ddp_worker.py"
    script.write_text(DDP_WORKER % (ROOT, ROOT))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", port, str(script), str(tmp_path / "ddp.pt")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([sys.executable, str(script), str(tmp_path / "single.pt")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    ddp, single = torch.load(tmp_path / "ddp.pt"), torch.load(tmp_path / "single.pt")
    for step in (0, 1):
        a, b = ddp["grads"][step], single["grads"][step]
        assert set(a) == set(b)
        num = sum(((a[n] - b[n]) ** 2).sum().item() for n in a)
        den = sum((b[n] ** 2).sum().item() for n in a)
        # DDP averages the two ranks' one-pair gradients = the gradient of the two-pair mean loss
        assert (num / den) ** 0.5 < 2e-3, f"step {step}: DDP-averaged gradient differs from the two-pair gradient: {(num / den) ** 0.5:.2e}"
