"""The training step as a whole on the GPU: Trainer.step = forward (model.train()) -> sequence loss ->
HIP backward -> ONE all-reduce of the flat gradient -> clip -> fused AdamW -> OneCycle.

* weights really move and the inference path sees them (packed-weight caches are invalidated: ADVICE r1);
* parameters the reference never touches (find_unused_parameters) are not decayed and carry no optimizer state;
* against torch.optim.AdamW + clip_grad_norm_ fed with the same gradients;
* data parallel: two ranks (gloo, sharing the one GPU of the box; the driver's 8-GPU runs use RCCL) with one pair each end
  up with exactly the parameters of one process that trains on both pairs."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest
import torch

from craft_amd import CRAFT, default_args
from craft_amd.synth import synth_pair, synth_state_dict
from craft_amd.train import Trainer, unused_parameters

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(device, seed=1234, **over):
    model = CRAFT(default_args(hip_precision="fp32", dropout_prob=0.0, **over))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=seed), strict=True)
    return model.to(device)


def _batch(B, H, W, seed):
    im1, im2, flow = synth_pair(B, H, W, seed=seed)
    g = torch.Generator().manual_seed(seed)
    valid = (torch.rand(B, H, W, generator=g) > 0.1).float()
    return im1, im2, flow, valid


def test_step_load_step_uses_the_loaded_weights(device, tmp_path):
    """the trainer's batched weight-pack registry (ops.WeightPackRegistry: every conv-weight operand of a step re-packed in one
    launch after the optimizer update) must not hand out packs of the OLD weights after a checkpoint load / load_state_dict between two
    steps.  step -> load other weights -> step must produce the loss of a FRESH trainer built on those weights, in a policy that packs
    (f16x3) -- with the stale packs the second loss is the old model's."""
    from craft_amd.train import load_checkpoint, save_checkpoint
    im1, im2, flow, valid = _batch(2, 128, 160, 5)

    def mk(seed):
        m = CRAFT(default_args(hip_precision="train_f16x3", dropout_prob=0.0))
        m.load_state_dict(synth_state_dict(m.state_dict(), seed=seed), strict=True)
        return m.to(device)
    other = mk(4321)
    tr_o = Trainer(other, lr=1e-4, num_steps=50, iters=2, clip=1.0)
    ck = str(tmp_path / "other.pth")
    save_checkpoint(ck, other, tr_o.optimizer, tr_o.scheduler)
    expect = tr_o.step(im1, im2, flow, valid)["loss"]                 # first step of a fresh trainer on the OTHER weights
    model = mk(1234)
    tr = Trainer(model, lr=1e-4, num_steps=50, iters=2, clip=1.0)
    l0 = tr.step(im1, im2, flow, valid)["loss"]
    tr.step(im1, im2, flow, valid)                                    # (the registry is fresh from here on: packs of THESE weights)
    assert abs(l0 - expect) > 1e-3 * abs(expect), "the two weight sets must give different losses for this test to mean anything"
    load_checkpoint(ck, model, tr.optimizer)                          # in-place copy into the flat buffer's views
    got = tr.step(im1, im2, flow, valid)["loss"]
    assert got == pytest.approx(expect, rel=2e-5), (got, expect, l0)
    # ... and the same through a bare load_state_dict (no helper that could bump an epoch)
    model.load_state_dict({k: v.detach().clone() for k, v in mk(1234).state_dict().items()}, strict=True)
    assert tr.step(im1, im2, flow, valid)["loss"] == pytest.approx(l0, rel=2e-5)


def test_steps_move_weights_and_inference_sees_them(device):
    model = _model(device)
    tr = Trainer(model, lr=2e-4, num_steps=50, iters=3, clip=1.0)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    im1, im2, flow, valid = _batch(2, 128, 192, 3)
    model.eval()
    with torch.no_grad():
        up_before = model(im1.to(device), im2.to(device), iters=3, test_mode=1)[1].clone()     # packs weights into the caches
    losses = [tr.step(im1, im2, flow, valid)["loss"] for _ in range(4)]
    assert all(l == l and l < 1e4 for l in losses), losses
    assert losses[-1] < losses[0], f"four steps on one batch should reduce its loss: {losses}"
    moved = [k for k, v in model.state_dict().items() if v.dtype.is_floating_point and not torch.equal(v, before[k])]
    assert len(moved) >= 150                      # (conv biases in front of a statistics-normalised layer have an exactly zero gradient)
    # inference after training: same result as a fresh model that loads the trained weights (no stale packed copies)
    model.eval()
    fresh = CRAFT(default_args(hip_precision="fp32"))
    fresh.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()}, strict=True)
    fresh = fresh.to(device).eval()
    with torch.no_grad():
        a = model(im1.to(device), im2.to(device), iters=3, test_mode=1)[1]
        b = fresh(im1.to(device), im2.to(device), iters=3, test_mode=1)[1]
    assert (a - b).abs().max().item() < 1e-4
    assert (a - up_before).abs().max().item() > 1e-3, "training did not change the prediction"


def test_trainer_overflow_is_skipped_counted_and_recovered(device):
    """Trainer(loss_scale="auto") = GradScaler (train.py:215, 231-238): an overflowing backward skips the update, shows up in the
    metrics, halves the scale, leaves the optimizer's step count alone; the next step applies."""
    model = _model(device)
    tr = Trainer(model, lr=2e-4, num_steps=50, iters=2, clip=1.0, freeze_bn=True)
    im1, im2, flow, valid = _batch(2, 128, 160, 4)
    m = tr.step(im1, im2, flow, valid)
    assert m["skipped_steps"] == 0 and m["applied_steps"] == 1 and m["loss_scale"] == tr.last_loss_scale > 1.0
    good = m["loss_scale"]
    w = {k: v.clone() for k, v in model.state_dict().items()}
    tr.optimizer._scaler_f[0] = 2.0 ** 126          # the scaled loss gradient (~1e-5 x 2^126) overflows fp32 within the backward
    m = tr.step(im1, im2, flow, valid)
    assert m["skipped_steps"] == 1 and m["applied_steps"] == 1 and m["loss_scale"] == 2.0 ** 125 and m["loss"] == m["loss"]
    assert all(torch.equal(v, w[k]) for k, v in model.state_dict().items() if v.dtype.is_floating_point), "a skipped step moved weights"
    assert float(tr.optimizer.state_dict()["state"][0]["step"]) == 1.0
    assert tr.scheduler.last_epoch == 2             # the LR schedule advances on a skipped step, as scheduler.step() does in train.py:236
    tr.optimizer._scaler_f[0] = good
    m = tr.step(im1, im2, flow, valid)
    assert m["skipped_steps"] == 1 and m["applied_steps"] == 2 and m["loss_scale"] == good
    assert any(not torch.equal(v, w[k]) for k, v in model.state_dict().items() if v.dtype.is_floating_point)


def test_unused_parameters_are_skipped_like_torch_adamw(device):
    model = _model(device)
    un = unused_parameters(model)
    assert len(un) == 2
    snap = [p.detach().clone() for p in un]
    tr = Trainer(model, lr=1e-3, wdecay=0.1, num_steps=20, iters=2)
    im1, im2, flow, valid = _batch(1, 128, 128, 5)
    tr.step(im1, im2, flow, valid)
    tr.step(im1, im2, flow, valid)
    for p, s in zip(un, snap):
        assert torch.equal(p.detach(), s), "a parameter without gradient was decayed"
    sd = tr.optimizer.state_dict()
    idx = {i for i, p in enumerate(model.parameters()) if any(p is u for u in un)}
    assert idx and not (idx & set(sd["state"].keys())) and len(sd["state"]) == len(list(model.parameters())) - len(idx)


def test_step_matches_torch_adamw_on_the_same_gradients(device):
    """One Trainer.step vs torch.optim.AdamW + clip_grad_norm_ applied to a copy of the model with the gradients our backward
    produced (isolates clip + optimizer + scheduler from the backward, which test_train_backward.py pins to the reference)."""
    model = _model(device)
    ref = _model(device)
    tr = Trainer(model, lr=3e-4, wdecay=1e-4, epsilon=1e-8, num_steps=100, iters=2, clip=0.5)
    im1, im2, flow, valid = _batch(2, 128, 160, 7)
    lr0 = tr.scheduler.get_last_lr()[0]
    tr.step(im1, im2, flow, valid)
    un = {id(p) for p in unused_parameters(ref)}
    params = [p for p in ref.parameters() if id(p) not in un]
    opt = torch.optim.AdamW(params, lr=lr0, weight_decay=1e-4, eps=1e-8)
    for p, g in zip(ref.parameters(), [q.grad for q in model.parameters()]):
        p.grad = g.detach().clone()
    torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
    opt.step()
    for (k, a), b in zip(model.named_parameters(), ref.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), k


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from craft_amd import CRAFT, default_args
    from craft_amd.synth import synth_pair, synth_state_dict
    from craft_amd.train import Trainer
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    backend = os.environ.get("CRAFT_TEST_BACKEND", "gloo")
    if world > 1:
        dist.init_process_group(backend, init_method="env://")
    # gloo: both ranks share GPU 0 (1-GPU box); nccl (= RCCL): one rank per GPU, the gradient all-reduce runs on the devices
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)) if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    model = CRAFT(default_args(hip_precision="fp32", dropout_prob=0.0))
    # rank 1 starts from DIFFERENT weights and BatchNorm statistics: Trainer must bring rank 0's over (DDP's construction broadcast)
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234 + 77 * rank), strict=True)
    model = model.to(dev)
    tr = Trainer(model, lr=2e-4, num_steps=50, iters=2, clip=1.0, freeze_bn=True)
    if world > 1:
        ref = synth_state_dict(model.state_dict(), seed=1234)
        for k, v in model.state_dict().items():
            assert torch.equal(v.cpu(), ref[k].to(v.dtype)), "rank %%d: %%s is not rank 0's after Trainer()" %% (rank, k)
    im1, im2, flow = synth_pair(2, 128, 160, seed=9)
    valid = torch.ones(2, 128, 160)
    sl = slice(rank, rank + 1) if world > 1 else slice(0, 2)
    m = tr.step(im1[sl], im2[sl], flow[sl], valid[sl])
    grad = (tr.optimizer.flat_grad / world / tr.last_loss_scale).cpu()   # the all-reduced (summed) gradient, averaged and un-scaled
    m = tr.step(im1[sl], im2[sl], flow[sl], valid[sl])
    if rank == 0:
        torch.save({"sd": {k: v.cpu() for k, v in model.state_dict().items()}, "loss": m["loss"], "grad": grad}, sys.argv[1])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
""")


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_rank_data_parallel_equals_one_process_on_both_pairs(device, tmp_path, backend):
    """Two data-parallel ranks (one pair each) against one process that trains on both pairs.  "gloo": the ranks share the box's
    GPU and the flat gradient is staged through the host.  "nccl": RCCL, one rank per GPU, the all-reduce of train_ddp.py:187-200 on
    the devices -- needs two GPUs, skipped on a 1-GPU box."""
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL data-parallel test needs >= 2 GPUs")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="2", CRAFT_TEST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", port, str(script), str(tmp_path / "dp.pt")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([sys.executable, str(script), str(tmp_path / "single.pt")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    dp, single = torch.load(tmp_path / "dp.pt"), torch.load(tmp_path / "single.pt")
    assert abs(dp["loss"] - single["loss"]) < 1e-4 * abs(single["loss"])
    rel = ((dp["grad"] - single["grad"]).norm() / single["grad"].norm()).item()
    assert rel < 2e-3, f"averaged data-parallel gradient differs from the two-pair gradient: relative L2 {rel:.2e}"
    worst = 0.0
    for k, v in single["sd"].items():
        if v.dtype.is_floating_point:
            worst = max(worst, (dp["sd"][k] - v).abs().max().item())
    # (AdamW's first steps move every weight by ~lr regardless of the gradient's size, so rounding-level gradient differences
    # between the two reduction orders can flip tiny updates: bound = a fraction of lr)
    assert worst < 2e-4, f"data-parallel and single-process parameters differ by {worst:.2e}"


def test_trainer_gma_variant_skips_unused_position_embeddings(device):
    """train-gma.sh's model (plain correlation + GMA attention): two steps run, the loss is finite, and the relative-position
    embeddings that content-only attention never reads (no gradient in the reference: tests/golden/train_plaingma_*.npz 'unused')
    are left untouched by the optimizer -- no weight decay on them either, like torch.optim.AdamW with grad None."""
    from craft_amd import CRAFT, default_args
    from craft_amd.synth import synth_state_dict
    model = CRAFT(default_args(hip_precision="fp32", craft=False, use_setrans=False))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=5), strict=True)
    model = model.to(device)
    tr = Trainer(model, lr=2e-4, wdecay=1e-2, num_steps=50, iters=2, clip=1.0)
    before = {k: v.clone() for k, v in model.state_dict().items() if k.startswith("att.pos_emb.")}
    assert len(before) >= 2                      # the two embedding tables (+ the rel_ind index buffer)
    im1, im2, flow, valid = _batch(2, 128, 160, 7)
    losses = [tr.step(im1, im2, flow, valid)["loss"] for _ in range(2)]
    assert all(l == l and l < 1e4 for l in losses), losses
    for k, v in before.items():
        assert torch.equal(model.state_dict()[k], v), k


def test_three_steps_follow_the_reference_training_loop(device):
    """Trainer.step x 3 against THREE iterations of the reference's own loop (train.py:215-236 on the imported model, optimizer and scheduler
    from train.py's fetch_optimizer, dropout ON with the masks of every pass handed in as data: tools/make_golden_train_traj.py): the
    losses of the three (different) batches, the learning rates, every parameter's change after the third update, BatchNorm statistics."""
    import json

    import numpy as np
    from golden_util import GOLDEN_DIR, sample_idx
    z = np.load(os.path.join(GOLDEN_DIR, "train_traj_b2_128x160_T2.npz"))
    zg = np.load(os.path.join(GOLDEN_DIR, "train_dropout_b2_128x160_T2.npz"))          # (its gradients tell which parameters have none)
    c = json.loads(str(z["meta"]))
    model = CRAFT(default_args(hip_precision="fp32"))                                   # dropout at the config's 0.1 / 0.2
    sd0 = synth_state_dict(model.state_dict(), seed=c["seed"], qk_gain=c["qk_gain"])
    model.load_state_dict(sd0, strict=True)
    model = model.to(device)
    tr = Trainer(model, lr=c["lr"], wdecay=c["wdecay"], epsilon=c["epsilon"], num_steps=c["num_steps"], clip=c["clip"], gamma=c["gamma"],
                 iters=c["iters"], loss_scale=None)
    torch.manual_seed(c["torch_seed"])                     # the dropout seeds of pass s: pass_base(torch_seed, s) (craft_amd/train_forward.py)
    model.__dict__["_train_calls"] = 0
    losses, lrs = [], []
    for s in range(c["steps"]):
        lrs.append(tr.scheduler.get_last_lr()[0] if hasattr(tr.scheduler, "get_last_lr") else None)
        m = tr.step(torch.from_numpy(z[f"image1.{s}"].astype(np.float32)).to(device), torch.from_numpy(z[f"image2.{s}"].astype(np.float32)).to(device),
                    torch.from_numpy(z[f"flow_gt.{s}"]).to(device), torch.from_numpy(z[f"valid.{s}"]).to(device))
        losses.append(float(m["loss"]))
    torch.cuda.synchronize()
    # losses: the first one is the dropout-on forward alone; the second and third have one / two updates behind them
    assert losses[0] == pytest.approx(z["losses"][0], rel=3e-5)
    # (measured: 2e-6 and 5e-6 relative)
    assert losses[1] == pytest.approx(z["losses"][1], rel=1e-4) and losses[2] == pytest.approx(z["losses"][2], rel=1e-4), (losses, z["losses"])
    if lrs[0] is not None:
        assert lrs == pytest.approx(z["lrs"].tolist(), rel=1e-6)
    # which parameters carry a real gradient (the others -- biases in front of a normalisation layer -- receive rounding noise, which AdamW's
    # first steps turn into +-lr whatever its size: not comparable, and without effect on the loss)
    r = [np.sqrt(zg[k][1] / max(1, int(np.prod(zg[k[:-2] + ".shape"])))) for k in zg.files if k.startswith("grad.") and k.endswith(".s")]
    scale = float(np.median(r))
    unused = set(json.loads(str(zg["unused"])))
    worst, worst_k, checked, seen = 0.0, None, 0, set()
    l2s = []
    for k, p in model.named_parameters():
        if id(p) in seen or k.startswith("corr_fn.setrans.key."):
            continue
        seen.add(id(p))
        dw = (p.detach().cpu() - sd0[k]).reshape(-1).numpy()
        if k in unused:
            assert float(np.abs(dw).max()) == 0.0, f"{k}: unused in the reference (no gradient, no decay)"
            continue
        gs = zg[f"grad.{k}.s"]
        if np.sqrt(gs[1] / dw.size) < 1e-4 * scale:
            continue
        ref = z[f"dw.{k}.v"]
        got = dw[sample_idx(dw.size)]
        l2 = float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))
        if l2 > worst:
            worst, worst_k = l2, k
        checked += 1
        l2s.append((l2, k, dw.size))
        # AdamW's first updates are -lr * m / (sqrt(v) + eps) ~ -lr * sign(g): an element whose gradient is near zero flips on rounding noise
        # (a flipped fraction f costs 2 sqrt(f) of relative L2: the 4e-2 of the large fnet weights is 0.04 % of their elements).  Measured on the
        # MI355X: median 1.2e-2, worst 7.3e-2 (a 96-element BatchNorm bias); bound = 2 x the worst.  The losses above are the sharp check.
        assert l2 < 0.15, f"{k}: change after three updates differs by {l2:.2e} (relative L2 over the sample)"
    assert checked >= 100
    for k in [f for f in z.files if f.startswith("bn.")]:
        assert np.allclose(model.state_dict()[k[3:]].cpu().numpy(), z[k], rtol=1e-3, atol=1e-5), k
    print("[trajectory] largest:", [(f"{a:.2e}", b, n) for a, b, n in sorted(l2s, reverse=True)[:8]], "median", f"{sorted(l2s)[len(l2s) // 2][0]:.2e}")
    print(f"[trajectory] losses {['%.6f' % v for v in losses]} vs reference {['%.6f' % v for v in z['losses']]}; worst relative L2 of a parameter's "
          f"three-step change {worst:.2e} ({worst_k}), {checked} parameters")
