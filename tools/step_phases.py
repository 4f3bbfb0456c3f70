"""Where a Trainer.step spends its wall time: device time (HIP events) and host time per phase (diagnostics).
usage: python tools/step_phases.py [3|4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from craft_amd import CRAFT, default_args
from craft_amd.autograd import sequence_loss as seq_loss
from craft_amd.synth import synth_pair, synth_state_dict
from craft_amd.train import Trainer, auto_loss_scale

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H, W, B, policy = {3: (368, 496, 8, "mixed"), 4: (368, 768, 4, "train_bf16attn")}[cfg]
dev = torch.device("cuda:0")
model = CRAFT(default_args(hip_precision=policy))
model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
model = model.to(dev)
tr = Trainer(model, lr=4e-4, wdecay=1e-4, num_steps=100000, iters=12, clip=1.0, freeze_bn=cfg != 3)
im1, im2, flow = synth_pair(B, H, W, seed=100)
im1, im2, flow = im1.to(dev), im2.to(dev), flow.to(dev)
valid = torch.ones(B, H, W, device=dev)
names = ["zero_grad", "forward", "loss+metrics", "backward", "optimizer", "readback"]
for it in range(6):
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
    ht = [time.perf_counter()]
    ev[0].record()
    tr.optimizer.zero_grad(); ev[1].record(); ht.append(time.perf_counter())
    preds = model(im1, im2, iters=12); ev[2].record(); ht.append(time.perf_counter())
    loss, metrics = seq_loss(preds, flow, valid, 0.8); ev[3].record(); ht.append(time.perf_counter())
    ls = auto_loss_scale(flow.numel())
    loss.backward(torch.full((), ls, device=dev)); ev[4].record(); ht.append(time.perf_counter())
    tr.optimizer.step(lr=tr.scheduler.get_last_lr()[0], max_norm=1.0, grad_mul=1.0 / ls); tr.scheduler.step(); ev[5].record(); ht.append(time.perf_counter())
    x = float(loss.detach()); ev[6].record(); ht.append(time.perf_counter())
    torch.cuda.synchronize()
    tot = time.perf_counter() - ht[0]
    if it >= 2:
        print(f"step {it}: wall {1e3 * tot:6.2f} ms | " + " | ".join(f"{n} host {1e3 * (ht[i + 1] - ht[i]):5.2f} dev {ev[i].elapsed_time(ev[i + 1]):5.2f}" for i, n in enumerate(names)), flush=True)
# the same step through Trainer.step, back to back (what bench.py times)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(5):
        m_ = tr.step(im1, im2, flow, valid)
    torch.cuda.synchronize()
    print(f"Trainer.step x5: {1e3 * (time.perf_counter() - t0) / 5:6.2f} ms per step", flush=True)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.step(im1, im2, flow, valid)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
