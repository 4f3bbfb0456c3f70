"""Which PyTorch (aten) kernels a Trainer.step still launches between the craft_* calls, by op, shape and Python call site (diagnostics).
usage: python tools/aten_profile.py [3|4]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from craft_amd import CRAFT, default_args
from craft_amd.synth import synth_pair, synth_state_dict
from craft_amd.train import Trainer

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H, W, B, policy = {3: (368, 496, 8, "mixed"), 4: (368, 768, 4, "train_bf16attn")}[cfg]
dev = torch.device("cuda:0")
model = CRAFT(default_args(hip_precision=policy))
model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
model = model.to(dev)
tr = Trainer(model, lr=4e-4, wdecay=1e-4, num_steps=100000, iters=12, clip=1.0, freeze_bn=cfg != 3)
im1, im2, flow = [t.to(dev) for t in synth_pair(B, H, W, seed=100)]
valid = torch.ones(B, H, W, device=dev)
for _ in range(5):
    tr.step(im1, im2, flow, valid)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(im1, im2, flow, valid)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time <= 0:
        continue
    site = "?"
    for fr in ev.stack or []:
        if "craft_amd" in fr and "autograd/function" not in fr:
            site = fr.split("/")[-1][:70]
            break
    key = (ev.name, str(ev.input_shapes)[:60], site)
    agg[key][0] += ev.device_time
    agg[key][1] += 1
tot = sum(v[0] for v in agg.values())
print(f"aten kernels with device time: {tot / 1e3:.2f} ms per step")
for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"{t / 1e3:7.3f} ms {n:4d}x  {k[0]:28s} {k[1]:60s} {k[2]}")
