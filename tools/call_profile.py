"""Per-C-ABI-call device time of one Trainer.step, by entry point and integer arguments (diagnostics): every craft_* call is bracketed
by HIP events on the stream it runs on; the time between two calls (torch's own kernels: fills, adds, cats, copies) is booked as
"(torch) before <entry>".  usage: python tools/call_profile.py [3|4] [top]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import craft_amd
from craft_amd import CRAFT, default_args, hip
from craft_amd.synth import synth_pair, synth_state_dict
from craft_amd.train import Trainer

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
H, W, B, policy = {3: (368, 496, 8, "mixed"), 4: (368, 768, 4, "train_bf16attn")}[cfg]
dev = torch.device("cuda:0")
model = CRAFT(default_args(hip_precision=policy))
model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
model = model.to(dev)
tr = Trainer(model, lr=4e-4, wdecay=1e-4, num_steps=100000, iters=12, clip=1.0, freeze_bn=cfg != 3)
im1, im2, flow = [t.to(dev) for t in synth_pair(B, H, W, seed=100)]
valid = torch.ones(B, H, W, device=dev)
for _ in range(6):
    tr.step(im1, im2, flow, valid)
torch.cuda.synchronize()

real = hip.call
log = []


def timed(name, *args):
    key = (name,) + tuple(a for a in args if isinstance(a, int) and not isinstance(a, bool) and abs(a) < (1 << 40))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    real(name, *args)
    e1.record()
    log.append((key, e0, e1))


mods = [m for n, m in sys.modules.items() if n.startswith("craft_amd") and getattr(m, "call", None) is real]
for m in mods:
    m.call = timed
STEPS = 3
start = torch.cuda.Event(enable_timing=True); start.record()
for _ in range(STEPS):
    tr.step(im1, im2, flow, valid)
end = torch.cuda.Event(enable_timing=True); end.record()
torch.cuda.synchronize()
for m in mods:
    m.call = real
tot = collections.defaultdict(lambda: [0.0, 0])
prev = start
for key, e0, e1 in log:
    g = prev.elapsed_time(e0)
    t = tot[("(torch) before " + key[0],)]; t[0] += g; t[1] += 1
    t = tot[key]; t[0] += e0.elapsed_time(e1); t[1] += 1
    prev = e1
wall = start.elapsed_time(end) / STEPS
print(f"config {cfg}: {wall:.2f} ms per step with events ({len(log) // STEPS} craft_* calls per step)")
by_name = collections.defaultdict(float)
for k, (t, n) in tot.items():
    by_name[k[0]] += t / STEPS
print("-- by entry point (ms per step)")
for k, t in sorted(by_name.items(), key=lambda kv: -kv[1])[:top]:
    print(f"{t:8.3f}  {k}")
print("-- by entry point and integer arguments (ms per step, calls per step, us per call)")
for k, (t, n) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{t / STEPS:8.3f} {n / STEPS:6.1f} {1e3 * t / n:8.1f}  {k[0]} {k[1:]}")
