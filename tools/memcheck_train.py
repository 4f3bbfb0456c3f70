import os, sys, gc
sys.path.insert(0, "/root/repo")
import torch
from craft_amd import CRAFT, default_args
from craft_amd.synth import synth_pair, synth_state_dict
from craft_amd.train import Trainer
H, W, B, policy = 368, 496, 8, "mixed"
dev = torch.device("cuda:0")
model = CRAFT(default_args(hip_precision=policy))
model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
tr = Trainer(model.to(dev), lr=2e-4, wdecay=1e-5, num_steps=100, iters=12, clip=1.0)
im1, im2, flow = synth_pair(B, H, W, seed=500, max_flow=8)
d = (im1.to(dev), im2.to(dev), flow.to(dev), torch.ones(B, H, W, device=dev))
for i in range(8):
    m = tr.step(*d)
    torch.cuda.synchronize()
    a0 = torch.cuda.memory_allocated() / 1e9
    n = gc.collect()
    a1 = torch.cuda.memory_allocated() / 1e9
    print(f"step {i}: allocated after step {a0:.2f} GB, after gc.collect() ({n} objects) {a1:.2f} GB", flush=True)
# what is still alive: tensors by size
import collections
sz = collections.Counter()
for o in gc.get_objects():
    try:
        if torch.is_tensor(o) and o.is_cuda:
            sz[(tuple(o.shape), str(o.dtype))] += o.numel() * o.element_size()
    except Exception:
        pass
for k, v in sz.most_common(12):
    print(f"{v / 1e6:9.1f} MB  {k}")
# which objects sit in reference cycles after a step (freed only by the cyclic collector)
gc.collect()
m = tr.step(*d)
torch.cuda.synchronize()
gc.set_debug(gc.DEBUG_SAVEALL)
gc.collect()
types = collections.Counter(type(o).__name__ for o in gc.garbage)
print("cyclic garbage after one step:", types.most_common(15))
for o in gc.garbage:
    if torch.is_tensor(o) and o.is_cuda and o.numel() * o.element_size() > 20e6:
        print("  tensor in a cycle:", tuple(o.shape), o.dtype, f"{o.numel() * o.element_size() / 1e6:.0f} MB")
for o in gc.garbage:
    if type(o).__name__ not in ("Tensor", "dict", "list", "tuple", "cell", "function", "Parameter"):
        print("  object:", type(o), [type(r).__name__ for r in gc.get_referents(o)][:12])
gc.set_debug(0)
