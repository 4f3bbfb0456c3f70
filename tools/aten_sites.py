"""Every aten op a Trainer.step dispatches that launches a device kernel, by op, shape and the craft_amd source line that issued it
(TorchDispatchMode + the Python stack; ops issued by the autograd engine itself -- gradient accumulation -- show as site 'engine').
usage: python tools/aten_sites.py [3|4]"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
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
for _ in range(3):
    tr.step(im1, im2, flow, valid)
torch.cuda.synchronize()
NOKERNEL = ("view", "as_strided", "expand", "slice", "select", "permute", "transpose", "unsqueeze", "squeeze", "detach", "alias", "t.", "reshape",
            "empty", "_unsafe_view", "unbind", "split", "_local_scalar_dense", "is_", "size", "stride", "numel", "set_", "resize", "_reshape_alias",
            "lift_fresh", "narrow", "chunk", "unfold", "_to_copy.default_nocopy")
agg = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        out = func(*args, **(kwargs or {}))
        short = name.replace("aten.", "")
        if not any(short.startswith(n) for n in NOKERNEL):
            tens = [a for a in list(args) + ([out] if isinstance(out, torch.Tensor) else []) if isinstance(a, torch.Tensor)]
            if any(t.is_cuda for t in tens):
                site = "engine"
                for fr in reversed(traceback.extract_stack()):
                    if "craft_amd" in fr.filename and "python_dispatch" not in fr.filename:
                        site = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                        break
                shp = "x".join(str(s) for s in tens[0].shape) if tens else ""
                agg[(short, shp, site)] += 1
        return out


with Log():
    tr.step(im1, im2, flow, valid)
torch.cuda.synchronize()
print(f"aten ops with device tensors in one step: {sum(agg.values())}")
for (op, shp, site), n in agg.most_common(90):
    print(f"{n:5d}  {op:32s} {shp:24s} {site}")
