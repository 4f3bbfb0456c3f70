"""Where a training step spends its time (diagnostics): encoder forward / hot-path forward / backward, by HIP events."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from craft_amd import CRAFT, default_args
from craft_amd import autograd as AG
from craft_amd.synth import synth_pair, synth_state_dict
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if len(sys.argv) > 2 and sys.argv[2] == "bench":
    torch.backends.cudnn.benchmark = True
H, W, B, policy = {3: (368, 496, 8, "train_f16x3"), 4: (368, 768, 4, "train_bf16attn")}[cfg]
dev = torch.device("cuda")
model = CRAFT(default_args(hip_precision=policy))
model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
model = model.to(dev).train()
im1, im2, flow = synth_pair(B, H, W, seed=100)
im1, im2, flow = im1.to(dev), im2.to(dev), flow.to(dev)
valid = torch.ones(B, H, W, device=dev)

def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e

for it in range(4):
    for p in model.parameters():
        p.grad = None
    t0 = ev()
    a = 2 * (im1 / 255.0) - 1; b = 2 * (im2 / 255.0) - 1
    f1, f2 = model.fnet([a, b]); cn = model.cnet(a)
    t1 = ev()
    g1, g2, g3 = torch.randn_like(f1), torch.randn_like(f2), torch.randn_like(cn)
    torch.autograd.backward([f1, f2, cn], [g1, g2, g3])
    t2 = ev()
    for p in model.parameters():
        p.grad = None
    t3 = ev()
    preds = model(im1, im2, iters=12)
    t4 = ev()
    loss, _ = AG.sequence_loss(preds, flow, valid, 0.8)
    loss.backward()
    t5 = ev()
    torch.cuda.synchronize()
    print(f"iter {it}: encoders fwd {t0.elapsed_time(t1):8.2f} ms  encoders bwd {t1.elapsed_time(t2):8.2f} ms | full fwd {t3.elapsed_time(t4):8.2f} ms  "
          f"loss+full bwd {t4.elapsed_time(t5):8.2f} ms", flush=True)
