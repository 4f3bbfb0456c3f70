import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from craft_amd import CRAFT, default_args
from craft_amd.synth import synth_pair, synth_state_dict
dev = torch.device("cuda")
model = CRAFT(default_args())
model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
model = model.to(dev).train()
B, H, W = 8, 368, 496
im1, im2, _ = synth_pair(B, H, W, seed=100)
a = (2 * (im1 / 255.0) - 1).to(dev); b = (2 * (im2 / 255.0) - 1).to(dev)
def run(cl, amp=None):
    if cl:
        model.fnet.to(memory_format=torch.channels_last); model.cnet.to(memory_format=torch.channels_last)
        x, y = a.contiguous(memory_format=torch.channels_last), b.contiguous(memory_format=torch.channels_last)
    else:
        x, y = a, b
    for it in range(4):
        for p in model.parameters(): p.grad = None
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t2 = torch.cuda.Event(enable_timing=True)
        t0.record()
        with torch.autocast('cuda', dtype=amp, enabled=amp is not None):
            f1, f2 = model.fnet([x, y]); cn = model.cnet(x)
        f1, f2, cn = f1.float(), f2.float(), cn.float()
        t1.record()
        torch.autograd.backward([f1, f2, cn], [torch.ones_like(f1), torch.ones_like(f2), torch.ones_like(cn)])
        t2.record(); torch.cuda.synchronize()
        print(f"channels_last={cl} amp={amp} iter {it}: fwd {t0.elapsed_time(t1):.2f} ms bwd {t1.elapsed_time(t2):.2f} ms", flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else 'all'
if mode != 'hip':
    run(False)
if mode == 'all':
    run(True)
if mode != 'hip':
    run(False, torch.bfloat16)
    run(False, torch.float16)
if mode == 'all':
    run(True, torch.bfloat16)


def run_hip(prec_name="train_f16x3"):
    from craft_amd.train_encoder import encoder_forward_train
    from craft_amd.hip import Precision
    prec = Precision.parse(prec_name)
    raw1, raw2 = im1.to(dev), im2.to(dev)
    for it in range(4):
        for p in model.parameters(): p.grad = None
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t2 = torch.cuda.Event(enable_timing=True)
        t0.record()
        f12 = encoder_forward_train(model.fnet, torch.cat([raw1, raw2], 0), prec); cn = encoder_forward_train(model.cnet, raw1, prec)
        t1.record()
        torch.autograd.backward([f12, cn], [torch.ones_like(f12), torch.ones_like(cn)])
        t2.record(); torch.cuda.synchronize()
        print(f"hip {prec_name} iter {it}: fwd {t0.elapsed_time(t1):.2f} ms bwd {t1.elapsed_time(t2):.2f} ms", flush=True)


if mode in ("all", "hip"):
    run_hip()
