"""Is the training step host-bound?  Wall time until Trainer.step has enqueued everything vs until the device is done.
(Trainer.step reads the loss back at its end, so the enqueue time is measured on forward + backward only.)
usage (GPU box): python tools/host_bound_train.py [3|4]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from craft_amd import CRAFT, default_args
from craft_amd import autograd as AG
from craft_amd.synth import synth_pair, synth_state_dict


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    H, W, B, policy = {3: (368, 496, 8, "train_f16x3"), 4: (368, 768, 4, "train_bf16attn")}[cfg]
    dev = torch.device("cuda:0")
    model = CRAFT(default_args(hip_precision=policy))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
    model = model.to(dev).train()
    if cfg == 4:
        model.freeze_bn()
    im1, im2, flow = synth_pair(B, H, W, seed=100)
    im1, im2, flow = im1.to(dev), im2.to(dev), flow.to(dev)
    valid = torch.ones(B, H, W, device=dev)
    for it in range(6):
        for p in model.parameters():
            p.grad = None
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        t0 = time.perf_counter()
        e[0].record()
        preds = model(im1, im2, iters=12)
        e[1].record()
        t1 = time.perf_counter()
        loss = AG.SequenceLoss.apply(flow, valid, 0.8, 400.0, *preds)          # (no metrics: nothing reads back)
        e[2].record()
        t1b = time.perf_counter()
        loss.backward(torch.full((), 65536.0, device=dev))
        e[3].record()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print(f"iter {it}: forward host {1e3 * (t1 - t0):6.2f} / device {e[0].elapsed_time(e[1]):6.2f} ms, loss host {1e3 * (t1b - t1):5.2f} / device "
              f"{e[1].elapsed_time(e[2]):5.2f}, backward host {1e3 * (t2 - t1b):6.2f} / device {e[2].elapsed_time(e[3]):6.2f} ms, all done {1e3 * (t3 - t0):6.2f} ms", flush=True)


if __name__ == "__main__":
    main()
