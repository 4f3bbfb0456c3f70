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
        t0 = time.perf_counter()
        preds = model(im1, im2, iters=12)
        loss, _ = AG.sequence_loss(preds, flow, valid, 0.8)
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print(f"iter {it}: forward enqueued {1e3 * (t1 - t0):7.2f} ms, backward enqueued {1e3 * (t2 - t1):7.2f} ms, device done {1e3 * (t3 - t0):7.2f} ms", flush=True)


if __name__ == "__main__":
    main()
