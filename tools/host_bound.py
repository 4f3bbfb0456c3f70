"""Is the forward host-bound?  Time from call to return (all launches enqueued) vs to device completion.
usage (GPU box): python tools/host_bound.py [H W B iters]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    H, W, B, T = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (448, 1024, 4, 12)))
    dev = torch.device("cuda:0")
    from craft_amd import CRAFT, default_args
    from craft_amd.synth import synth_pair, synth_state_dict
    model = CRAFT(default_args(hip_precision="mixed"))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
    model = model.to(dev).eval()
    im1, im2, _ = synth_pair(B, H, W, seed=100)
    im1, im2 = im1.to(dev), im2.to(dev)
    with torch.no_grad():
        for _ in range(3):
            model(im1, im2, iters=T, test_mode=1)
        torch.cuda.synchronize()
        enq, tot = [], []
        for _ in range(8):
            t0 = time.perf_counter()
            model(im1, im2, iters=T, test_mode=1)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
    print(f"enqueue {sum(enq) / len(enq):.2f} ms   total {sum(tot) / len(tot):.2f} ms   (min enqueue {min(enq):.2f}, min total {min(tot):.2f})")


if __name__ == "__main__":
    main()
