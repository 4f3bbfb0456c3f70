#!/usr/bin/env python
"""Eager launches vs one hipGraph replay of the same forward pass (configs[1] by default).

The forward pass enqueues ~600 kernels on up to three streams; between dependent kernels of one stream the command processor leaves a few
microseconds (profiles/r5/trace_gaps.txt).  A captured graph hands the whole dependency structure to the runtime at once.  This probe times
both and checks that the replay reproduces the eager result.

    python tools/graph_probe.py [--batch 4 --height 448 --width 1024 --iters 12 --steps 20]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--height", type=int, default=448)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--precision", default="mixed")
    a = ap.parse_args()
    from craft_amd import CRAFT, default_args
    from craft_amd.synth import synth_pair, synth_state_dict

    dev = torch.device("cuda", 0)
    model = CRAFT(default_args(hip_precision=a.precision))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
    model = model.to(dev).eval()
    im1, im2, _ = synth_pair(a.batch, a.height, a.width, seed=100)
    im1, im2 = im1.to(dev), im2.to(dev)

    def eager():
        with torch.no_grad():
            return model(im1, im2, iters=a.iters, test_mode=1)

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    for _ in range(4):
        lo_e, up_e = eager()
    torch.cuda.synchronize()
    t_eager = timed(eager, a.steps)
    lo_e, up_e = [t.clone() for t in eager()]

    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eager()                                  # side-stream warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.graph(g, stream=s):
        lo_g, up_g = eager()
    t_capture = 1e3 * (time.perf_counter() - t0)
    g.replay()
    torch.cuda.synchronize()
    t_graph = timed(g.replay, a.steps)
    t_eager2 = timed(eager, a.steps)
    d_up = float((up_g - up_e).abs().max())
    d_lo = float((lo_g - lo_e).abs().max())
    print(json.dumps({"shape": [a.batch, a.height, a.width], "iters": a.iters, "eager_ms": round(t_eager, 3), "eager_again_ms": round(t_eager2, 3),
                      "graph_ms": round(t_graph, 3), "capture_ms": round(t_capture, 1), "max_abs_diff_flow_up": d_up, "max_abs_diff_flow_lo": d_lo,
                      "pairs_per_s_eager": round(a.batch / t_eager * 1e3, 2), "pairs_per_s_graph": round(a.batch / t_graph * 1e3, 2)}))


if __name__ == "__main__":
    main()
