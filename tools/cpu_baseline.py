#!/usr/bin/env python3
"""Time the CPU oracle (the reference's forward restated on torch-CPU, fp32) on the host cores.
Prints one JSON object.  Used by bench.py (in a subprocess with a wall-clock bound)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--height", type=int, default=448)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--train", action="store_true",
                    help="time one TRAINING step (oracle.craft_train_forward + sequence_loss + torch autograd backward) on one pair")
    ap.add_argument("--freeze-bn", action="store_true")
    a = ap.parse_args()
    if a.threads > 0:
        os.environ["OMP_NUM_THREADS"] = str(a.threads)
        os.environ["MKL_NUM_THREADS"] = str(a.threads)
    import torch
    if a.threads > 0:
        torch.set_num_threads(a.threads)
    from craft_amd import CRAFT, default_args
    from craft_amd.synth import synth_pair, synth_state_dict
    from oracle import craft_oracle as O
    sd = synth_state_dict(CRAFT(default_args()).state_dict(), seed=1234)
    im1, im2, flow = synth_pair(1, a.height, a.width, seed=0)
    if a.train:
        names = {k for k, _ in CRAFT(default_args()).named_parameters()}
        sd = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
        t0 = time.time()
        preds, _ = O.craft_train_forward(sd, O.OracleConfig(), im1, im2, iters=a.iters, freeze_bn=a.freeze_bn)
        loss = O.sequence_loss(preds, flow, torch.ones(1, a.height, a.width), 0.8)
        loss = loss[0] if isinstance(loss, tuple) else loss
        loss.backward()
        dt = time.time() - t0
        print(json.dumps({"value": round(1.0 / dt, 4), "unit": "image-pairs/sec", "cores": torch.get_num_threads(), "kind": "port",
                          "sample": f"1 pair {a.height}x{a.width}, {a.iters} iters: forward + sequence loss + autograd backward of the fp32 "
                                    f"torch-CPU oracle (oracle/craft_oracle.py craft_train_forward; no optimizer step), {dt:.1f} s wall, "
                                    f"{torch.get_num_threads()} threads of {os.cpu_count()} logical CPUs"}))
        return
    t0 = time.time()
    O.craft_forward(sd, O.OracleConfig(), im1, im2, iters=a.iters, test_mode=1)
    dt = time.time() - t0
    print(json.dumps({"value": round(1.0 / dt, 4), "unit": "image-pairs/sec", "cores": torch.get_num_threads(), "kind": "port",
                      "sample": f"1 pair {a.height}x{a.width}, {a.iters} iters, fp32 torch-CPU oracle (oracle/craft_oracle.py), "
                                f"{dt:.1f} s wall, {torch.get_num_threads()} threads of {os.cpu_count()} logical CPUs"}))


if __name__ == "__main__":
    main()
