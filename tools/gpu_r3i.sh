#!/usr/bin/env bash
export TMPDIR=/tmp
python - <<'PY' 2>/dev/null
import sys, time, torch, gc
sys.path.insert(0, '.')
from craft_amd import CRAFT, default_args
from craft_amd.synth import synth_pair, synth_state_dict
from craft_amd.train import Trainer
dev = torch.device("cuda")
model = CRAFT(default_args(hip_precision="train_f16x3")); model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234)); model = model.to(dev)
tr = Trainer(model, lr=4e-4, wdecay=1e-4, num_steps=100000, iters=12, clip=1.0)
im1, im2, flow = synth_pair(8, 368, 496, seed=100); im1, im2, flow = im1.to(dev), im2.to(dev), flow.to(dev); valid = torch.ones(8, 368, 496, device=dev)
for mode in ("default", "gc.freeze after warm-up", "gc disabled"):
    if mode == "gc disabled": gc.disable()
    ts = []
    for i in range(24):
        if mode.startswith("gc.freeze") and i == 2: gc.collect(); gc.freeze()
        torch.cuda.synchronize(); t0 = time.perf_counter(); tr.step(im1, im2, flow, valid); torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    print(mode, " ".join(f"{t:.0f}" for t in ts), "gc counts", gc.get_count())
PY
