"""Developer experiment: error of a build of the library against the full-size golden
(tests/golden/canon_448x1024_T12.npz, the reference's fp32 output) under the "mixed" policy -- used with
CRAFT_HIP_LIB=<variant built by tools/build_variant.py -DCRAFT_X3_TERMS=5|6> to price the two-MFMA forms of the split-fp16 product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from golden_util import Golden
from test_hip_e2e import build

dev = torch.device("cuda:0")
g = Golden("canon_448x1024_T12")
model = build(g, dev, sys.argv[1] if len(sys.argv) > 1 else "mixed")
im1, im2 = g.images()
with torch.no_grad():
    lo, up = model(im1.to(dev), im2.to(dev), iters=12, test_mode=1)
z = g.z
from golden_util import sample_idx
last = f"up{g.meta['iters'] - 1}"
a = up.cpu().numpy().reshape(-1)
idx = sample_idx(a.size)
d = a[idx] - z[last + ".v"]
# the sample interleaves x / y components of different pixels, so report component errors (|d| <= EPE delta)
print(f"flow_up vs the reference's fp32 output (strided sample of {d.size}): mean |d| {np.abs(d).mean():.5f} px  rms {np.sqrt((d * d).mean()):.5f} px  "
      f"max |d| {np.abs(d).max():.5f} px")
