import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_train_update import _step
dev = torch.device("cuda")
over = eval(sys.argv[1]) if len(sys.argv) > 1 else {}
pol = sys.argv[2] if len(sys.argv) > 2 else "train_f16x3"
T = int(sys.argv[3]) if len(sys.argv) > 3 else 3
la, pa, ga = _step(dev, True, over, 2, 128, 160, T, pol)
la2, pa2, ga2 = _step(dev, True, over, 2, 128, 160, T, pol)
lb, pb, gb = _step(dev, False, over, 2, 128, 160, T, pol)
print("loss", la, la2, lb)
rows = []
for k, g in gb.items():
    n = float(g.norm())
    rows.append((float((ga[k] - g).norm()) / max(n, 1e-30), float((ga[k] - ga2[k]).norm()) / max(n, 1e-30), k, n, float(ga[k].norm())))
for l2, rep, k, n, na in sorted(rows, reverse=True)[:12]:
    print(f"{l2:10.3e} (run-to-run {rep:9.2e}) |ref| {n:10.3e} |fused| {na:10.3e} {k}")
