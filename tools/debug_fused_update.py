import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_train_update import _step
dev = torch.device("cuda")
over = eval(sys.argv[1]) if len(sys.argv) > 1 else {}
pol = sys.argv[2] if len(sys.argv) > 2 else "train_f16x3"
la, pa, ga = _step(dev, True, over, 2, 128, 160, 3, pol)
lb, pb, gb = _step(dev, False, over, 2, 128, 160, 3, pol)
print("loss", la, lb)
rows = []
for k, g in gb.items():
    n = float(g.norm())
    rows.append((float((ga[k] - g).norm()) / max(n, 1e-30), k, n))
for l2, k, n in sorted(rows, reverse=True)[:40]:
    print(f"{l2:10.3e} {n:10.3e} {k}")
