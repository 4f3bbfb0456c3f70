"""Soundness check of the guard-page allocator itself: aten-only traffic (no craft_* kernel) through randomly sized buffers, every result
compared with the CPU.  A failure here is the harness (or the driver's unmap path), not the product.  python tools/guard_run.py --script tools/guard_selftest.py"""
import sys
import torch
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
bad = 0
keep = []
for it in range(4000):
    n = int(torch.randint(1, 200000, (1,), generator=g))
    x = torch.randn(n, generator=g)
    xd = x.to(dev)
    yd = xd * 2 + 1
    z = torch.zeros(n, device=dev)
    z += yd
    if not torch.equal(z.cpu(), x * 2 + 1) or not torch.equal(xd.cpu(), x):
        bad += 1
        if bad < 5:
            print(f"iteration {it}: mismatch at n = {n}", flush=True)
    if it % 7 == 0:
        keep.append(xd)
        if len(keep) > 50:
            keep.pop(0)
print(f"guard allocator self-test: {bad} mismatches in 4000 iterations")
sys.exit(1 if bad else 0)
