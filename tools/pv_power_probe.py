"""Clocks / power while k_pv16 runs back to back (rocm-smi sampled from a thread), for the library named by CRAFT_HIP_LIB: is the
MFMA phase's cost in an HBM-bound kernel a clock / power effect?  usage: [CRAFT_HIP_LIB=...] python tools/pv_power_probe.py"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from craft_amd import ops
from craft_amd.hip import PROB_DTYPE, Precision, pick

dev = torch.device("cuda")
prec = Precision.parse("mixed")
B, H8, W8, M, Dv = 4, 56, 128, 4, 128
N = H8 * W8
pv = pick(prec, "pv")
P = ops.probs_tiled(torch.rand(B, M, N, N, device=dev).div_(N / 2).to(PROB_DTYPE[pv]))
vT = torch.randn(B, M * Dv, N, device=dev).to(PROB_DTYPE[pv])
O = torch.empty(B, M, N, Dv, device=dev)


def sample(stop, out):
    while not stop.is_set():
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        keep = []
        for l in r.splitlines():
            for k in ("sclk", "mclk", "fclk", "socclk", "Power"):
                if k in l and "GPU[0]" in l:
                    keep.append(k + "=" + l.split(":")[-1].strip().split("(")[-1].rstrip(")"))
        out.append(" ".join(keep))
        time.sleep(0.3)


for _ in range(5):
    ops.attn_apply(P, vT, Dv, pv, out=O)
torch.cuda.synchronize()
stop, out = threading.Event(), []
th = threading.Thread(target=sample, args=(stop, out)); th.start()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < 4.0:
    for _ in range(50):
        ops.attn_apply(P, vT, Dv, pv, out=O)
    n += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop.set(); th.join()
print(f"{os.environ.get('CRAFT_HIP_LIB', 'default').split('/')[-1]}: {e0.elapsed_time(e1) / n * 1e3:7.1f} us per launch over {n} launches")
for s in out[2:-1][:8]:
    print("   ", s)
