"""Clock / power of the chip while one kernel family runs back to back for a few seconds (rocm-smi sampled from a thread): the f16x3
tile stream of craft_gemm_pk against an HBM-bound pass.  usage: python tools/power_probe.py"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from craft_amd import autograd as AG, hip
from craft_amd.hip import round_up

dev = torch.device("cuda")
B, Mh, N, C, T = 8, 4, 46 * 62, 128, 12
ld = round_up(N, 32)
P = torch.zeros(B, Mh, N, ld, device=dev)
P[..., :N] = torch.softmax(torch.randn(B, Mh, N, N, device=dev), dim=-1)
dO = torch.randn(B, Mh, N, T * C, device=dev)
dV = torch.empty(B, N, Mh, T * C, device=dev)


def sample(stop, out):
    while not stop.is_set():
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        sclk = [l.split("(")[-1].rstrip(")") for l in r.splitlines() if "sclk" in l]
        pw = [l.split(":")[-1].strip() for l in r.splitlines() if "Power" in l and "W" in l]
        out.append((sclk[:1], pw[:1]))
        time.sleep(0.4)


for name, prec in (("f16x3 dV = P^T dO (craft_gemm_pk, 256 x 256 tiles)", hip.PREC_F16X3), ("bf16  dV = P^T dO", hip.PREC_BF16), ("HBM pass (softmax fwd)", None)):
    if prec is not None:
        Ppk = AG.PkMat(B * Mh, N, ld, prec, dev).fill(P)
        dOpk = AG.PkMat(B * Mh, N, T * C, prec, dev).fill(dO)
        TC = T * C
        fn = lambda: AG.gemm_pk(Ppk, Ppk.desc(AG.PK_ROWS, Mh, 1), dOpk, dOpk.desc(AG.PK_ROWS, Mh, 1), dV, Mh * TC, N * Mh * TC, TC, Mh, B * Mh, N, TC, N)  # noqa: E731
    else:
        S = torch.randn(B, Mh, N, ld, device=dev)
        fn = lambda: AG.call("craft_attn_softmax_fwd", S, ld, B, Mh, 46, 62, None, 0, 0.0, -1, None, None, None, 0.0, 0, None, 0, 0, 0)  # noqa: E731
    fn(); torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 4.0:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    print(f"{name}: {e0.elapsed_time(e1) / n * 1e3:8.1f} us per launch over {n} launches; rocm-smi samples (sclk, power): {out[1:-1]}", flush=True)
