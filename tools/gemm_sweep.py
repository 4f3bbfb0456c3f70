"""Developer sweep: per-K-tile cost of the backward GEMM engine (craft_gemm k-major x k-major, split-K) vs grid size.
usage (GPU box): python tools/gemm_sweep.py [policy]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from craft_amd import hip as H


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    pol = H.Precision.parse(sys.argv[1] if len(sys.argv) > 1 else "mixed")
    cp = H.pick(pol, "conv")
    dev = "cuda:0"
    M, N = 256, 512
    for K in (22816,):
        x = torch.randn(K, N, device=dev)
        dy = torch.randn(K, M, device=dev)
        dw = torch.zeros(M, N, device=dev)
        for ks in (1, 2, 4, 8, 16, 32, 64, 128):
            t = timeit(lambda: H.call("craft_gemm", dy, 1, M, 0, 0, x, 1, N, 0, 0, dw, N, 0, 0, 1, 1, M, N, K, 1.0, 1, ks, cp))
            blocks = 8 * ks
            tiles = (K + ks - 1) // ks / 32
            print(f"TT K={K} ksplit={ks:4d} blocks={blocks:5d} k-tiles/block={tiles:7.1f}  {t:8.1f} us  -> {t * 1e-6 * 2.2e9 / tiles:7.0f} cycles per k-tile-round",
                  flush=True)
        # rows x rows (NT) of the same size for comparison: C[M,N] = A[M,K] B[N,K]^T
        a = torch.randn(M, K, device=dev)
        b = torch.randn(N, K, device=dev)
        for ks in (1, 8, 64):
            t = timeit(lambda: H.call("craft_gemm", a, K, 1, 0, 0, b, K, 1, 0, 0, dw, N, 0, 0, 1, 1, M, N, K, 1.0, 1, ks, cp))
            tiles = (K + ks - 1) // ks / 32
            print(f"NT K={K} ksplit={ks:4d} {t:8.1f} us -> {t * 1e-6 * 2.2e9 / tiles:7.0f} cycles per k-tile-round", flush=True)
    # big square-ish: all 256 CUs busy without split-K
    for (M2, N2, K2) in ((4096, 4096, 1024), (2048, 2048, 4096)):
        a = torch.randn(K2, M2, device=dev); b = torch.randn(K2, N2, device=dev); c = torch.zeros(M2, N2, device=dev)
        t = timeit(lambda: H.call("craft_gemm", a, 1, M2, 0, 0, b, 1, N2, 0, 0, c, N2, 0, 0, 1, 1, M2, N2, K2, 1.0, 0, 1, cp))
        print(f"TT {M2}x{N2}x{K2}: {t:8.1f} us  {2.0 * M2 * N2 * K2 / t * 1e-9:.3f} PF/s logical", flush=True)
        a = torch.randn(M2, K2, device=dev); b = torch.randn(N2, K2, device=dev)
        t = timeit(lambda: H.call("craft_gemm", a, K2, 1, 0, 0, b, K2, 1, 0, 0, c, N2, 0, 0, 1, 1, M2, N2, K2, 1.0, 0, 1, cp))
        print(f"NT {M2}x{N2}x{K2}: {t:8.1f} us  {2.0 * M2 * N2 * K2 / t * 1e-9:.3f} PF/s logical", flush=True)


if __name__ == "__main__":
    main()
