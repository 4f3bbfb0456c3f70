#!/usr/bin/env python3
"""Launch one hot-path kernel a few times at the bench shapes (for rocprofv3 --pmc passes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from craft_amd import ops  # noqa: E402
from craft_amd.hip import PROB_DTYPE, Precision, pick  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "pv"
    prec = Precision.parse(sys.argv[2] if len(sys.argv) > 2 else "mixed")
    B, H8, W8 = int(os.environ.get("B", 4)), 56, 128
    N, M, Dv = H8 * W8, 4, 128
    dev = torch.device("cuda")
    if which == "pv":
        ldp = ops.round_up(N, 32)
        pv = pick(prec, "pv")
        P = torch.rand(B, M, N, ldp, device=dev).div_(N / 2).to(PROB_DTYPE[pv])
        vT = torch.randn(B, M * Dv, ldp, device=dev)
        O = torch.empty(B, M, N, Dv, device=dev)
        for _ in range(5):
            ops.attn_apply(P, vT, Dv, pv, out=O)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
