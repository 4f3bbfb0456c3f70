"""The dropout masks of the HIP training step, restated (test infrastructure; nothing of the product is called here).

``craft_dropout`` and the fused softmax kernels (csrc/kernels_train.hip) keep element i of a tensor iff
    mix32(seed * 0x9E3779B97F4A7C15 + i) >= (unsigned)(p * 2^32)          (mix32 = murmur3's 64-bit finaliser, low 32 bits)
and scale the kept values by 1 / (1 - p) in float32; i is the flat element index of the tensor -- for attention probabilities, which live in
rows of ld = N rounded up to 32 floats, i = ((b M + m) N + row) ld + column.  craft_amd/train_forward.py gives every dropout site of a pass
its own seed: base + 1 F2 tokens, + 2 F2 probabilities, + 3 / + 4 the correlation block's encoder on frame 1 / frame 2, + 5 / + 6 the
intra-frame attention's tokens / probabilities, base = (torch.initial_seed() * 1000003 + 64 * pass number) & (2^59 - 1).
The reference applies nn.Dropout at exactly these six places (setrans.py:791-795 on each SETransInputFeatEncoder, :553-557 on the two
self-attentions; verified by hooking nn.Dropout.forward of the imported reference: tools/make_golden_train_dropout.py).
"""
import numpy as np
import torch

PHI = 0x9E3779B97F4A7C15
MASK64 = (1 << 64) - 1


def mix32(k: np.ndarray) -> np.ndarray:
    k = k.copy()
    with np.errstate(over="ignore"):
        k ^= k >> np.uint64(33)
        k *= np.uint64(0xff51afd7ed558ccd)
        k ^= k >> np.uint64(33)
        k *= np.uint64(0xc4ceb9fe1a85ec53)
        k ^= k >> np.uint64(33)
    return (k & np.uint64(0xFFFFFFFF)).astype(np.uint64)


def dropout_scale(seed: int, index: np.ndarray, p: float) -> np.ndarray:
    """Per flat element index: 1 / (1 - p) where the element is kept, 0 where it is dropped (float32)."""
    p32 = np.float32(p)
    thr = np.uint64(int(min(np.float32(p32 * np.float32(4294967296.0)), np.float32(4294967040.0))))
    base = np.uint64((int(seed) * PHI) & MASK64)
    with np.errstate(over="ignore"):
        keep = mix32(base + index.astype(np.uint64)) >= thr
    inv = np.float32(1.0) / (np.float32(1.0) - p32)
    return np.where(keep, inv, np.float32(0.0)).astype(np.float32)


def token_mask(seed: int, shape, p: float) -> torch.Tensor:
    n = int(np.prod(shape))
    return torch.from_numpy(dropout_scale(seed, np.arange(n, dtype=np.uint64), p).reshape(shape))


def probs_mask(seed: int, B: int, M: int, N: int, p: float) -> torch.Tensor:
    ld = (N + 31) // 32 * 32
    rows = np.arange(B * M * N, dtype=np.uint64)[:, None] * np.uint64(ld)
    idx = rows + np.arange(N, dtype=np.uint64)[None, :]
    return torch.from_numpy(dropout_scale(seed, idx, p).reshape(B, M, N, N))


def pass_base(torch_seed: int, pass_number: int = 0) -> int:
    return (int(torch_seed) * 1000003 + pass_number * 64) & 0x7FFFFFFFFFFFFFF


def pass_masks(base: int, B: int, N: int, M: int = 4, p_hidden: float = 0.1, p_attn: float = 0.2, c_feat: int = 256, c_ctx: int = 128):
    """The six masks of one training pass of the canonical configuration, keyed like oracle.craft_oracle.DROPOUT_MASKS."""
    return {"f2_trans.hidden": token_mask(base + 1, (B, N, c_feat), p_hidden), "f2_trans.attn": probs_mask(base + 2, B, M, N, p_attn),
            "corr_fn.x1": token_mask(base + 3, (B, N, c_feat), p_hidden), "corr_fn.x2": token_mask(base + 4, (B, N, c_feat), p_hidden),
            "att.hidden": token_mask(base + 5, (B, N, c_ctx), p_hidden), "att.attn": probs_mask(base + 6, B, M, N, p_attn)}
