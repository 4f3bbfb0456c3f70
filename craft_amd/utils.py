"""Small host-side helpers around the model (the reference's ``core/utils/utils.py``)."""
from __future__ import annotations

import argparse

import torch
import torch.nn.functional as F


class InputPadder:
    """Pad images so that H and W are divisible by ``mod`` (utils.py:14-31): 'sintel' pads
    symmetrically, anything else ('kitti') pads the bottom only; replicate padding."""

    def __init__(self, dims, mode: str = "sintel", mod: int = 8):
        self.ht, self.wd = dims[-2:]
        ph = (mod - self.ht % mod) % mod
        pw = (mod - self.wd % mod) % mod
        if mode == "sintel":
            self._pad = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]
        else:
            self._pad = [pw // 2, pw - pw // 2, 0, ph]

    def pad(self, *inputs):
        return [F.pad(x, self._pad, mode="replicate") for x in inputs]

    def unpad(self, x):
        ht, wd = x.shape[-2:]
        return x[..., self._pad[2]: ht - self._pad[3], self._pad[0]: wd - self._pad[1]]


def forward_interpolate(flow: torch.Tensor) -> torch.Tensor:
    """Warm-start helper (utils.py:34-62): map the flow of each pixel to the location it points to (nearest forward-
    warped source per pixel).  flow [2, H, W] or [B, 2, H, W] on the GPU -> same shape, on the GPU (``craft_forward_interpolate``;
    the reference round-trips through numpy / scipy on the CPU)."""
    from .hip import call
    if not flow.is_cuda:
        raise RuntimeError("forward_interpolate runs as a HIP kernel: the flow must be on the GPU (no CPU fallback)")
    f = flow.detach().float().contiguous()
    f4 = f[None] if f.dim() == 3 else f
    B, two, H, W = f4.shape
    if two != 2:
        raise ValueError("flow must be [2, H, W] or [B, 2, H, W]")
    out = torch.empty_like(f4)
    call("craft_forward_interpolate", f4, B, H, W, out)
    return out[0] if f.dim() == 3 else out


def coords_grid(batch: int, ht: int, wd: int, device=None) -> torch.Tensor:
    """[B, 2, ht, wd] with channel 0 = x (column), channel 1 = y (row)  (utils.py:82-85)."""
    ys, xs = torch.meshgrid(torch.arange(ht, device=device, dtype=torch.float32),
                            torch.arange(wd, device=device, dtype=torch.float32), indexing="ij")
    return torch.stack([xs, ys], dim=0)[None].expand(batch, -1, -1, -1)


def default_args(**overrides) -> argparse.Namespace:
    """The Namespace ``train.py`` / ``evaluate.py`` build for the released configuration
    (``--craft --f2 full --setrans``; train.py:311-406), for callers without an argparse front-end."""
    a = argparse.Namespace(craft=True, use_setrans=True, f1trans="none", f2trans="full", corr_radius=4,
                           pos_bias_radius=7, mixed_precision=False, dropout=0.0, num_heads=1, position_only=False,
                           position_and_content=False, f2_pos_code_weight=0.5, f2_attn_mask_radius=-1,
                           inter_num_modes=4, intra_num_modes=4, f2_num_modes=4, inter_qk_have_bias=True,
                           inter_pos_code_type="bias", inter_pos_code_weight=0.5, intra_pos_code_type="bias",
                           intra_pos_code_weight=1.0)
    for k, v in overrides.items():
        setattr(a, k, v)
    return a


def read_checkpoint(path: str, trusted: bool = False):
    """torch.load for a reference-layout checkpoint file.  The reference's trainers store {'model', 'optimizer',
    'lr_scheduler', 'logger'} (train.py:132-145); 'logger' holds numpy scalars and 'lr_scheduler' the OneCycleLR state, which
    the safe unpickler (weights_only=True, the default since torch 2.6) rejects.  First try the safe load with the numpy
    scalar globals allow-listed; only if that still fails AND the caller says the file is ``trusted`` fall back to the
    full unpickler."""
    import pickle
    try:
        import numpy as np
        safe = [np.dtype, np.ndarray]
        for name in ("_core.multiarray.scalar", "core.multiarray.scalar", "_core.multiarray._reconstruct", "core.multiarray._reconstruct"):
            mod, _, attr = name.rpartition(".")
            try:
                safe.append(getattr(__import__("numpy." + mod, fromlist=[attr]), attr))
            except (ImportError, AttributeError):
                pass
        safe += [type(np.dtype(t)) for t in ("float64", "float32", "int64", "int32")]
        with torch.serialization.safe_globals(safe):
            return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if not trusted:
            raise pickle.UnpicklingError(f"{path}: not loadable with the safe unpickler ({str(e).splitlines()[0]}); pass trusted=True "
                                         "(evaluate: --trust-checkpoint) for a checkpoint from a source you trust") from e
        return torch.load(path, map_location="cpu", weights_only=False)


def load_checkpoint(model: torch.nn.Module, path_or_dict, strict: bool = False, trusted: bool = False):
    """Load a reference-layout checkpoint: ``{'model': {'module.<key>': tensor}, ...}`` or a legacy bare
    state dict, with or without the DataParallel ``module.`` prefix (evaluate.py:1540-1547, train.py:147-154).
    Only ``ck['model']`` is read."""
    ck = read_checkpoint(path_or_dict, trusted) if isinstance(path_or_dict, str) else path_or_dict
    sd = ck["model"] if isinstance(ck, dict) and "model" in ck else ck
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    msg = model.load_state_dict(sd, strict=strict)
    from .hip import bump_weights_epoch
    bump_weights_epoch()              # packed / folded weight caches (and a trainer's batched re-pack registry) must not outlive the load
    return msg
