"""Helpers for the committed golden fixtures (tests/golden/*.npz, written by tools/make_golden.py)."""
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
K_SAMPLE = 4096
STRIDE = 7919
CASES = ["canon_128x256_T4", "canon_b2_128x192_T3_init", "clamp_128x160_T2", "gma_128x160_T2", "nocraft_128x160_T2",
         "f2mask_128x160_T2", "f1shared_128x160_T2", "f1private_b2_128x160_T2", "gmapos_128x160_T2", "gmaposonly_128x160_T2",
         "lsinu_b2_128x160_T2_init", "interlsinu_128x160_T2", "lsinu_warm_b2_128x160_T2", "gmaheads2_128x160_T2"]
# BASELINE.json configs[1] / configs[2] at full size, one pair each, captured from the reference (tools/make_golden.py)
FULL_CASES = ["canon_448x1024_T12", "canon_768x1024_T12"]


def sample_idx(numel: int) -> np.ndarray:
    if numel <= K_SAMPLE:
        return np.arange(numel)
    return (np.arange(K_SAMPLE, dtype=np.int64) * STRIDE) % numel


class Golden:
    def __init__(self, name: str):
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.meta = json.loads(str(self.z["meta"]))
        self.name = name

    def images(self):
        return (torch.from_numpy(self.z["image1"].astype(np.float32)), torch.from_numpy(self.z["image2"].astype(np.float32)))

    def flow_init(self):
        return torch.from_numpy(self.z["flow_init"]) if "flow_init" in self.z.files else None

    def has(self, key: str) -> bool:
        return key + ".v" in self.z.files

    def check(self, key: str, t: torch.Tensor, rtol: float, atol: float, what: str = ""):
        """Compare tensor ``t`` with the pinned sample and global (sum, sum^2) of reference tensor ``key``."""
        ref_v = self.z[key + ".v"]
        ref_s = self.z[key + ".s"]
        shape = tuple(int(x) for x in self.z[key + ".shape"])
        assert tuple(t.shape) == shape, f"{self.name}:{key}{what} shape {tuple(t.shape)} != reference {shape}"
        a = t.detach().float().cpu().contiguous().numpy().reshape(-1)
        got = a[sample_idx(a.size)]
        err = np.abs(got - ref_v)
        tol = atol + rtol * np.abs(ref_v)
        worst = int(np.argmax(err - tol))
        assert np.all(err <= tol), (f"{self.name}:{key}{what} sample mismatch: max|d|={err.max():.3e} at sample {worst} "
                                    f"(got {got[worst]:.6f}, ref {ref_v[worst]:.6f}), rtol={rtol} atol={atol}")
        d = a.astype(np.float64)
        # the second moment is a scale-aware global check (catches errors the strided sample misses)
        s2 = (d * d).sum()
        assert abs(s2 - ref_s[1]) <= (10 * rtol) * abs(ref_s[1]) + 10 * atol * np.sqrt(max(ref_s[1], 1e-30) * a.size) + 1e-12, \
            f"{self.name}:{key}{what} sum-of-squares {s2:.8e} vs reference {ref_s[1]:.8e}"
        return float(err.max())


def layout(over: dict) -> dict:
    with open(os.path.join(GOLDEN_DIR, "state_dict_layout.json")) as f:
        return json.load(f)[json.dumps(over, sort_keys=True)]
