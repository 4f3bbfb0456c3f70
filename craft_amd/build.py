"""Build ``craft_amd/libcraft_hip.so`` in-tree with hipcc for gfx950 (no GPU needed to compile).

    python -m craft_amd.build [--force]

The .so is git-ignored but travels with the tree to the GPU box (see .gitignore / gpurun).
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libcraft_hip.so")
SOURCES = ["kernels_gemm.hip", "kernels_conv.hip", "kernels_conv_wf.hip", "kernels_conv_c64.hip", "kernels_stem.hip", "kernels_flash.hip", "kernels_attn.hip", "kernels_attn_d32.hip", "kernels_attn_d64.hip", "kernels_attn_d128.hip", "kernels_attn_w.hip",
           "kernels_misc.hip", "kernels_convf1.hip", "kernels_gemm_gen.hip", "kernels_gemm_pk.hip", "kernels_gemm_pkb_tt.hip", "kernels_gemm_pkb_ct.hip", "kernels_gemm_pkb_cc.hip", "kernels_train.hip", "kernels_enc_train.hip", "kernels_augment.hip", "craft_hip.hip"]
HEADERS = ["common.hpp", "gemm_engine.hpp", "launch.hpp", "attn_probs.inc.hpp", "conv_epilogue.hpp", "gemm_pkb.inc.hpp", os.path.join("..", "..", "include", "craft_hip.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, extra) -> str:
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]
    if _stale(obj, deps):
        # -amdgpu-mfma-vgpr-form: MFMA accumulators in VGPRs (every kernel here fits 256 unified registers), so the
        # epilogues read them directly instead of through one v_accvgpr_read per value
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
               "-mllvm", "-amdgpu-mfma-vgpr-form",
               "-c", os.path.join(CSRC, src), "-o", obj] + list(extra)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build_extension(force: bool = False, extra_flags=()) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
        if os.path.exists(LIB):
            os.remove(LIB)
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(lambda s: _compile(s, extra_flags), SOURCES))
    if _stale(LIB, objs):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv))
