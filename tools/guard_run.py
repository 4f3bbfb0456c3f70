#!/usr/bin/env python3
"""Run pytest (or a script) with every device tensor in its own guard-paged virtual range (tools/guard_alloc.cpp): an out-of-bounds or
use-after-free access by ANY kernel -- ours or aten's -- becomes a deterministic GPU page fault, charged to the entry point that caused it
(CRAFT_HIP_DEBUG synchronises after every craft_* call and keeps the last call's name and arguments in a file).  VERDICT r5 "next" 1 (ii).

  python tools/guard_run.py [--mode end|start] [--no-sync] [--log FILE] -- <pytest args ...>
  python tools/guard_run.py --script tools/foo.py [args]
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="end", choices=("end", "start"))
ap.add_argument("--no-sync", action="store_true", help="do not synchronise after every craft_* call (asynchronous races stay possible)")
ap.add_argument("--log", default=os.path.join(ROOT, "gpurun_out", "guard_last_call.txt"))
ap.add_argument("--fill", default="255")
ap.add_argument("--script", default=None)
ap.add_argument("rest", nargs=argparse.REMAINDER)
a = ap.parse_args()
rest = [r for r in a.rest if r != "--"]

so = os.path.join(ROOT, "tools", "libguard_alloc.so")
if not os.path.exists(so):
    subprocess.check_call(["g++", "-O1", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tools", "guard_alloc.cpp"),
                           "-L/opt/rocm/lib", "-lamdhip64", "-o", so])
os.makedirs(os.path.dirname(a.log), exist_ok=True)
os.environ["GUARD_ALLOC_MODE"] = a.mode
os.environ["GUARD_ALLOC_FILL"] = a.fill
os.environ["CRAFT_NO_ZERO_POOL"] = "1"
if not a.no_sync:
    os.environ["CRAFT_HIP_DEBUG"] = a.log
    os.environ.setdefault("HIP_LAUNCH_BLOCKING", "1")
    os.environ.setdefault("AMD_SERIALIZE_KERNEL", "3")
sys.path.insert(0, ROOT)
import torch

alloc = torch.cuda.memory.CUDAPluggableAllocator(so, "guard_malloc", "guard_free")
torch.cuda.memory.change_current_allocator(alloc)
print(f"[guard_run] allocator installed ({a.mode}); last craft_* call -> {a.log}", file=sys.stderr, flush=True)
if a.script:
    import runpy
    sys.argv = [a.script] + rest
    runpy.run_path(a.script, run_name="__main__")
    rc = 0
else:
    import pytest
    rc = pytest.main(rest)
import ctypes
ctypes.CDLL(so).guard_stats()
sys.exit(rc)
