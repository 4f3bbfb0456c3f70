#!/usr/bin/env python3
"""Static ISA scan of the library's kernels (no GPU needed): per kernel VGPRs / scratch / occupancy from hipcc's resource remarks, the
number of vector-memory loads, and how many of them are followed within three instructions by `s_waitcnt vmcnt(0)` -- the signature of
a load under a bounds test or of a rolled copy loop (branch + load + full wait: one dependent round trip each), which cost the small
per-iteration kernels 30-40 % before round 4's second half.  Static counts include cold paths (ragged-edge epilogues).
usage: python tools/isa_scan.py [source.hip ...]   (default: every craft_amd/csrc/kernels_*.hip; ~1 min each, run in parallel)"""
import concurrent.futures, glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-mllvm", "-amdgpu-mfma-vgpr-form"]


def scan(src):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-S", "--cuda-device-only", "-o", asm, src],
                           capture_output=True, text=True)
        if r.returncode:
            return src, None, r.stderr[-500:]
        res = {}
        for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)", r.stderr, re.S):
            res[m.group(1)] = dict(vgpr=int(m.group(2)), scratch=int(m.group(3)), occ=int(m.group(4)), loads=0, adj=0)
        name, last = None, -100
        for i, l in enumerate(open(asm)):
            m = re.match(r"^(_Z\w+):", l)
            if m:
                name, last = m.group(1), -100
                continue
            if name not in res:
                continue
            ls = l.strip()
            if ls.startswith(("global_load", "buffer_load")):
                res[name]["loads"] += 1
                last = i
            elif ls.startswith("s_waitcnt") and "vmcnt(0)" in ls and i - last <= 3:
                res[name]["adj"] += 1
        return src, res, ""


def main():
    srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "craft_amd", "csrc", "kernels_*.hip")))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        out = list(ex.map(scan, srcs))
    try:
        import subprocess as sp
        filt = lambda n: sp.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n      # noqa: E731
    except Exception:       # noqa: BLE001
        filt = lambda n: n                                                                                # noqa: E731
    print(f"{'file':22s} {'vgpr':>4s} {'scr':>4s} {'occ':>3s} {'loads':>5s} {'ld->vmcnt(0)':>12s}  kernel")
    for src, res, err in out:
        if res is None:
            print(os.path.basename(src), "FAILED:", err)
            continue
        for k, v in sorted(res.items(), key=lambda kv: (-kv[1]["scratch"], -kv[1]["adj"])):
            if v["scratch"] or v["adj"] >= 3:
                print(f"{os.path.basename(src)[:22]:22s} {v['vgpr']:4d} {v['scratch']:4d} {v['occ']:3d} {v['loads']:5d} {v['adj']:12d}  {filt(k)[:110]}")


if __name__ == "__main__":
    main()
