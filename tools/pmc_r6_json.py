#!/usr/bin/env python3
"""gpurun_out/pmc_r6 (tools/pmc_r6.sh) -> profiles/r6/{pmc_kernels.json, pmc_corr_build.json, pmc_traffic_wgrad.json}: the three files
bench.py reads its PMC figures from.  usage: python tools/pmc_r6_json.py gpurun_out/pmc_r6 profiles/r6"""
import json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def first_kernel(path, needle):
    """(mean KiB, launches) of the first kernel whose name contains `needle` in a tools/kstats.py counter summary."""
    cur = None
    for l in open(path):
        if not l.startswith(" "):
            cur = l.strip()
        elif cur and needle in cur:
            m = re.match(r"\s+(\S+)\s+n=\s*(\d+) mean=(\S+)", l)
            if m:
                return float(m.group(3)), int(m.group(2)), cur
    raise SystemExit(f"{path}: no kernel matching {needle!r}")


def tree_stamp():
    """The commit the counters were collected on (HEAD when this script runs: call it right after tools/pmc_r6.sh, before the next commit)
    and its commit time: tests/test_profiles.py holds every kernel source to be no newer than this."""
    g = lambda *a: subprocess.run(["git", *a], cwd=ROOT, capture_output=True, text=True).stdout.strip()      # noqa: E731
    dirty = [l for l in g("status", "--porcelain", "--", "craft_amd/csrc", "include").splitlines() if l.strip()]
    return {"tree": g("rev-parse", "--short", "HEAD"), "tree_commit_time": int(g("log", "-1", "--format=%ct") or 0), "tree_dirty_kernel_files": dirty}


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    stamp = tree_stamp()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), src], capture_output=True, text=True, check=True).stdout
    pk = json.loads(out)
    pk.update(stamp)
    with open(os.path.join(dst, "pmc_kernels.json"), "w") as f:
        json.dump(pk, f, indent=1)
    f_kb, n, name = first_kernel(os.path.join(src, "corr2_FETCH_SIZE.txt"), "k_corr_build4t")
    w_kb, _, _ = first_kernel(os.path.join(src, "corr2_WRITE_SIZE.txt"), "k_corr_build4t")
    H8, W8 = 96, 128
    nq = H8 * W8
    alg = 4 * nq * (96 * 128 + 48 * 64 + 24 * 32 + 12 * 16) + 2 * nq * 256 * 4
    with open(os.path.join(dst, "pmc_corr_build.json"), "w") as f:
        json.dump({**stamp, "kernel": "k_corr_build4t (fused scores + mode pooling + 4-level pyramid, tiled levels 0 / 1, level 0 staged through LDS)",
                   "shape": [1, H8, W8], "shape_legend": "B, H8, W8 (configs[2]: 768x1024)",
                   "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over tools/bench_corr.py --reps 4 (tools/pmc_r6.sh); {n} launches averaged",
                   "fetch_kb_raw": f_kb, "write_kb_raw": w_kb, "fetch_correction": 2.0,
                   "hbm_bytes_per_launch": int(f_kb * 1024 * 2 + w_kb * 1024), "hbm_bytes_per_launch_fetch_uncorrected": int(f_kb * 1024 + w_kb * 1024),
                   "algorithmic_bytes": alg, "traffic_over_algorithmic": round((f_kb * 2048 + w_kb * 1024) / alg, 3),
                   "note": "FETCH_SIZE x2 is the guide's correction for wide coalesced reads (MI355X_MICROARCH.md, HBM); part of the fetches here are "
                           "read-modify-write fills behind the partial-line stores of levels 2 and 3, whose request width is uncalibrated: the x2 figure is an upper bound"}, f, indent=1)
    f_kb, n, name = first_kernel(os.path.join(src, "wgrad_FETCH_SIZE.txt"), "k_gemm_pk")
    w_kb, _, _ = first_kernel(os.path.join(src, "wgrad_WRITE_SIZE.txt"), "k_gemm_pk")
    with open(os.path.join(dst, "pmc_traffic_wgrad.json"), "w") as f:
        json.dump({**stamp, "kernel": "k_gemm_pk", "kernel_instantiation": name, "shape": [128, 256, 3, 3, 8 * 46 * 62, 12],
                   "shape_legend": "cin, cout, KH, KW, pixels per call, calls per launch (flow head / mask head conv1 at configs[3])",
                   "mfmas_per_product": 1, "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over tools/run_wgrad_pk.py 4 fp16 (tools/pmc_r6.sh); {n} launches averaged",
                   "fetch_kb_raw": f_kb, "write_kb_raw": w_kb, "fetch_correction": 2.0, "hbm_bytes_per_launch": int(f_kb * 2048 + w_kb * 1024)}, f, indent=1)
    print("wrote", dst, "kernels:", len(pk.get("kernels", {})))


if __name__ == "__main__":
    main()
