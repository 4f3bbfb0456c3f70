"""The committed PMC summaries bench.py quotes from (profiles/<bench.PMC_DIR>/): every kernel instantiation the bench line looks up is
there, at the benchmarked shape, with the fields the roofline objects carry -- a renamed kernel must not turn them into nulls silently --
and NO counter file is older than the kernel source it describes (VERDICT r5 "next" 9: round 5's line quoted a round-4 counter for a
kernel that had changed since)."""
import importlib.util
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_profiles", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# (kernel instantiation as rocprofv3 names it, group of tools/run_kernel.py): what bench.py's roofline objects look up at configs[1]
LOOKUPS = (("k_flash_attn2<64, 256, 2>", "flash"), ("k_pv16<2, 7, 2>", "pv"), ("k_conv_halo_wf<3, 1, 4, false, 5, 7>", "convtok"))


def test_bench_finds_its_pmc_entries():
    bench = _bench()
    B, H8, W8 = 4, 448 // 8, 1024 // 8                       # configs[1]
    for kernel, group in LOOKUPS:
        e = bench.pmc_lookup(kernel, group, B, H8, W8)
        assert e is not None, f"{kernel} ({group}) missing from profiles/{bench.PMC_DIR}/pmc_kernels.json"
        assert e["hbm_bytes_per_launch"] > 0 and 0.0 < e["mfma_busy"] < 1.0
    assert bench.pmc_lookup(LOOKUPS[1][0], "pv", B, H8, W8 + 1) is None          # another shape: no stale constant


def test_bench_source_looks_up_existing_kernels():
    bench = _bench()
    src = open(os.path.join(ROOT, "bench.py")).read()
    kernels = json.load(open(os.path.join(ROOT, "profiles", bench.PMC_DIR, "pmc_kernels.json")))["kernels"]
    flash = re.search(r'name = "(k_flash_attn\w*)"', src).group(1)
    assert f"{flash}<64, 256, 2>" in kernels
    # the other two files of the directory carry the shapes bench.py checks before quoting them
    corr = json.load(open(os.path.join(ROOT, "profiles", bench.PMC_DIR, "pmc_corr_build.json")))
    assert corr["shape"] == [1, 96, 128] and corr["hbm_bytes_per_launch"] >= corr["algorithmic_bytes"] > 0
    wg = json.load(open(os.path.join(ROOT, "profiles", bench.PMC_DIR, "pmc_traffic_wgrad.json")))
    assert wg["kernel"] == "k_gemm_pk" and wg["mfmas_per_product"] == 1 and wg["hbm_bytes_per_launch"] > 0
    assert "profiles\", \"r" not in src.replace('profiles", PMC_DIR', ""), "bench.py must read PMC figures from profiles/<PMC_DIR> only"


def _commit_time(path):
    r = subprocess.run(["git", "log", "-1", "--format=%ct", "--", path], cwd=ROOT, capture_output=True, text=True)
    out = r.stdout.strip()
    return int(out) if r.returncode == 0 and out else None


# counter file -> the kernel sources whose behaviour it describes (a header every kernel includes counts for all of them)
COMMON = ["craft_amd/csrc/common.hpp", "craft_amd/csrc/gemm_engine.hpp", "craft_amd/csrc/launch.hpp"]
DESCRIBES = {
    "pmc_kernels.json": COMMON + ["craft_amd/csrc/kernels_gemm.hip", "craft_amd/csrc/kernels_conv_wf.hip", "craft_amd/csrc/conv_epilogue.hpp",
                                  "craft_amd/csrc/kernels_flash.hip", "craft_amd/csrc/kernels_attn.hip", "craft_amd/csrc/kernels_attn_w.hip",
                                  "craft_amd/csrc/kernels_conv_c64.hip", "craft_amd/csrc/kernels_stem.hip", "craft_amd/csrc/kernels_convf1.hip",
                                  "craft_amd/csrc/kernels_conv.hip", "craft_amd/csrc/kernels_misc.hip"],
    "pmc_corr_build.json": COMMON + ["craft_amd/csrc/kernels_attn.hip"],
    "pmc_traffic_wgrad.json": COMMON + ["craft_amd/csrc/kernels_gemm_pk.hip"],
}


@pytest.mark.parametrize("pmc_file", sorted(DESCRIBES))
def test_no_pmc_figure_is_older_than_its_kernel(pmc_file):
    """`git log -1 --format=%ct` of every kernel source a counter file describes must not be later than the commit the counters were
    collected on.  Needs the repository's history: skipped in a bare snapshot (the GPU box)."""
    bench = _bench()
    rel = os.path.join("profiles", bench.PMC_DIR, pmc_file)
    if _commit_time("bench.py") is None:
        pytest.skip("no git history here (a snapshot of the tree)")
    meta = json.load(open(os.path.join(ROOT, rel)))
    # the commit the counters were COLLECTED on (tools/pmc_r6_json.py records it; the file's own commit time would not move when a
    # re-collection reproduces the same numbers)
    t_pmc = meta.get("tree_commit_time") or _commit_time(rel)
    assert t_pmc, f"{rel} carries no collection stamp and is not committed"
    assert not meta.get("tree_dirty_kernel_files"), f"{rel} was collected on a tree with uncommitted kernel changes: {meta['tree_dirty_kernel_files']}"
    stale = []
    for srcf in DESCRIBES[pmc_file]:
        t_src = _commit_time(srcf)
        if t_src is not None and t_src > t_pmc:
            stale.append(srcf)
    assert not stale, f"{rel} is older than {stale}: re-run tools/pmc_r6.sh on the current tree (or bench.py quotes a counter of a kernel that no longer exists)"
