"""The committed PMC summary bench.py quotes from (profiles/r5/pmc_kernels.json): every kernel instantiation the bench line looks up is in
it, at the benchmarked shape, with the fields the roofline objects carry -- a renamed kernel must not turn them into nulls silently."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_profiles", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_finds_its_pmc_entries():
    bench = _bench()
    B, H8, W8 = 4, 448 // 8, 1024 // 8                       # configs[1]
    for kernel, group in (("k_flash_attn2<64, 256, 2>", "flash"), ("k_pv16<2, 7>", "pv"), ("k_conv_halo_wf<3, 1, 4, false, 5, 7>", "convtok")):
        e = bench.pmc_lookup(kernel, group, B, H8, W8)
        assert e is not None, f"{kernel} ({group}) missing from profiles/r5/pmc_kernels.json"
        assert e["hbm_bytes_per_launch"] > 0 and 0.0 < e["mfma_busy"] < 1.0
    assert bench.pmc_lookup("k_pv16<2, 7>", "pv", B, H8, W8 + 1) is None          # another shape: no stale constant


def test_bench_source_looks_up_existing_kernels():
    import json
    import re
    src = open(os.path.join(ROOT, "bench.py")).read()
    kernels = json.load(open(os.path.join(ROOT, "profiles", "r5", "pmc_kernels.json")))["kernels"]
    flash = re.search(r'name = "k_flash_attn" if v1 else "(\w+)"', src).group(1)
    assert f"{flash}<64, 256, 2>" in kernels
