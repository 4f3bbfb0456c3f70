"""bench.py's output contract on the GPU box: one JSON line with the driver's keys plus roofline / roofline_conv /
cpu_baseline at N = 1, and the N > 1 launch path (torch.distributed.run, barrier + max-over-ranks timing, rank 0
prints) exercised with two ranks that share the one GPU over gloo (CRAFT_BENCH_BACKEND=gloo; the driver's real runs use
RCCL with one rank per GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def _json_line(out: str) -> dict:
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected exactly one JSON line, got {len(lines)}:\n{out[-2000:]}"
    return json.loads(lines[0])


def test_single_gpu_line(device):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--batch", "1", "--height", "128", "--width", "256",
                        "--iters", "2"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "workload" in d["config"]
    # the short configs[3] training leg rides on the default line (always at its own shape: 368x496, batch 8)
    t3 = d["train_cfg3"]
    assert t3["ms_per_step"] > 0 and t3["pairs_per_s"] > 0 and t3["steps"] == 5 and "368x496" in t3["workload"]
    assert t3["amp_fp16"]["policy"] == "train_amp_fp16" and t3["amp_fp16"]["ms_per_step"] > 0 and t3["amp_fp16"]["loss"] == t3["amp_fp16"]["loss"]
    rw = t3["roofline"]
    assert rw["bound"] == "mfma" and rw["unit"] == "TFLOP/s" and abs(rw["frac"] - rw["achieved"] / rw["peak"]) < 1e-3
    assert abs(rw["achieved"] - rw["flops_per_launch"] / (rw["ms_per_launch"] * 1e-3) / 1e12) / rw["achieved"] < 0.02
    # configs[4] (training, 368x768 batch 4) and configs[2] (correlation stress at 768x1024) ride on the same line
    t4 = d["train_cfg4"]
    assert t4["ms_per_step"] > 0 and t4["pairs_per_s"] > 0 and t4["steps"] == 5 and "368x768" in t4["workload"] and t4["loss"] == t4["loss"]
    c2 = d["corr_cfg2"]
    assert c2["build_ms"] > 0 and c2["lookup_ms"] > 0 and "768x1024" in c2["workload"] and c2["bound"] == "hbm"
    # SURVEY 8(d) bytes: the 4-level pyramid of a 96x128 key image per query, written once, + Q and K read once
    n = 96 * 128
    assert c2["bytes"] == 4 * n * (96 * 128 + 48 * 64 + 24 * 32 + 12 * 16) + 2 * n * 256 * 4
    assert abs(c2["frac"] - c2["bytes"] / (c2["build_ms"] * 1e-3) / 8e12) < 2e-3


def test_training_line(device):
    """`bench.py --train 3` (reduced shape): the same contract keys, roofline of the weight-gradient kernel, the CPU oracle's
    training step as cpu_baseline, vs_baseline null (BASELINE.md publishes no number for this metric)."""
    r = subprocess.run([sys.executable, "bench.py", "--train", "3", "--steps", "2", "--warmup", "1", "--batch", "2", "--height", "128",
                        "--width", "160", "--iters", "2"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["value"] > 0 and d["vs_baseline"] is None and d["loss"] == d["loss"]
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["launches_timed"] > 0
    assert d["amp_fp16"]["policy"] == "train_amp_fp16" and d["amp_fp16"]["pairs_per_s"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and "backward" in cb["sample"]


def _backends():
    import torch
    return ["gloo"] + (["nccl"] if torch.cuda.device_count() >= 2 else [])


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_rank_training_launch_path(device, backend):
    """`bench.py --train 3 --gpus 2`: two ranks, one gradient all-reduce per step.  gloo: ranks share the GPU (host-staged
    all-reduce); nccl: RCCL with one rank per GPU (skipped on a 1-GPU box) -- the first execution of the device all-reduce."""
    if backend not in _backends():
        pytest.skip("needs >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CRAFT_BENCH_BACKEND"] = backend
    r = subprocess.run([sys.executable, "bench.py", "--train", "3", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "1",
                        "--height", "128", "--width", "160", "--iters", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 and d["value"] > 0 and "cpu_baseline" not in d
    assert d["allreduce_ms_per_step"] is not None and d["allreduce_ms_per_step"] > 0


def test_two_rank_inference_nccl(device):
    """The inference launch path under RCCL with one rank per GPU (skipped on a 1-GPU box)."""
    if "nccl" not in _backends():
        pytest.skip("needs >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CRAFT_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "1", "--height", "128",
                        "--width", "256", "--iters", "2", "--no-train-leg"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 and d["value"] > 0


def test_two_rank_launch_path(device):
    import socket
    with socket.socket() as sk:                      # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, CRAFT_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "1", "--height", "128",
           "--width", "256", "--iters", "2", "--no-train-leg"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 and d["value"] > 0 and "cpu_baseline" not in d
    # value = pairs of ALL ranks / slowest rank's time
    assert abs(d["value"] - 2 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) / d["value"] < 0.02


def test_self_launch_gpus_flag(device):
    """Plain `python bench.py --gpus 2` (no launcher, RANK unset) starts two ranks itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CRAFT_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "1", "--height", "128",
                        "--width", "256", "--iters", "2", "--no-train-leg"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 and d["value"] > 0


def test_gpus_flag_refuses_missing_devices(device):
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "CRAFT_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "visible" in (r.stderr + r.stdout)
