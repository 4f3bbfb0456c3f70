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
    # roofline_conv is timed LIVE, in situ: events around the craft_sepconv_gru_step calls of real forward passes (VERDICT r5 "next" 9)
    rc = d["roofline_conv"]
    assert rc["bound"] == "mfma" and rc["calls_timed"] >= 2 and rc["ms_per_call"] > 0 and "HIP events" in rc["timing"]
    assert abs(rc["achieved"] - rc["flops_per_call"] / (rc["ms_per_call"] * 1e-3) / 1e12) <= 0.06 + 0.01 * rc["achieved"]
    assert abs(rc["frac"] - rc["achieved"] / rc["peak"]) < 1e-3 and rc["standalone_ms_per_launch"] > 0
    # the same pass replayed as one hipGraph, timed beside the eager headline (captured in a child process first)
    hg = d["hipgraph"]
    assert hg["ms_per_step"] > 0 and hg["pairs_per_s"] > 0 and hg["steps"] == 2 and hg["max_abs_px_vs_eager"] < 1e-4, hg
    assert d["config"]["launch"] == "eager launches"
    # the reference's own default inference arithmetic beside the fp32-class headline, with its deviation from it
    ia = d["infer_amp_fp16"]
    assert ia["pairs_per_s"] > 0 and ia["finite"] and 0 <= ia["epe_vs_fp32class_mean"] < 0.5 and "evaluate.py:1455" in ia["policy"]
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


def test_four_rank_training_launch_path_gloo(device):
    """`bench.py --train 3 --gpus 4` at a miniature shape, four ranks sharing the box's GPU over gloo: the launch path, port selection,
    rank-0-only extras (roofline, amp leg, no cpu_baseline) and the barriers beyond two ranks, before the driver's SCALE run."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CRAFT_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "bench.py", "--train", "3", "--gpus", "4", "--steps", "3", "--warmup", "1", "--batch", "1",
                        "--height", "128", "--width", "160", "--iters", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 4 and d["config"]["global_batch"] == 4 and d["value"] > 0 and "cpu_baseline" not in d
    assert d["allreduce_ms_per_step"] is not None and d["allreduce_ms_per_step"] > 0 and d["skipped_steps"] == 0
    assert abs(d["value"] - 4 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) / d["value"] < 0.02
    assert d["amp_fp16"]["pairs_per_s"] > 0 and d["roofline"]["launches_timed"] > 0


def test_two_rank_default_line_at_the_benchmarked_shapes(device):
    """The driver's SCALE command at N = 2 (`bench.py --gpus 2 --steps K --warmup W`, default workload: the 448x1024 headline + the
    configs[3] / [4] training legs at their FULL shapes), two gloo ranks sharing the GPU.  Rank 0's pinned first-step loss must hold
    although the logged loss is the mean over ranks that see different pairs (the pin compares `loss_rank`); rank-0-only extras
    (rooflines, corr_cfg2) must not dead-lock the other rank; no cpu_baseline with N > 1."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CRAFT_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    import bench
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["value"] > 0 and "cpu_baseline" not in d
    assert abs(d["train_cfg3"]["first_loss"] - bench.FIRST_LOSS[(3, 368, 496, 8, 12)]) < 5e-3 * bench.FIRST_LOSS[(3, 368, 496, 8, 12)]
    assert abs(d["train_cfg4"]["first_loss"] - bench.FIRST_LOSS[(4, 368, 768, 4, 12)]) < 5e-3 * bench.FIRST_LOSS[(4, 368, 768, 4, 12)]
    assert d["train_cfg3"]["n_gpus"] == 2 and d["train_cfg3"]["allreduce_ms_per_step"] > 0 and d["roofline"]["frac"] > 0
    assert d["corr_cfg2"]["build_ms"] > 0


def test_eight_rank_default_line_miniature_gloo(device):
    """The driver's SCALE command at N = 8 (`bench.py --gpus 8 --steps K --warmup W`) as a launch-path rehearsal: eight
    gloo ranks sharing the box's GPU run the default line -- headline + training legs + corr leg -- with the extra legs at miniature shapes
    (`--mini`), so that port selection, 8-way barriers / MAX / SUM reductions, rank-0-only extras and the per-rank seeds are executed once
    before an 8-GPU node sees them.  The headline itself runs small here too (8 replicas of the 448x1024 batch would only measure contention)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CRAFT_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--mini", "--height", "128", "--width", "256", "--batch", "1",
                        "--iters", "4"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 8 and d["value"] > 0 and "cpu_baseline" not in d and "failed_legs" not in d
    assert d["train_cfg3"]["n_gpus"] == 8 and d["train_cfg3"]["allreduce_ms_per_step"] > 0 and d["train_cfg3"]["first_loss_ok"] is None
    assert d["train_cfg4"]["pairs_per_s"] > 0 and d["corr_cfg2"]["build_ms"] > 0


def test_wrong_pin_is_reported_after_the_json_line(device):
    """A first-step loss that misses its pin must not cost the run its numbers: the JSON line still goes out, carries
    first_loss / first_loss_pinned / first_loss_ok = false / pin_failed = true, and only then the process exits non-zero."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CRAFT_BENCH_BACKEND")}
    env["CRAFT_BENCH_PIN_SCALE"] = "1.5"
    r = subprocess.run([sys.executable, "bench.py", "--train", "3", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--precision", "mixed"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    d = _json_line(r.stdout)
    import bench
    pin = bench.FIRST_LOSS[(3, 368, 496, 8, 12)]
    assert d["pin_failed"] is True and d["first_loss_ok"] is False and abs(d["first_loss_pinned"] - 1.5 * pin) < 1e-6 * pin
    assert abs(d["first_loss"] - pin) < 5e-3 * pin and d["value"] > 0 and "WARNING" in r.stderr


def test_rccl_world1_training_and_inference_lines(device):
    """RCCL executes on a one-GPU box (VERDICT r3 missing #1: every multi-rank run so far was gloo): CRAFT_FORCE_COLLECTIVES=1 makes
    bench.py initialise a ONE-rank "nccl" group and the training step run its collectives instead of short-circuiting them -- the
    communicator set-up, the in-place all-reduce of the flat gradient buffer in device memory, the replica broadcasts at Trainer
    construction and the timing protocol's barrier / MAX / SUM reductions are RCCL kernels on this GPU.  Over one rank they are the
    identity, so the line must equal the plain run's."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CRAFT_BENCH_BACKEND")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    args = ["--train", "3", "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "1", "--height", "128", "--width", "160", "--iters", "2",
            "--no-cpu-baseline"]
    lines = {}
    for force in ("0", "1"):
        r = subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, env=dict(env, CRAFT_FORCE_COLLECTIVES=force), capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines[force] = _json_line(r.stdout)
    plain, rccl = lines["0"], lines["1"]
    assert plain["allreduce_ms_per_step"] is None and rccl["allreduce_ms_per_step"] is not None and rccl["allreduce_ms_per_step"] > 0
    assert rccl["n_gpus"] == 1 and rccl["first_loss"] == plain["first_loss"] and rccl["skipped_steps"] == 0
    # (the loss after two updates differs run to run in the 5th digit: atomic accumulation order in the weight gradients)
    assert abs(rccl["loss"] - plain["loss"]) < 5e-4 * plain["loss"]
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "1", "--height", "128", "--width", "256",
                        "--iters", "2", "--no-train-leg", "--no-cpu-baseline"], cwd=ROOT, env=dict(env, CRAFT_FORCE_COLLECTIVES="1"), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 0 and abs(d["value"] - d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) / d["value"] < 0.02


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


@pytest.mark.parametrize("cfg", [3, 4])
def test_bench_training_workload_is_the_pinned_one(device, cfg):
    """bench.py's training legs assert their first-step loss against bench.FIRST_LOSS (`loss == loss` said nothing
    about WHAT was trained).  Here: (a) the constant exists for the default workload and a fresh `bench.py --train cfg` run satisfies
    its own assertion and prints that loss; (b) the same weights and pairs with dropout off: the HIP training forward's loss equals
    the CPU oracle's (batch 8 with BatchNorm batch statistics at configs[3] -- the largest oracle-checked training batch); (c) the pinned
    dropout-ON loss equals the oracle's when it is fed that pass's six masks."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from craft_amd import CRAFT, autograd as AG, default_args
    from craft_amd.synth import synth_pair, synth_state_dict
    from oracle import craft_oracle as O
    H, W, B, policy, _ = bench.TRAIN_CFG[cfg]
    pinned = bench.FIRST_LOSS[(cfg, H, W, B, 12)]
    r = subprocess.run([sys.executable, "bench.py", "--train", str(cfg), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--precision",
                        policy], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert abs(d["first_loss"] - pinned) < 5e-3 * pinned and d["skipped_steps"] == 0 and d["first_loss_ok"] is True
    # (b) dropout off: HIP vs oracle on bench's weights (seed 1234) and pairs (seed 100)
    model = CRAFT(default_args(hip_precision=policy, dropout_prob=0.0, hip_loss_scaled=True))
    sd0 = synth_state_dict(model.state_dict(), seed=1234)
    model.load_state_dict(sd0, strict=True)
    model = model.to(device).train()
    if cfg != 3:
        model.freeze_bn()
    im1, im2, flow = synth_pair(B, H, W, seed=100)
    valid = torch.ones(B, H, W)
    preds = model(im1.to(device), im2.to(device), iters=12)
    loss, _ = AG.sequence_loss(preds, flow, valid, 0.8)
    sd = {k: v.clone() for k, v in sd0.items()}
    sd["corr_fn.setrans.key.weight"], sd["corr_fn.setrans.key.bias"] = sd["corr_fn.setrans.query.weight"], sd["corr_fn.setrans.query.bias"]
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        preds_r, _ = O.craft_train_forward(sd, O.OracleConfig(), im1, im2, iters=12, freeze_bn=cfg != 3)
        loss_r, _ = O.sequence_loss(preds_r, flow, valid, 0.8)
    tol = 1e-4 if policy == "mixed" else 3e-4          # (configs[4] runs bf16 attention operands)
    assert float(loss.detach()) == pytest.approx(float(loss_r), rel=tol)
    worst = max((a.detach().cpu() - b).abs().max().item() for a, b in zip(preds, preds_r))
    print(f"[bench workload] configs[{cfg}] B={B} {H}x{W} T=12 {policy}: dropout-off loss {float(loss.detach()):.6f} vs oracle {float(loss_r):.6f}; "
          f"max prediction error {worst:.2e} px; pinned first-step loss (dropout on) {pinned}")
    assert worst < (1e-2 if policy == "mixed" else 0.1)
    # (c) the pinned constant itself -- the leg's first step WITH dropout on -- is what the oracle computes when it is fed that pass's masks:
    # the leg seeds torch with 1234 + rank (bench.py), the fresh model's first pass has number 0, and the six masks follow from the seed
    # alone (tests/dropout_hash.py: the kernels' hash restated; tests/test_train_dropout_parity.py holds gradients to the same construction)
    from dropout_hash import pass_base, pass_masks
    O.DROPOUT_MASKS = pass_masks(pass_base(1234), B, (H // 8) * (W // 8))
    try:
        with torch.no_grad():
            preds_d, _ = O.craft_train_forward(sd, O.OracleConfig(), im1, im2, iters=12, freeze_bn=cfg != 3)
            loss_d, _ = O.sequence_loss(preds_d, flow, valid, 0.8)
    finally:
        O.DROPOUT_MASKS = None
    print(f"[bench workload] configs[{cfg}]: first-step loss with dropout on: bench {d['first_loss']:.4f}, pinned {pinned}, oracle + masks {float(loss_d):.4f}")
    assert float(loss_d) == pytest.approx(d["first_loss"], rel=5e-4 if policy == "mixed" else 2e-3)
    assert float(loss_d) == pytest.approx(pinned, rel=1e-3 if policy == "mixed" else 3e-3)
