#!/usr/bin/env python3
"""Headline benchmark: image-pairs/sec at 448x1024, 12 refinement iterations (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision bf16|fp16|fp32] [--batch 4]

A "step" is one forward pass of craft_amd.CRAFT (CNN encoders and the hot path both on the HIP kernels of
libcraft_hip.so) over one batch of synthetic 448x1024 pairs already resident in HBM, test_mode=1, 12 iterations —
BASELINE.json configs[1].  N>1: one rank per GPU over RCCL.  Either the caller starts the ranks
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: RANK / WORLD_SIZE in the environment), or
plain `python bench.py --gpus N` re-executes itself under torch.distributed.run with N ranks (and refuses to run when
fewer than N GPUs are visible).  Pairs shard by batch with no data-path collective (weak scaling), the timed region is
bracketed by barrier + synchronize and the max over ranks is reported.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      the dominant kernel by time, k_pv16 (attention apply O = P.V of the motion aggregator, ~19 % of a
                forward), timed live with HIP events on the launch stream: algorithmic bytes = the attention
                probabilities it must stream (B*M*N*ldp*sizeof(P)) + V^T + O, against the 8 TB/s HBM peak
                (MI355X_MICROARCH.md).
  roofline_conv the conv engine IN SITU: HIP events around every craft_sepconv_gru_step call of real forward passes (the four
                k_conv_halo_wf launches of a SepConvGRU update: z|r and q convolutions of both passes, gates in the epilogues):
                algorithmic flops / time against the dense MFMA peak; standalone_*: the z|r convolution alone in a 20-launch loop.
  infer_amp_fp16 the same forward in the reference's own default arithmetic (fp16 operands everywhere, evaluate.py:1455) with its
                end-point deviation from the fp32-class headline policy; reported beside the headline, never instead of it.
  cpu_baseline  the CPU oracle (oracle/craft_oracle.py, fp32 torch-CPU restatement of the reference's
                forward) timed on this box's host cores on ONE 448x1024 pair, 12 iterations.
  train_cfg3    a short leg of BASELINE.json configs[3] (training step at 368x496, batch 8/GPU: 6 warm-up + 5 timed steps, same
                barrier / max-over-ranks protocol; 6 warm-up steps): ms_per_step, pairs_per_s and the roofline of the backward's dominant kernel
                (the packed weight-gradient kernel k_gemm_pk, timed live), and `amp_fp16`: the same step under the reference's own
                --mixed_precision arithmetic (fp16 operands + loss scaling; reported beside the fp32-class headline, not instead of it).
  train_cfg4    the same short leg for configs[4] (368x768 crops, batch 4/GPU, frozen BatchNorm, bf16 MFMA attention).
  corr_cfg2     BASELINE.json configs[2]: correlation build + radius-4 lookup at 768x1024, HIP-event timed, against the HBM peak
                with SURVEY 8(d)'s bytes (rank 0 only; tools/bench_corr.py is the standalone form).
                `python bench.py --train 3|4` is the full training benchmark (its own JSON line with roofline, amp_fp16 and the CPU
                oracle's training step as cpu_baseline).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # RCCL across ranks needs dmabuf IPC on this driver (must be set before HIP initialises)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="pairs per GPU per step (configs[1]: 4)")
    ap.add_argument("--height", type=int, default=448)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--precision", default="mixed",
                    help="fp32 | bf16 | fp16 or a per-role policy such as score=bf16,pv=fp16,conv=fp32 (craft_amd.hip.Precision)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU baseline (0: min(32, logical CPUs))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph-probe", action="store_true",
                    help="(internal) capture + replay the headline pass once in THIS process and exit 0: bench.py runs it as a child first, so a "
                         "runtime crash inside hipStreamEndCapture costs the line its graph, not its existence")
    ap.add_argument("--no-graph", action="store_true",
                    help="skip the `hipgraph` leg (the headline pass recorded once with CRAFT.capture and replayed per step, timed beside the "
                         "eager headline)")
    ap.add_argument("--no-train-leg", action="store_true",
                    help="skip the short configs[3] training leg (3 warm-up + 5 timed steps) that the default line carries as `train_cfg3`")
    ap.add_argument("--mini", action="store_true",
                    help="launch-path rehearsal: the default line's extra legs at miniature shapes (training legs 128x160, batch 1, 2 iterations, no "
                         "pinned loss; corr_cfg2 at 128x256) -- eight gloo ranks sharing one GPU run the driver's N = 8 command in seconds")
    ap.add_argument("--ops", action="store_true", help="also print a per-operator timing table to stderr")
    ap.add_argument("--torch-encoders", action="store_true",
                    help="--train only: keep the two CNN encoders on PyTorch-ROCm / MIOpen under torch autograd (developer A/B)")
    ap.add_argument("--train", type=int, default=0, choices=[0, 3, 4],
                    help="3 / 4: time the TRAINING step of BASELINE.json configs[3] (FlyingChairs 368x496, batch 8/GPU) / configs[4] "
                         "(Sintel 368x768 crops, batch 4/GPU, bf16 MFMA attention) instead of the inference headline: forward + "
                         "HIP backward + one RCCL gradient all-reduce + clip + fused AdamW per step")
    return ap.parse_args()


def op_table(model, im1, im2, iters):
    """Per-operator wall time via events around each C-ABI call (diagnostics; stderr only)."""
    from craft_amd import hip
    orig = hip.call
    acc = {}

    def timed(name, *a):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        orig(name, *a)
        e.record()
        acc.setdefault(name, []).append((s, e))
    import craft_amd.hip_encoder as enc_mod
    import craft_amd.ops as ops_mod
    import craft_amd.update as upd_mod
    hip.call = ops_mod.call = upd_mod.call = enc_mod.call = timed
    try:
        with torch.no_grad():
            model(im1, im2, iters=iters, test_mode=1)
        torch.cuda.synchronize()
    finally:
        hip.call = ops_mod.call = upd_mod.call = enc_mod.call = orig
    rows = [(n, len(v), sum(s.elapsed_time(e) for s, e in v)) for n, v in acc.items()]
    tot = sum(r[2] for r in rows)
    print(f"[ops] per-operator time for one forward (B={im1.shape[0]}): total {tot:.2f} ms", file=sys.stderr)
    for n, c, t in sorted(rows, key=lambda r: -r[2]):
        print(f"[ops] {n:28s} calls {c:4d}  {t:9.3f} ms  {100 * t / tot:5.1f} %", file=sys.stderr)


# Every PMC figure on the line comes from THIS directory: re-collected each round on the final tree; tests/test_profiles.py fails when a
# kernel source is newer (by commit time) than the counter file that describes it (VERDICT r5 "next" 9)
PMC_DIR = "r6"


def pmc_lookup(kernel, group, B, H8, W8):
    """This round's committed PMC summary of one kernel (profiles/r6/pmc_kernels.json, tools/pmc_r6.sh + tools/pmc_r6_json.py; PMC_DIR): HBM-side bytes per launch
    (FETCH_SIZE x2 on gfx950 + WRITE_SIZE) and the matrix pipe's busy fraction -- only when the entry was taken at THIS shape for THIS
    kernel instantiation; otherwise None rather than a stale constant."""
    try:
        with open(os.path.join(ROOT, "profiles", PMC_DIR, "pmc_kernels.json")) as fh:
            pmc = json.load(fh)
        sh = pmc.get("shape", {})
        if (sh.get("B"), sh.get("H8"), sh.get("W8")) != (B, H8, W8):
            return None
        for e in pmc["kernels"].get(kernel, []):
            if e.get("group") == group:
                return e
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return None


def roofline_flash(model, im1, im2, iters, prec, forwards=3):
    """The F2 feature transformer's fused attention (k_flash_attn: Q.K^T, online softmax and P.V in one pass, setrans.py:507-557 +
    :364-410) timed LIVE with HIP events around every craft_flash_attention call of real forward passes (the call also enqueues the two
    k_pack_qk launches, ~25 us).  Algorithmic flops = SURVEY 8(d): 2 * (256 + 1024) * N^2 per sample; executed = 2 * (3 * 256 + 1024) * N^2
    under f16x3 scores."""
    import craft_amd.ops as ops_mod
    from craft_amd.hip import PREC_F16X3, pick
    B, _, H, W = im1.shape
    H8, W8 = H // 8, W // 8
    N = H8 * W8
    orig = ops_mod.call
    evs = []

    def timed(name, *a):
        if name != "craft_flash_attention":
            return orig(name, *a)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        orig(name, *a)
        e.record()
        evs.append((s, e))
    ops_mod.call = timed
    try:
        with torch.no_grad():
            for _ in range(forwards):
                model(im1, im2, iters=min(iters, 1), test_mode=1)
        torch.cuda.synchronize()
    finally:
        ops_mod.call = orig
    if not evs:
        return None
    ms = sum(s.elapsed_time(e) for s, e in evs) / len(evs)
    x3 = pick(prec, "score") == PREC_F16X3
    flops = 2.0 * 1280 * N * N * B
    executed = 2.0 * ((3 if x3 else 1) * 256 + 1024) * N * N * B
    ach = flops / (ms * 1e-3) / 1e12
    name = "k_flash_attn2"
    e = pmc_lookup(f"{name}<64, 256, 2>" if x3 else f"{name}<64, 256, 1>", "flash", B, H8, W8)
    return {"bound": "mfma", "kernel": "k_flash_attn2 (F2 feature transformer: scores + online softmax + P.V fused, 1 launch per forward; + 2 k_pack_qk)",
            "achieved": round(ach, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4),
            "executed_frac": round(executed / (ms * 1e-3) / 1e12 / 2500.0, 4), "flops_per_launch": flops, "ms_per_launch": round(ms, 4),
            "launches_timed": len(evs), "mfma_busy": e.get("mfma_busy") if e else None, "traffic": e.get("hbm_bytes_per_launch") if e else None,
            "note": "timed live (HIP events on the launch stream) around craft_flash_attention in real forward passes; mfma_busy = rocprofv3 "
                    "SQ_VALU_MFMA_BUSY_CYCLES per SIMD / GRBM_GUI_ACTIVE per XCD and traffic = FETCH_SIZE x2 + WRITE_SIZE of the same kernel at this "
                    "shape (profiles/r6/pmc_kernels.json; null when not profiled at this shape): busy is a fraction of the cycles the chip actually "
                    "ran, frac is against the 2.4 GHz peak"}


def roofline_pv(model, im1, im2, iters, prec, forwards=3):
    """The aggregator's P.V kernel timed LIVE: HIP events around every craft_attn_apply call of `forwards` real forward
    passes (the kernel runs on the main stream, the events are recorded on it), so the figure is what rocprof's kernel
    trace of the same command reports (profiles/r1/bench_kernel_stats_short.txt)."""
    from craft_amd import hip
    from craft_amd.hip import PREC_F16, PROB_DTYPE, pick
    import craft_amd.ops as ops_mod
    pv = pick(prec, "pv")
    B, _, H, W = im1.shape
    H8, W8 = H // 8, W // 8
    N, M, Dv = H8 * W8, 4, 128
    ldp = (N + 31) // 32 * 32
    orig = ops_mod.call
    evs = []

    def timed(name, *a):
        if name != "craft_attn_apply":
            return orig(name, *a)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        orig(name, *a)
        e.record()
        evs.append((s, e))
    ops_mod.call = timed
    try:
        with torch.no_grad():
            for _ in range(forwards):
                model(im1, im2, iters=iters, test_mode=1)
        torch.cuda.synchronize()
    finally:
        ops_mod.call = orig
    if not evs:
        return {"bound": "hbm", "kernel": "k_pv16", "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None,
                "note": "no refinement iteration ran: the kernel was not launched"}
    ms = sum(s.elapsed_time(e) for s, e in evs) / len(evs)
    esz = torch.empty(0, dtype=PROB_DTYPE[pv]).element_size()
    bytes_alg = B * M * N * ldp * esz + B * M * Dv * ldp * esz + B * M * N * Dv * 4
    ach = bytes_alg / (ms * 1e-3) / 1e9
    # HBM bytes per launch from the committed PMC passes of this kernel (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in
    # separate runs, gfx950 x2 correction on FETCH_SIZE: profiles/<PMC_DIR>/pmc_kernels.json).  Quoted only when the live launch is
    # the profiled one: same shape, same element type AND the same kernel instantiation (rows per block chosen by the launcher's
    # cost function, replicated here) -- otherwise null rather than a stale constant.
    traffic = None
    try:
        best, best_wr, best_cost = 4, 1, None      # launch_pv16 (kernels_gemm.hip), replicated: the 4-wave pick, then the 8-wave kernel if it fills the chip
        for mt in (4, 5, 6, 7):
            blocks = ((N + 32 * mt - 1) // (32 * mt)) * (Dv // 128) * B * M
            slots = 256 * (3 if mt == 4 else 2)
            cost = ((blocks + slots - 1) // slots) * mt
            if best_cost is None or cost < best_cost:
                best, best_cost = mt, cost
        if not os.environ.get("CRAFT_PV_NO_WR2"):
            rows_cu1, best2, rows_cu2 = best_cost * (3 if best == 4 else 2), 0, None
            for mt in (4, 5, 6, 7):
                blocks = ((N + 64 * mt - 1) // (64 * mt)) * (Dv // 128) * B * M
                if blocks < 256:
                    continue
                c = ((blocks + 255) // 256) * 2 * mt
                if rows_cu2 is None or c < rows_cu2:
                    best2, rows_cu2 = mt, c
            if best2 and rows_cu2 <= rows_cu1:
                best, best_wr = best2, 2
        live = f"k_pv16<{pv}, {best}, {best_wr}>"
        e5 = pmc_lookup(live, "pv", B, H8, W8) if pv == PREC_F16 else None      # this round's passes of the SAME instantiation, else null
        if e5:
            traffic = int(e5["hbm_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        pass
    return {"bound": "hbm", "kernel": f"k_pv16 (attention apply O = P.V of the motion aggregator, {iters} launches per forward)",
            "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": traffic,
            "bytes_per_launch": bytes_alg, "ms_per_launch": round(ms, 4), "launches_timed": len(evs),
            "note": "timed live around every launch of real forward passes (HIP events on the launch stream); algorithmic bytes "
                    "= P (fp16, read once) + V^T + O; traffic = HBM bytes per launch from the PMC passes committed under "
                    "profiles/r6/pmc_kernels.json (FETCH_SIZE x2 on gfx950 + WRITE_SIZE; null when the live kernel instantiation is not the profiled one); measured read-only ceiling of this access pattern "
                    "on the same chip: 6.1 TB/s plain / 6.9 TB/s non-temporal with NO arithmetic (tools/ubench/hbm_rows_dma.hip); with the P.V MFMAs on, the "
                    "package sits at its 1400 W limit and sclk falls 2.4 -> 1.58 GHz (profiles/r5/pv16_power.txt): the kernel is bound by the power cap"}


def roofline_conv(B, H8, W8, prec, reps=20):
    """Time the GRU's z|r convolution alone (1x5, 384 -> 256 channels -- [h | motion | aggregated] after the context
    hoist: the largest K loop of the update block, k_conv_halo_wf) with HIP events; algorithmic flops = 2 * pixels *
    Cout * KH*KW*Cin."""
    from craft_amd import ops
    from craft_amd.hip import PREC_F32, pick
    cp = pick(prec, "conv")
    dev = torch.device("cuda")
    N = H8 * W8
    x = torch.randn(B, N, 384, device=dev)
    w = torch.randn(256, 384, 1, 5, device=dev) * 0.02
    bias = torch.zeros(256, device=dev)
    wp = ops.pack_conv_prec(w, cp)
    y = torch.empty(B, N, 256, device=dev)
    for _ in range(3):
        ops.conv2d_tokens(x, (H8, W8), wp, bias, 256, 1, 5, 0, cp, packed=cp != PREC_F32, out=y)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        ops.conv2d_tokens(x, (H8, W8), wp, bias, 256, 1, 5, 0, cp, packed=cp != PREC_F32, out=y)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    flops = 2.0 * B * N * 256 * 5 * 384
    ach = flops / (ms * 1e-3) / 1e12
    peak = 157.3 if cp == PREC_F32 else 2500.0
    from craft_amd.hip import PREC_F16X3
    e = pmc_lookup("k_conv_halo_wf<3, 1, 4, false, 5, 7>", "convtok", B, H8, W8) if cp == PREC_F16X3 else None
    return {"bound": "mfma", "kernel": "k_conv_halo_wf (SepConvGRU z|r conv, 1x5, 384->256, 24 launches per forward)", "achieved": round(ach, 1), "peak": peak,
            "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": e.get("hbm_bytes_per_launch") if e else None,
            "mfma_busy": e.get("mfma_busy") if e else None, "valu_per_mfma": e.get("valu_per_mfma") if e else None,
            "lds_bank_conflict_share": e.get("lds_bank_conflict_share") if e else None, "flops_per_launch": flops,
            "ms_per_launch": round(ms, 4),
            "note": "algorithmic (fp32-equivalent) flops; the f16x3 scheme executes 3 fp16 MFMAs per product, so the "
                    "matrix pipe runs at 3x this rate; peak = dense fp16 MFMA (fp32 MFMA for the fp32 policy)"}


def roofline_conv_live(model, im1, im2, iters, prec, forwards=3):
    """The convolution family IN SITU (VERDICT r5 "next" 9): HIP events on the launch stream around every craft_sepconv_gru_step call of
    real forward passes -- one call = the four halo convolutions of a SepConvGRU update (z|r and q of the 1x5 and the 5x1 pass, 384 ->
    256 / 128 channels after the context hoist, gates fused into the epilogues; update.py:49-64), the largest convolution group of the
    refinement loop.  Algorithmic flops per call = 2 * pixels * (256 + 128) * 5 * 384 * 2 passes; under f16x3 the matrix pipe executes
    3x that."""
    import craft_amd.update as upd_mod
    from craft_amd.hip import PREC_F16X3, PREC_F32, pick
    B, _, H, W = im1.shape
    N = (H // 8) * (W // 8)
    orig = upd_mod.call
    evs = []

    def timed(name, *a):
        if name != "craft_sepconv_gru_step":
            return orig(name, *a)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        orig(name, *a)
        e.record()
        evs.append((s, e))
    upd_mod.call = timed
    try:
        with torch.no_grad():
            for _ in range(forwards):
                model(im1, im2, iters=iters, test_mode=1)
        torch.cuda.synchronize()
    finally:
        upd_mod.call = orig
    if not evs:
        return None
    ms = sum(s.elapsed_time(e) for s, e in evs) / len(evs)
    cp = pick(prec, "conv")
    flops = 2.0 * B * N * (256 + 128) * 5 * 384 * 2
    peak = 157.3 if cp == PREC_F32 else 2500.0
    ach = flops / (ms * 1e-3) / 1e12
    return {"ms_per_call": round(ms, 4), "calls_timed": len(evs), "flops_per_call": flops, "achieved": round(ach, 1),
            "frac": round(ach / peak, 4), "executed_frac": round((3 if cp == PREC_F16X3 else 1) * ach / peak, 4)}


def infer_amp_fp16(H, W, B, iters, steps, warmup, dev, seed):
    """The reference's own default inference arithmetic beside the fp32-class headline (evaluate.py:1455: --mixed_precision is on unless
    --fullprec; network.py:179-199 runs the encoders and the attention under fp16 autocast): policy `train_amp_fp16` = fp16 MFMA operands
    in every contraction, fp32 accumulation and activations.  Same weights, pairs and timing protocol as the headline (one rank);
    `epe_vs_fp32class` = mean / max end-point distance of its flow from the headline policy's flow on the same pairs (the headline is
    held to the oracle at 5e-5 px by tests/test_full_size_parity.py, so this is the policy's deviation from the oracle to that margin)."""
    from craft_amd import CRAFT, default_args
    from craft_amd.synth import synth_pair, synth_state_dict
    out = {}
    flows = {}
    im1, im2, _ = synth_pair(B, H, W, seed=seed)
    im1, im2 = im1.to(dev), im2.to(dev)
    for policy in ("mixed", "train_amp_fp16"):
        model = CRAFT(default_args(hip_precision=policy, mixed_precision=True))
        model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
        model = model.to(dev).eval()
        with torch.no_grad():
            for _ in range(warmup if policy != "mixed" else 1):
                _, up = model(im1, im2, iters=iters, test_mode=1)
            if policy != "mixed":
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    _, up = model(im1, im2, iters=iters, test_mode=1)
                torch.cuda.synchronize()
                out["ms_per_step"] = round(1e3 * (time.perf_counter() - t0) / steps, 3)
                out["pairs_per_s"] = round(B * steps / (time.perf_counter() - t0), 3)
        flows[policy] = up.float()
        del model
    d = (flows["train_amp_fp16"] - flows["mixed"]).pow(2).sum(1).sqrt()
    out.update(policy="train_amp_fp16 (fp16 MFMA operands everywhere, f32 accumulate: the reference's --mixed_precision default, evaluate.py:1455)",
               epe_vs_fp32class_mean=round(float(d.mean()), 5), epe_vs_fp32class_max=round(float(d.max()), 4), finite=bool(torch.isfinite(d).all()),
               steps=steps, warmup=warmup)
    return out


def corr_cfg2(reps=10, H=768, W=1024):
    """BASELINE.json configs[2] on the default line: the 768x1024 correlation build (fused scores + mode pooling + 4-level pyramid +
    statistics) and the radius-4 lookup, timed with HIP events on the launch stream (tools/bench_corr.py).  frac = SURVEY 8(d) bytes
    (the pyramid written once + Q / K read once) / time / 8 TB/s; traffic = FETCH_SIZE x2 + WRITE_SIZE of the build kernel from this
    round's committed PMC passes (profiles/<PMC_DIR>/pmc_corr_build.json; null when absent or taken at another shape)."""
    from tools.bench_corr import measure
    r = measure(H, W, 1, reps, "mixed")
    b, lk = r["corr_build"], r["corr_lookup"]
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", PMC_DIR, "pmc_corr_build.json")) as fh:
            pmc = json.load(fh)
        if pmc.get("shape") == [1, H // 8, W // 8]:
            traffic = int(pmc["hbm_bytes_per_launch"])
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return {"workload": r["workload"], "build_ms": b["ms"], "lookup_ms": lk["ms"], "bytes": b["bytes"],
            "achieved": b["achieved_GBs"], "peak": 8000.0, "unit": "GB/s", "frac": b["frac_of_hbm_peak"], "traffic": traffic,
            "lookup_achieved": lk["achieved_GBs"], "lookup_frac": lk["frac_of_hbm_peak"], "tflops_algorithmic": b["tflops_algorithmic"],
            "bound": "hbm", "kernel": "k_corr_build4t (+ k_split_planes, statistics fill); k_corr_lookup"}


def cpu_baseline(H, W, iters, threads):
    """The CPU oracle on ONE pair of the same workload, in a subprocess with a wall-clock bound (a bounded
    sample: ~10 s of CPU work on 8 cores).  Returns None if it can not finish in time."""
    import subprocess
    threads = threads or min(32, os.cpu_count() or 1)   # measured best on the 2x64-core EPYC host: 16-32 threads
    cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--threads", str(threads), "--height", str(H),
           "--width", str(W), "--iters", str(iters)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(f"[bench] cpu baseline failed: {e!r}", file=sys.stderr)
        return None


def free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run
    (127.0.0.1 rendezvous on a free port), pass their output through, return the launcher's exit code."""
    import subprocess
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL across processes needs dmabuf IPC on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


# configs[3] runs the library's default policy "mixed" -- the policy of the inference headline: f16x3 (fp32-class) operands for projections,
# scores and every convolution, fp16 operands for the attention products P.V / dV / dP (legal in training under the Trainer's loss scale);
# gradient parity at this shape and depth: tests/test_cfg_step_parity.py::test_training_step_at_configs3_size_against_oracle[...mixed].
# `--precision train_f16x3` times the all-f16x3 policy (+8 % step time), the `amp_fp16` sub-leg the reference's --mixed_precision arithmetic.
TRAIN_CFG = {3: (368, 496, 8, "mixed", "configs[3]: FlyingChairs-size 368x496, batch 8/GPU"),
             4: (368, 768, 4, "train_bf16attn", "configs[4]: Sintel-crop 368x768, batch 4/GPU, bf16 MFMA attention")}


# first-step loss of the training legs by (cfg, H, W, B, iters) -- measured on the MI355X, recomputed by tests/test_bench_contract.py
FIRST_LOSS = {(3, 368, 496, 8, 12): 191.2346, (4, 368, 768, 4, 12): 76.5932}


def roofline_wgrad(step, policy, steps=2):
    """The dominant kernel family of the backward pass, the convolution weight gradients (per tap a cout x cin product over K = all
    pixels: k_gemm_pk on packed operands, craft_wgrad_pk; k_conv_wgrad in the fp32 policy), timed LIVE: HIP events around every launch
    of `steps` real training steps (events on the launch stream).  Algorithmic flops per launch = 2 * pixels * calls * cout * cin *
    KH * KW from the launch's own arguments.  Reported: the launch shape with the largest total time (`kernel`, per-launch numbers)
    and the aggregate over all weight-gradient launches."""
    import craft_amd.autograd as ag_mod
    import craft_amd.train_encoder as te_mod
    orig_call, orig_pk = ag_mod.call, ag_mod.wgrad_pk
    evs = []

    def ev_pair():
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(name, *args):                    # round-2 kernel (fp32 policy, odd channel counts): one launch per call
        if name != "craft_conv2d_wgrad":
            return orig_call(name, *args)
        s, e = ev_pair()
        s.record()
        orig_call(name, *args)
        e.record()
        # (x, ldx, cin, dy, ldy, cout, KH, KW, B, H, W, dW, db, ws, ws_floats, prec)
        evs.append((("k_conv_wgrad", args[2], args[5], args[6], args[7], args[8] * args[9] * args[10], 1), s, e))

    def timed_pk(pairs, KH, KW, acc):          # packed-operand kernel: ONE launch over the calls of a layer in the pass
        pairs = [pairs] if isinstance(pairs, tuple) else pairs
        s, e = ev_pair()
        s.record()
        orig_pk(pairs, KH, KW, acc)
        e.record()
        gp, xp = pairs[0]
        evs.append((("k_gemm_pk", xp.C, gp.C, KH, KW, gp.rows, len(pairs)), s, e))
    ag_mod.call, ag_mod.wgrad_pk = timed, timed_pk
    te_orig = getattr(te_mod, "call", None)
    if te_orig is not None:
        te_mod.call = timed
    try:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    finally:
        ag_mod.call, ag_mod.wgrad_pk = orig_call, orig_pk
        if te_orig is not None:
            te_mod.call = te_orig
    if not evs:
        return None
    by = {}
    for key, s, e in evs:
        by.setdefault(key, []).append(s.elapsed_time(e))
    flop = lambda k: 2.0 * k[5] * k[6] * k[1] * k[2] * k[3] * k[4]      # noqa: E731   (2 * pixels * calls * cin * cout * taps)
    tot_ms = sum(sum(v) for v in by.values())
    tot_fl = sum(flop(k) * len(v) for k, v in by.items())
    key = max(by, key=lambda k: sum(by[k]))
    ms = sum(by[key]) / len(by[key])
    ach = flop(key) / (ms * 1e-3) / 1e12
    from craft_amd.hip import PREC_F16, PREC_F16X3, Precision
    pol = Precision.parse(policy)
    # MFMAs issued per product of a weight gradient: 3 for f16x3 operands, 2 with the activations as one fp16 plane (role wgx), 1 with dY too (wgy: "mixed")
    mult = (1 if pol.wgy == PREC_F16 else 2 if pol.wgx == PREC_F16 else 3) if pol.conv == PREC_F16X3 else 1
    traffic = None
    kern, cin, cout, KH, KW, rows, calls = key
    try:        # HBM bytes per launch from this round's PMC passes -- quoted only for the launch shape and operand mode profiled
        with open(os.path.join(ROOT, "profiles", PMC_DIR, "pmc_traffic_wgrad.json")) as fh:
            pmc = json.load(fh)
        if list(pmc.get("shape", [])) == list(key[1:]) and pmc.get("kernel") == kern and pmc.get("mfmas_per_product", 3) == mult:
            traffic = int(pmc["hbm_bytes_per_launch"])
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return {"bound": "mfma", "kernel": f"{kern} (weight gradient of the {KH}x{KW} convolution {cin}->{cout} over {rows} pixels x {calls} "
                                       f"call(s) of the layer per launch: {len(by[key]) // steps} launch(es) per step, the launch shape with "
                                       "the largest total time; operand packing not included)",
            "achieved": round(ach, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4), "traffic": traffic,
            "flops_per_launch": flop(key), "ms_per_launch": round(ms, 4), "launches_timed": len(by[key]),
            "executed_frac": round(mult * ach / 2500.0, 4),
            "all_wgrad_launches": {"launches_per_step": len(evs) // steps, "ms_per_step": round(tot_ms / steps, 3),
                                   "achieved": round(tot_fl / (tot_ms * 1e-3) / 1e12, 1), "frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / 2500.0, 4)},
            "mfmas_per_product": mult,
            "note": "algorithmic (fp32-equivalent) flops against the dense fp16 MFMA peak; executed_frac counts the MFMAs issued "
                    f"({mult} per product in this policy); timed live with HIP events around every launch of real training steps"}


def cpu_baseline_train(H, W, iters, threads, freeze_bn):
    """The CPU oracle's training step (craft_train_forward + sequence loss + torch autograd) on ONE pair, in a subprocess."""
    import subprocess
    threads = threads or min(32, os.cpu_count() or 1)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--train", "--threads", str(threads), "--height", str(H),
           "--width", str(W), "--iters", str(iters)] + (["--freeze-bn"] if freeze_bn else [])
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(f"[bench] cpu training baseline failed: {e!r}", file=sys.stderr)
        return None


def train_leg(cfg, rank, world, dev, steps, warmup, iters, B=None, H=None, W=None, policy=None, torch_encoders=False, roofline=True):
    """Time whole training steps of BASELINE.json configs[cfg] on this rank (every rank calls it: the step contains the
    gradient all-reduce).  -> dict (rank-0 view: aggregate pairs/s over all ranks, max-over-ranks time)."""
    from craft_amd import CRAFT, default_args
    from craft_amd.dist import aggregate_throughput, timed_steps
    from craft_amd.synth import synth_pair, synth_state_dict
    from craft_amd.train import Trainer
    H0, W0, B0, pol0, name = TRAIN_CFG[cfg]
    H, W, B, policy = H or H0, W or W0, B or B0, policy or pol0
    # the dropout masks are a counter-based hash seeded from torch.initial_seed() (train_forward.forward_train), which this PyTorch build
    # draws at random per process: seed it like the reference's trainers do (train.py:407: torch.manual_seed(1234)) so that the leg is
    # the same workload in every run -- its first-step loss is pinned below
    torch.manual_seed(1234 + rank)
    model = CRAFT(default_args(hip_precision=policy, hip_encoders=not torch_encoders))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
    model = model.to(dev)
    tr = Trainer(model, lr=4e-4 if cfg == 3 else 1.25e-4, wdecay=1e-4 if cfg == 3 else 1e-5, num_steps=100000, iters=iters,
                 clip=1.0, freeze_bn=cfg != 3)
    im1, im2, flow = synth_pair(B, H, W, seed=100 + rank)
    im1, im2, flow = im1.to(dev), im2.to(dev), flow.to(dev)
    valid = torch.ones(B, H, W, device=dev)
    last = {}

    def step():
        last["m"] = tr.step(im1, im2, flow, valid)
        # (this rank's own loss: with N > 1 ranks "loss" is the mean over the ranks, which see different pairs)
        last.setdefault("first", last["m"].get("loss_rank", last["m"]["loss"]))

    dt_rank = timed_steps(step, steps=steps, warmup=warmup, sync=torch.cuda.synchronize)
    value, dt = aggregate_throughput(pairs_per_rank_step=B, steps=steps, dt=dt_rank)
    finite = last["m"]["loss"] == last["m"]["loss"]
    # the first step's loss (synthetic weights seed 1234, pairs seed 100, dropout hash seeded by torch.manual_seed(1234 + rank) above) is a constant of
    # the workload: tests/test_bench_contract.py::test_bench_training_workload_is_the_pinned_one recomputes it, and holds the same batch
    # with dropout off to the CPU oracle's loss -- a leg that trains something else (other weights, shape, iterations) fails here; the
    # step WITH dropout on is held to the oracle (same masks) at these shapes by tests/test_train_dropout_parity.py
    pinned = FIRST_LOSS.get((cfg, H, W, B, iters)) if rank == 0 and not torch_encoders else None
    if pinned is not None and os.environ.get("CRAFT_BENCH_PIN_SCALE"):      # (test hook: a deliberately wrong pin, tests/test_bench_contract.py)
        pinned *= float(os.environ["CRAFT_BENCH_PIN_SCALE"])
    # REPORTED, not asserted (round 5): a drifting constant -- another GPU / ROCm, a legitimate rounding change -- must not cost the run its
    # headline and every other leg.  The line carries first_loss / first_loss_pinned / first_loss_ok, main() sets "pin_failed" and exits
    # non-zero AFTER the JSON is out; the hard assertion lives in tests/test_bench_contract.py.
    pin_ok = None
    if pinned is not None and warmup + steps > 0:
        pin_ok = bool(finite and abs(last["first"] - pinned) < 5e-3 * pinned)
        if not pin_ok:
            print(f"[bench] WARNING configs[{cfg}] first-step loss {last['first']:.5f}, pinned {pinned:.5f}: the training leg does not train "
                  "the pinned workload (or the arithmetic changed)", file=sys.stderr)
    if not finite:
        print(f"[bench] WARNING configs[{cfg}] {policy}: non-finite loss after {warmup + steps} steps", file=sys.stderr)
    if last["m"].get("skipped_steps"):
        print(f"[bench] configs[{cfg}] {policy}: {last['m']['skipped_steps']} step(s) skipped on gradient overflow (loss scale now "
              f"{last['m']['loss_scale']:g})", file=sys.stderr)
    out = {"H": H, "W": W, "B": B, "policy": policy, "name": name, "value": value, "dt": dt, "steps": steps, "warmup": warmup,
           "loss": last["m"]["loss"], "first_loss": last.get("first"), "first_loss_pinned": pinned, "first_loss_ok": pin_ok, "finite": bool(finite), "skipped_steps": last["m"].get("skipped_steps", 0), "numel": tr.optimizer.numel, "freeze_bn": cfg != 3,
           "allreduce_ms": tr.allreduce_ms(), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2)}
    if roofline:
        out["roofline"] = roofline_wgrad(step, policy)      # every rank runs the extra steps (they contain the collective)
    del tr, model
    return out


AMP_NOTE = ("the same step in the reference's own arithmetic: every shipped training script passes --mixed_precision (train.py:215,231-238: fp16 "
            "autocast + GradScaler); here fp16 MFMA operands in every contraction, fp32 accumulation / activations / master weights, the "
            "Trainer's loss scale with skip-on-overflow.  Reported beside the headline, which stays on the fp32-class policy")


def amp_leg(cfg, rank, world, dev, iters, **kw):
    """The training step of configs[cfg] under policy train_amp_fp16 (5 timed steps) -> a compact dict for the bench line."""
    r = train_leg(cfg, rank, world, dev, steps=5, warmup=6, iters=iters, policy="train_amp_fp16", roofline=False, **kw)
    return {"policy": "train_amp_fp16", "ms_per_step": round(1e3 * r["dt"] / r["steps"], 3), "pairs_per_s": round(r["value"], 3),
            "steps": r["steps"], "warmup": r["warmup"], "loss": round(r["loss"], 4), "note": AMP_NOTE}


def train_bench(a, rank, world, dev, dist):
    """BASELINE.json configs[3] / configs[4]: whole training steps (train.py:215-236 / train_ddp.py:230-262) on synthetic
    pairs resident in HBM.  One rank per GPU, full replica, its own pairs; the one data-path collective is the all-reduce of
    the flat gradient buffer (RCCL over xGMI)."""
    argv = " ".join(sys.argv[1:])
    r = train_leg(a.train, rank, world, dev, a.steps, a.warmup, a.iters, B=a.batch if "--batch" in argv else None,
                  H=a.height if "--height" in argv else None, W=a.width if "--width" in argv else None,
                  policy=a.precision if "--precision" in argv else None, torch_encoders=a.torch_encoders)
    H, W, B, policy, name = r["H"], r["W"], r["B"], r["policy"], r["name"]
    value, dt = r["value"], r["dt"]
    amp = None
    if "--precision" not in argv:
        amp = amp_leg(a.train, rank, world, dev, a.iters, B=B, H=H, W=W, torch_encoders=a.torch_encoders)
    if rank == 0:
        line = {
            "metric": f"training image-pairs/sec at {H}x{W}, {a.iters} iters (forward + backward + gradient all-reduce + AdamW)",
            "value": round(value, 3), "unit": "image-pairs/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak",
            # BASELINE.md holds no published number for this metric (its section 1 only derives 20.6 / 10.5 pairs/s from the ETA
            # column of the reference's logs: whole job on 2 unnamed GPUs, DataParallel + AMP) -> null, the derived figure as a note
            "vs_baseline": None,
            "baseline_note": f"not a published number: BASELINE.md section 1 derives {20.6 if a.train == 3 else 10.5} pairs/s from the ETA "
                             "column of the reference's own logs (whole job, 2 unnamed GPUs, DataParallel + AMP"
                             + (")" if a.train == 3 else ", batch 6 there vs 4 per GPU here)"),
            "dtype": policy + " (fp32 activations / probabilities in HBM; MFMA operand mode per role, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": name + f", {a.iters} iters, model.train(): dropout 0.1 / 0.2, "
                                          + ("BatchNorm batch statistics" if a.train == 3 else "frozen BatchNorm")
                                          + ", synthetic weights and pairs", "global_batch": B * world,
                       "parallelism": f"dp{world} (one all-reduce of the {r['numel'] * 4 / 1e6:.1f} MB flat gradient per step)"},
            "loss": round(r["loss"], 4), "first_loss": round(r["first_loss"], 4), "first_loss_pinned": r["first_loss_pinned"],
            "first_loss_ok": r["first_loss_ok"], "skipped_steps": r["skipped_steps"],
            "peak_mem_GB": r["peak_mem_GB"], "allreduce_ms_per_step": r["allreduce_ms"],
            "roofline": r.get("roofline"), "amp_fp16": amp}
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_train(H, W, a.iters, a.cpu_threads, r["freeze_bn"])
        if r["first_loss_ok"] is False or not r["finite"]:
            line["pin_failed"] = True
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    return 3 if (r["first_loss_ok"] is False or not r["finite"]) else 0       # (rank 0 checks the pin; non-zero only after the JSON is out)


def graph_probe_child(a, local_dev):
    """Run `bench.py --graph-probe` for this workload on this rank's device in a CHILD process: (ok, note).  A capture that takes the
    process down (a segmentation fault inside hipStreamEndCapture was seen once this round, with the refinement loop sliced over two
    streams) must cost the headline its graph, not the line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                                                           "LOCAL_WORLD_SIZE", "ROLE_RANK", "TORCHELASTIC_RUN_ID", "CRAFT_FORCE_COLLECTIVES")}
    env["CRAFT_PROBE_DEVICE"] = str(local_dev)
    cmd = [sys.executable, os.path.abspath(__file__), "--graph-probe", "--batch", str(a.batch), "--height", str(a.height), "--width", str(a.width),
           "--iters", str(a.iters), "--precision", a.precision]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)      # (normally ~8 s)
    except subprocess.TimeoutExpired:
        return False, "graph probe timed out"
    if r.returncode == 0 and "graph-probe ok" in r.stdout:
        return True, ""
    return False, f"graph probe exit code {r.returncode}: {(r.stderr or '').strip().splitlines()[-1:] or ''}"[:300]


def main():
    a = parse()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "RANK" not in os.environ and a.gpus > 1:
        shared = os.environ.get("CRAFT_BENCH_BACKEND", "nccl") != "nccl"     # gloo: ranks may share a device (tests)
        if not shared and torch.cuda.device_count() < a.gpus:
            raise SystemExit(f"bench.py --gpus {a.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
        raise SystemExit(self_launch(a.gpus))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # one rank per GPU; CRAFT_BENCH_BACKEND=gloo lets several ranks share a device to exercise this path on a 1-GPU box
    backend = os.environ.get("CRAFT_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if world > 1 and backend == "nccl" and local >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {ndev} GPU(s) visible")
    local_dev = local % ndev
    if a.graph_probe:
        local_dev = int(os.environ.get("CRAFT_PROBE_DEVICE", local_dev)) % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    force1 = world == 1 and os.environ.get("CRAFT_FORCE_COLLECTIVES", "0") not in ("", "0")
    if force1:                            # a one-rank group: every collective of the N > 1 path runs (RCCL on a one-GPU box), see dist.py
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force1:
        import torch.distributed as dist
        dist.init_process_group(backend, init_method="env://")

    from craft_amd import CRAFT, default_args
    from craft_amd.hip import Precision
    from craft_amd.synth import synth_pair, synth_state_dict

    if a.train:
        return train_bench(a, rank, world, dev, dist)
    graph_ok, graph_note = (False, "") if a.no_graph or a.graph_probe else graph_probe_child(a, local_dev)
    if dist and not a.graph_probe:            # the leg's timing protocol is collective: every rank runs it, or none does
        t_ok = torch.tensor([1.0 if graph_ok else 0.0], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
        if graph_ok and t_ok.item() < 1.0:
            graph_ok, graph_note = False, "graph probe failed on another rank"
    prec = Precision.parse(a.precision)
    model = CRAFT(default_args(hip_precision=a.precision))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
    model = model.to(dev).eval()
    im1, im2, _ = synth_pair(a.batch, a.height, a.width, seed=100 + rank)
    im1, im2 = im1.to(dev), im2.to(dev)

    def step():
        with torch.no_grad():
            return model(im1, im2, iters=a.iters, test_mode=1)

    from craft_amd.dist import aggregate_throughput, timed_steps
    last = {}
    failures = []                    # legs that failed: the line is printed anyway, the exit code says so afterwards

    def run_step():
        last["out"] = step()

    if a.graph_probe:
        step()
        g = model.capture(im1, im2, iters=a.iters, test_mode=1)
        g(im1, im2)
        torch.cuda.synchronize()
        print("graph-probe ok", flush=True)
        return 0

    # The headline: eager launches (~280 kernels on three streams per pass), the protocol of every earlier round.
    dt_rank = timed_steps(run_step, steps=a.steps, warmup=a.warmup, sync=torch.cuda.synchronize)
    value, dt = aggregate_throughput(pairs_per_rank_step=a.batch, steps=a.steps, dt=dt_rank)

    # Beside it: the same pass recorded once as a hipGraph (CRAFT.capture: same kernels, same streams, bit-identical results --
    # tests/test_graphed_forward.py) and replayed per step; a step copies the batch into the graph's input buffers and replays.  Timed with
    # the same protocol right after the headline; box to box it is between 2.3 % slower and 2 % faster than the eager sequence (it runs second, on a warmed-up chip)
    # (profiles/r6/graph_ab.txt).  The
    # capture runs in a child process first (graph_probe_child): a crash inside the runtime must not cost the line.
    graph_leg = None
    if not a.no_graph and not graph_ok:
        print(f"[bench] WARNING {graph_note}: no hipGraph leg", file=sys.stderr)
        graph_leg = {"skipped": graph_note}
    if graph_ok:
        try:
            graphed = model.capture(im1, im2, iters=a.iters, test_mode=1)
            eager_up = last["out"][1].clone()
            box = {}

            def run_graphed():
                box["out"] = graphed(im1, im2)

            gv, gdt = aggregate_throughput(pairs_per_rank_step=a.batch, steps=a.steps,
                                           dt=timed_steps(run_graphed, steps=a.steps, warmup=a.warmup, sync=torch.cuda.synchronize))
            d_graph = float((box["out"][1] - eager_up).abs().max())
            graph_leg = {"launch": "one hipGraph replay per step (CRAFT.capture; inputs copied into the graph's buffers inside the timed region)",
                         "ms_per_step": round(1e3 * gdt / a.steps, 3), "pairs_per_s": round(gv, 3), "steps": a.steps, "warmup": a.warmup,
                         "max_abs_px_vs_eager": d_graph}
            if not d_graph < 1e-4:
                failures.append(f"configs[1]: hipGraph replay deviates from the eager pass by {d_graph} px")
            del graphed, box
        except Exception as e:      # noqa: BLE001  (every rank falls back the same way: capture is deterministic)
            failures.append(f"graph leg: {type(e).__name__}: {e}"[:300])
            print(f"[bench] WARNING hipGraph leg failed: {e}", file=sys.stderr)
    if not bool(torch.isfinite(last["out"][1]).all()):
        failures.append("configs[1]: non-finite flow")

    # the short training leg of the default line (every rank runs it: a training step contains the gradient all-reduce).  A leg that
    # raises (a deterministic error hits every rank at the same point) is recorded and the rest of the line still goes out.
    tl = tl4 = None
    if not a.no_train_leg:
        del last["out"]
        torch.cuda.empty_cache()
        try:
            mini = dict(B=1, H=128, W=160) if a.mini else {}
            it_ = 2 if a.mini else 12
            tl = train_leg(3, rank, world, dev, steps=5, warmup=6, iters=it_, **mini)      # (6 warm-up steps: the caching allocator's pool settles after ~5)
            tl["amp"] = amp_leg(3, rank, world, dev, it_, **mini)
        except Exception as e:      # noqa: BLE001
            failures.append(f"train_cfg3: {type(e).__name__}: {e}"[:300])
            print(f"[bench] WARNING training leg configs[3] failed: {e}", file=sys.stderr)
        torch.cuda.empty_cache()
        try:
            tl4 = train_leg(4, rank, world, dev, steps=5, warmup=6, iters=it_, roofline=False, **mini)
        except Exception as e:      # noqa: BLE001
            failures.append(f"train_cfg4: {type(e).__name__}: {e}"[:300])
            print(f"[bench] WARNING training leg configs[4] failed: {e}", file=sys.stderr)
    for t_ in (tl, tl4):
        if t_ is not None and (t_["first_loss_ok"] is False or not t_["finite"]):
            failures.append(f"{t_['name']}: first-step loss {t_['first_loss']} vs pinned {t_['first_loss_pinned']} (finite: {t_['finite']})")

    if rank == 0:
        line = {
            "metric": "image-pairs/sec at 448x1024, 12 iters",
            "value": round(value, 3), "unit": "image-pairs/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "mixed": "f16x3 (fp32 operands split into 2 fp16 planes, 3 fp16 MFMAs per product, f32 "
                      "accumulate: fp32-class) for projections / Q.K^T / convolutions; f16 P.V (f32 accumulate)"}.get(a.precision, a.precision),
            "data": "synthetic",
            "config": {"workload": f"configs[1]: {a.height}x{a.width} synthetic pairs, batch {a.batch}/GPU, {a.iters} iters, "
                                   "craft-sintel architecture with synthetic weights (checkpoints absent), test_mode=1",
                       "global_batch": a.batch * world, "parallelism": f"dp{world} (pairs sharded by batch, no collective)",
                       "launch": "eager launches"},
        }
        if graph_leg is not None:
            line["hipgraph"] = graph_leg
        if a.ops:
            op_table(model, im1, im2, a.iters)
        line["roofline"] = roofline_pv(model, im1, im2, a.iters, prec)
        rc = roofline_conv(a.batch, a.height // 8, a.width // 8, prec)
        live = roofline_conv_live(model, im1, im2, a.iters, prec)
        if rc is not None and live is not None:
            # headline figures of the object = the LIVE in-situ measurement; the stand-alone single launch stays beside it
            rc.update(standalone_ms_per_launch=rc["ms_per_launch"], standalone_frac=rc["frac"], standalone_achieved=rc["achieved"],
                      achieved=live["achieved"], frac=live["frac"], executed_frac=live["executed_frac"], ms_per_launch=round(live["ms_per_call"] / 4, 4),
                      ms_per_call=live["ms_per_call"], calls_timed=live["calls_timed"], flops_per_call=live["flops_per_call"],
                      timing="achieved / frac / ms_per_call: HIP events on the launch stream around every craft_sepconv_gru_step call of real "
                             "forward passes (4 k_conv_halo_wf launches per call: z|r and q convolutions of both SepConvGRU passes, gates in the "
                             "epilogues); ms_per_launch = ms_per_call / 4; standalone_*: the z|r convolution alone in a 20-launch loop")
        line["roofline_conv"] = rc
        line["roofline_flash"] = roofline_flash(model, im1, im2, a.iters, prec)
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.height, a.width, a.iters, a.cpu_threads)
        if not a.no_train_leg:          # (the same switch keeps a quick run quick)
            line["corr_cfg2"] = corr_cfg2(3, 128, 256) if a.mini else corr_cfg2()
            if a.precision == "mixed":
                try:
                    line["infer_amp_fp16"] = (infer_amp_fp16(128, 160, 1, 2, 2, 1, dev, 900) if a.mini else
                                              infer_amp_fp16(a.height, a.width, a.batch, a.iters, max(3, a.steps // 2), 2, dev, 900))
                except Exception as e:      # noqa: BLE001
                    failures.append(f"infer_amp_fp16: {type(e).__name__}: {e}"[:300])
        if tl is not None:
            line["train_cfg3"] = {
                "workload": tl["name"] + ", 12 iters, whole training steps (forward + backward + gradient all-reduce + clip + AdamW), "
                            "model.train(): dropout on, BatchNorm batch statistics; policy " + tl["policy"],
                "ms_per_step": round(1e3 * tl["dt"] / tl["steps"], 3), "pairs_per_s": round(tl["value"], 3), "steps": tl["steps"],
                "warmup": tl["warmup"], "n_gpus": world, "loss": round(tl["loss"], 4), "first_loss": round(tl["first_loss"], 4),
                "first_loss_pinned": tl["first_loss_pinned"], "first_loss_ok": tl["first_loss_ok"],
                "skipped_steps": tl["skipped_steps"], "allreduce_ms_per_step": tl["allreduce_ms"],
                "roofline": tl.get("roofline"), "amp_fp16": tl.get("amp")}
        if tl4 is not None:
            line["train_cfg4"] = {
                "workload": tl4["name"] + ", 12 iters, whole training steps, model.train(): dropout on, frozen BatchNorm; policy " + tl4["policy"],
                "ms_per_step": round(1e3 * tl4["dt"] / tl4["steps"], 3), "pairs_per_s": round(tl4["value"], 3), "steps": tl4["steps"],
                "warmup": tl4["warmup"], "n_gpus": world, "loss": round(tl4["loss"], 4), "first_loss": round(tl4["first_loss"], 4),
                "first_loss_pinned": tl4["first_loss_pinned"], "first_loss_ok": tl4["first_loss_ok"],
                "skipped_steps": tl4["skipped_steps"], "allreduce_ms_per_step": tl4["allreduce_ms"]}
        if failures:
            line["pin_failed"] = any("first-step loss" in f for f in failures)
            line["failed_legs"] = failures
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    return 3 if failures else 0           # non-zero only AFTER the JSON line is out (a failed leg must not cost the run its other numbers)


if __name__ == "__main__":
    raise SystemExit(main())
