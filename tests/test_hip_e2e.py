"""End-to-end parity of craft_amd.CRAFT (HIP hot path) with the reference.

1. Against the committed golden fixtures (captured from the imported reference, tools/make_golden.py)
   for the canonical configuration, batch 2 + flow_init, the clamp-triggering weights and the GMA /
   plain-correlation / F2-mask variants — fp32 mode, tolerance BASELINE.md §3: final flow_up max-abs <=
   1e-2 px (we hold 3e-3), low-res flow <= 1e-3.
2. bf16 / fp16 MFMA modes against the same goldens with the stated mixed-precision tolerance
   (mean EPE delta <= 0.01 px per BASELINE.md §3).
3. Size-independent properties at BASELINE.json's full sizes (448x1024, 768x1024), where the oracle is
   too slow to be a per-element checker: attention rows sum to one, batch independence, lookup of the
   normalised volume has zero mean / unit variance statistics, determinism.
"""
import numpy as np
import pytest
import torch

from craft_amd import CRAFT, default_args, ops
from craft_amd.hip import PREC_BF16, PREC_F16, PREC_F32
from craft_amd.synth import synth_pair, synth_state_dict
from golden_util import CASES, Golden

pytestmark = pytest.mark.gpu


def build(g: Golden, device, precision="fp32"):
    m = g.meta
    args = default_args(hip_precision=precision, **m["over"])
    model = CRAFT(args)
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=m["seed"], qk_gain=m["qk_gain"]), strict=True)
    return model.to(device).eval()


@pytest.mark.parametrize("case", CASES)
def test_forward_matches_reference_fp32(device, case):
    g = Golden(case)
    model = build(g, device, "fp32")
    im1, im2 = g.images()
    fi = g.flow_init()
    with torch.no_grad():
        if g.meta.get("warm"):      # the first pair of a warm-started sequence: no flow_init; it decides the cached lsinu code
            model(im1.to(device), im2.to(device), iters=g.meta["iters"], test_mode=2)
        flow_lo, preds = model(im1.to(device), im2.to(device), iters=g.meta["iters"],
                               flow_init=None if fi is None else fi.to(device), test_mode=2)
    assert len(preds) == g.meta["iters"]
    g.check("flow_lo", flow_lo, 1e-3, 1e-3)
    for it, p in enumerate(preds):
        g.check(f"up{it}", p, 1e-3, 3e-3)
    if "up_last.full" in g.z.files:
        d = (preds[-1].cpu() - torch.from_numpy(g.z["up_last.full"])).abs()
        assert d.max().item() < 3e-3, f"flow_up max abs diff {d.max().item():.3e} px"
        d = (flow_lo.cpu() - torch.from_numpy(g.z["flow_lo.full"])).abs()
        assert d.max().item() < 5e-4


def test_test_modes_and_skipped_mask_head(device):
    """test_mode=1 skips the mask head / upsampling on all but the last iteration; the result must be
    bit-identical to the last prediction of test_mode=2 and test_mode=0 (network.py:262-267)."""
    g = Golden("canon_128x256_T4")
    model = build(g, device)
    im1, im2 = (t.to(device) for t in g.images())
    with torch.no_grad():
        lo1, up1 = model(im1, im2, iters=4, test_mode=1)
        lo2, ups = model(im1, im2, iters=4, test_mode=2)
        ups0 = model(im1, im2, iters=4, test_mode=0)
    assert torch.equal(lo1, lo2) and torch.equal(up1, ups[-1]) and torch.equal(ups0[-1], up1)
    assert isinstance(ups0, list) and len(ups0) == 4


@pytest.mark.parametrize("precision,mean_tol,max_tol", [("mixed", 0.001, 0.005), ("mixed_fp32conv", 0.01, 0.05), ("fp16", 0.03, 0.1),
                                                        ("bf16", 0.3, 0.8)])
def test_forward_mixed_precision_modes(device, precision, mean_tol, max_tol):
    """16-bit MFMA operands, fp32 accumulate, against the reference's fp32 output (4 iterations).
    "mixed" is the shipped mixed-precision policy: fp16 storage of the attention probabilities + fp16 MFMA for
    P.V, and split-fp16 (F16X3, fp32-class) MFMA for projections, Q K^T and the convolutions; it is held to the
    fp32 bound of BASELINE.md (mean EPE delta <= 1e-3 px).  "mixed_fp32conv" (fp16 attention contractions + exact
    fp32 MFMA convolutions) holds the 16-bit-attention bound (<= 0.01 px; measured 0.0014-0.0074, 0.0042 px at
    448x1024 / 12 iterations).  All-fp16 (what the reference's autocast does) and all-bf16 are selectable but
    do NOT meet that bound on the synthetic weights (measured 0.012 / 0.11 px): their bounds here only guard
    against regressions."""
    g = Golden("canon_128x256_T4")
    model = build(g, device, precision)
    im1, im2 = g.images()
    with torch.no_grad():
        lo, up = model(im1.to(device), im2.to(device), iters=4, test_mode=1)
    ref = torch.from_numpy(g.z["up_last.full"])
    epe = (up.cpu() - ref).pow(2).sum(1).sqrt()
    assert epe.mean().item() < mean_tol, f"mean EPE delta {epe.mean().item():.4f} px"
    assert epe.max().item() < max_tol, f"max EPE delta {epe.max().item():.4f} px"


def test_checkpoint_layout_roundtrip(device, tmp_path):
    """Reference-layout checkpoint ({'model': {'module.<key>': ...}}) loads and reproduces the outputs."""
    from craft_amd import load_checkpoint
    g = Golden("canon_128x256_T4")
    model = build(g, device)
    ck = {"model": {"module." + k: v.cpu() for k, v in model.state_dict().items()}, "optimizer": {}, "lr_scheduler": {}}
    path = str(tmp_path / "craft-synth.pth")
    torch.save(ck, path)
    m2 = CRAFT(default_args()).eval()
    res = load_checkpoint(m2, path, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    m2 = m2.to(device)
    im1, im2 = (t.to(device) for t in g.images())
    with torch.no_grad():
        a = model(im1, im2, iters=2, test_mode=1)[1]
        b = m2(im1, im2, iters=2, test_mode=1)[1]
    assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------
# full-size properties (BASELINE.json configs 2 and 3)
# ------------------------------------------------------------------------------------------------
def _full_model(device, precision):
    model = CRAFT(default_args(hip_precision=precision))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234), strict=True)
    return model.to(device).eval()


@pytest.mark.parametrize("precision", ["fp32", "mixed"])
def test_full_size_448x1024_properties(device, precision):
    model = _full_model(device, precision)
    im1, im2, _ = synth_pair(2, 448, 1024, seed=3)
    im1, im2 = im1.to(device), im2.to(device)
    with torch.no_grad():
        lo2, up2 = model(im1, im2, iters=12, test_mode=1)
        lo1, up1 = model(im1[:1], im2[:1], iters=12, test_mode=1)
        lo1b, up1b = model(im1[:1], im2[:1], iters=12, test_mode=1)
    assert up2.shape == (2, 2, 448, 1024) and lo2.shape == (2, 2, 56, 128)
    assert torch.isfinite(up2).all()
    # determinism and batch independence (eval-mode forward is exactly batch independent in the reference;
    # ours up to the double-precision atomics of the global-LayerNorm statistics)
    assert (up1 - up1b).abs().max().item() < 1e-4
    assert (up2[:1] - up1).abs().max().item() < 2e-3
    if precision == "mixed":
        ref = _full_model(device, "fp32")(im1[:1], im2[:1], iters=12, test_mode=1)[1]
        epe = (up1 - ref).pow(2).sum(1).sqrt()
        assert epe.mean().item() < 2e-3, f"mixed-precision mean EPE delta {epe.mean().item():.4f} px at 448x1024 / 12 iters"
    # the flow must be non-trivial (the synthetic pair moves by up to 12 px)
    assert up2.abs().max().item() > 0.5


def test_full_size_attention_and_volume_statistics(device):
    """768x1024 (KITTI-size stress): probabilities sum to one; the lazily-normalised volume sampled on the
    identity grid (level-0 centre tap = channel 40) has the statistics the global LayerNorm implies."""
    import math
    from oracle import craft_oracle as O
    B, H8, W8, C, M = 1, 96, 128, 128, 4
    N = H8 * W8
    g = torch.Generator().manual_seed(0)
    x = O.layernorm_lastdim(torch.randn(B, N, C, generator=g)).to(device)
    Wq = (torch.randn(C, C, generator=g) * math.sqrt(2.5 / C)).to(device)
    Wk = (torch.randn(C, C, generator=g) * math.sqrt(2.5 / C)).to(device)
    tab = (torch.randn(15, 15, generator=g) * 0.5).to(device)
    q = ops.linear(x, Wq, None, PREC_F32)
    k = ops.linear(x, Wk, None, PREC_F32)
    scale = 1 / math.sqrt(C // M)
    mx = ops.score_max(q, k, H8, W8, M, scale, PREC_F32)
    P = ops.attn_probs(q, k, H8, W8, M, scale, tab, 1.0, -1, mx, PREC_F32)
    rs = P.sum(-1)
    assert (rs - 1).abs().max().item() < 2e-5
    del P
    pyr = ops.CorrPyramid(B, H8, W8, 4, device)
    ops.corr_build(q, k, H8, W8, M, scale, tab, 0.5, 0.8, mx, pyr, True, PREC_F32)
    l0 = pyr.lv[0].double()
    mu, rstd = pyr.mu_rstd[0, 0].item(), pyr.mu_rstd[0, 1].item()
    assert abs(l0.mean().item() - mu) < 1e-5 * max(1.0, abs(mu))
    assert abs(1.0 / math.sqrt(l0.var(unbiased=False).item() + 1e-12) - rstd) < 1e-4 * rstd
    c0 = ops.coords_init(None, B, H8, W8, device)[0]
    look = ops.corr_lookup(pyr, c0, 4)
    centre = look[0, :, 40]                       # level 0, a = b = 4: the (i, i) entries, normalised
    diag = (torch.diagonal(pyr.lv[0].reshape(N, N)) - mu) * rstd
    assert (centre - diag).abs().max().item() < 1e-4
    # pooling linearity: level-1 mean equals level-0 mean (even sizes)
    assert abs(pyr.lv[1].double().mean().item() - l0.mean().item()) < 1e-5


@pytest.mark.parametrize("which", ["fnet", "cnet"])
@pytest.mark.parametrize("precision", ["fp32", "mixed"])
@pytest.mark.parametrize("H,W", [(128, 256), (376, 1248), (136, 200)])      # (KITTI-padded and small: statistics tiles straddle images)
def test_hip_encoder_matches_pytorch_module(device, which, precision, H, W):
    """craft_amd.hip_encoder.HipEncoder (lazy InstanceNorm / folded BatchNorm, fused residual tails) against the
    PyTorch BasicEncoder module it wraps, same weights, on the GPU (extractor.py:124-196)."""
    from craft_amd.hip import Precision
    from craft_amd.hip_encoder import HipEncoder
    model = _full_model(device, precision)
    enc = getattr(model, which)
    im1, im2, _ = synth_pair(2 if H * W < 200000 else 1, H, W, seed=5)
    raw = torch.cat([im1, im2, im1[:1]]).to(device)            # odd batch: tiles straddle more often
    with torch.no_grad():
        ref = enc(2 * (raw / 255.0) - 1)
        got = HipEncoder(enc).forward_tokens(raw, Precision.parse(precision))
    B, C, H8, W8 = ref.shape
    got = ops.tokens_to_nchw(got, H8, W8)
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err < 2e-4 * max(1.0, scale), f"{which}/{precision}: max abs diff {err:.3e} (|ref| max {scale:.2f})"


@pytest.mark.parametrize("precision", ["fp32", "mixed", "bf16"])
def test_hip_encoder_reads_the_two_frames_in_place(device, precision):
    """fnet's batch = [frames 1 | frames 2] (extractor.py:171-176): handed over as a pair, the MFMA stem reads the two tensors where they lie
    (craft_stem_conv7x7_mfma_pair) -- the same bits as the concatenated batch, InstanceNorm statistics included."""
    from craft_amd.hip import Precision
    from craft_amd.hip_encoder import HipEncoder
    model = _full_model(device, precision)
    im1, im2, _ = synth_pair(3, 136, 200, seed=9)
    im1, im2 = im1.to(device), im2.to(device)
    for which in ("fnet", "cnet"):
        henc = HipEncoder(getattr(model, which))
        with torch.no_grad():
            a = henc.forward_tokens(torch.cat([im1, im2]), Precision.parse(precision))
            b = henc.forward_tokens((im1, im2), Precision.parse(precision))
        # (the statistics' double-precision atomics make two passes agree to rounding, not to the bit)
        assert a.shape == b.shape and (a - b).abs().max().item() < 1e-5 * max(1.0, a.abs().max().item()), which
    with pytest.raises(ValueError, match="same shape"):
        henc.forward_tokens((im1, im2[:1]), Precision.parse(precision))


@pytest.mark.parametrize("H,W,B", [(136, 200, 2), (128, 264, 1)])
def test_ragged_sizes_against_oracle(device, H, W, B):
    """Image sizes whose token grid is not a multiple of any tile (17x25 = 425 tokens, 16x33 = 528): ragged MFMA
    tiles, padded P rows, partial conv patches, floor pooling with odd sizes, and the PyTorch fallback of the
    encoder runner (stride-2 statistics tiles do not divide the image) — compared with the CPU oracle directly."""
    from oracle import craft_oracle as O
    model = _full_model(device, "fp32")
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    im1, im2, _ = synth_pair(B, H, W, seed=11)
    g = torch.Generator().manual_seed(3)
    fi = 2.0 * torch.randn(B, 2, H // 8, W // 8, generator=g)
    with torch.no_grad():
        lo, ups = model(im1.to(device), im2.to(device), iters=3, flow_init=fi.to(device), test_mode=2)
    lo_ref, ups_ref = O.craft_forward(sd, O.OracleConfig(), im1, im2, iters=3, flow_init=fi, test_mode=2)
    assert (lo.cpu() - lo_ref).abs().max().item() < 1e-3
    for a, b in zip(ups, ups_ref):
        assert (a.cpu() - b).abs().max().item() < 5e-3


def test_iters_zero_and_single(device):
    g = Golden("canon_128x256_T4")
    model = build(g, device)
    im1, im2 = (t.to(device) for t in g.images())
    with torch.no_grad():
        lo, up = model(im1, im2, iters=1, test_mode=1)
        preds = model(im1, im2, iters=1, test_mode=0)
    assert torch.equal(preds[0], up) and lo.shape == (1, 2, 16, 32)


def test_batch_sliced_streams_match_single_stream(device):
    """args.hip_streams > 1 runs the refinement loop of batch slices on separate HIP streams: same results."""
    g = Golden("canon_b2_128x192_T3_init")
    im1, im2 = (t.to(device) for t in g.images())
    outs = []
    for n in (1, 2):
        model = build(g, device)
        model.args.hip_streams = n
        with torch.no_grad():
            outs.append(model(im1, im2, iters=3, test_mode=2))
    (lo1, ups1), (lo2, ups2) = outs
    assert len(ups1) == len(ups2) == 3
    # (two separate forwards: equal up to the summation order of the double-precision statistics atomics)
    assert (lo1 - lo2).abs().max().item() < 1e-4
    for a, b in zip(ups1, ups2):
        assert (a - b).abs().max().item() < 1e-4


def test_side_streams_and_fused_attention_do_not_change_results(device, monkeypatch):
    """The context chain / flow branch on side streams (args.hip_fork, CRAFT_NO_FORK) only reorder independent work, and the
    flash-fused F2 attention (CRAFT_NO_FLASH -> attn_probs + attn_apply) is the same layer with an online softmax: the
    predictions agree to the fp16 rounding of P (mixed policy: fp16 P.V in both paths)."""
    g = Golden("canon_b2_128x192_T3_init")
    im1, im2 = (t.to(device) for t in g.images())

    def run(env):
        for k in ("CRAFT_NO_FORK", "CRAFT_NO_FLASH"):
            monkeypatch.delenv(k, raising=False)
        for k in env:
            monkeypatch.setenv(k, "1")
        model = build(g, device, precision="mixed")
        with torch.no_grad():
            out = model(im1, im2, iters=3, test_mode=1)
        torch.cuda.synchronize()
        return out
    lo, up = run(())
    lo_nf, up_nf = run(("CRAFT_NO_FORK",))
    lo_nl, up_nl = run(("CRAFT_NO_FLASH",))
    assert (up - up_nf).abs().max().item() < 1e-4, "side streams changed the result"
    assert (up - up_nl).abs().max().item() < 5e-3 and (lo - lo_nl).abs().max().item() < 1e-3, "fused attention deviates"


def test_an_exception_between_fork_and_join_leaves_the_streams_joined(device, monkeypatch):
    """CRAFT.forward forks the context chain onto a side stream (and, for batches of 6..12, the refinement loop onto two): when the pass
    dies between a fork and its join -- here: the F2 transformer raises while the side stream still runs cnet and the intra-frame
    attention into `hx`, a tensor of the CALLER's stream -- the side stream must be joined before the exception leaves (network._JoinOnError),
    and the next forward on the same model gives the result of an undisturbed one."""
    B, H, W = 2, 128, 192
    model = CRAFT(default_args(hip_precision="mixed", mixed_precision=True))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=9), strict=True)
    model = model.to(device).eval()
    im1, im2, _ = synth_pair(B, H, W, seed=10)
    im1, im2 = im1.to(device), im2.to(device)
    with torch.no_grad():
        ref = model(im1, im2, iters=3, test_mode=1)[1].clone()
    side = model._streams(1, im1.device)[0]
    orig = model.f2_trans.forward_tokens

    def boom(*a, **k):
        raise RuntimeError("injected failure between fork and join")
    for _ in range(3):
        monkeypatch.setattr(model.f2_trans, "forward_tokens", boom)
        with torch.no_grad(), pytest.raises(RuntimeError, match="injected failure"):
            model(im1, im2, iters=3, test_mode=1)
        # everything the side stream was given has been ordered in front of whatever the caller's stream does next
        torch.cuda.current_stream().synchronize()
        assert side.query(), "the side stream still has work the caller's stream never waited for"
        monkeypatch.setattr(model.f2_trans, "forward_tokens", orig)
        with torch.no_grad():
            again = model(im1, im2, iters=3, test_mode=1)[1]
        assert (again - ref).abs().max().item() < 1e-4          # (px; the statistics' double atomics may reorder the last bit)
