#!/usr/bin/env python3
"""Error of a CHEAPER split product, emulated on the CPU oracle (no kernel exists): DESIGN 7's unmeasured lead.

The library's fp32-class convolution is x.w = hi.hi + lo.hi + hi.lo over fp16 planes (three fp16 MFMAs).  The two correction terms are
2^-11 of the main term; as MX-scaled fp8 MFMAs (e4m3 values, one power-of-two scale per 32 elements along K -- gfx950's
v_mfma_scale_f32_32x32x64_f8f6f4, twice the fp16 rate) a product would cost 2 instead of 3 fp16-MFMA equivalents.  This script replaces
every F.conv2d of oracle/craft_oracle.py by an emulation of

    f16x3  : conv(hi_x, hi_w) + conv(lo_x, hi_w) + conv(hi_x, lo_w)                          (what the kernels compute today)
    fp8corr: conv(hi_x, hi_w) + conv(q8(lo_x), q8(hi_w)) + conv(q8(hi_x), q8(lo_w))           (q8: MX e4m3 blocks of 32 channels)
    f16x1  : conv(hi_x, hi_w)                                                                  (plain fp16 operands)

(fp32 accumulation in all of them) and reports the deviation of the final prediction from the plain fp32 oracle at bench.py's workload
(synthetic weights seed 1234, one 448x1024 pair, 12 iterations).  Only the convolutions are emulated: the attention products stay fp32.

    python tools/fp8_correction_eval.py [--height 448 --width 1024 --iters 12 --modes "f16x3;fp8corr;f16x1"]
    python tools/fp8_correction_eval.py --modes "fnet.=w16;cnet.=w16;update_block.mask=w16"       (a plan per run: weight-name prefix = mode)
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def split16(t):
    hi = t.half().float()
    return hi, (t - hi).half().float()


def q8_blocks(t, dim):
    """MX e4m3: blocks of 32 along `dim`, one power-of-two scale per block chosen so that the block's largest magnitude lands in [256, 448]
    (e4m3's top binade), values rounded to float8_e4m3fn."""
    n = t.shape[dim]
    pad = (-n) % 32
    tt = t.movedim(dim, -1)
    if pad:
        tt = F.pad(tt, (0, pad))
    shp = tt.shape
    b = tt.reshape(*shp[:-1], -1, 32)
    amax = b.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    scale = torch.exp2(torch.floor(torch.log2(448.0 / amax)))
    q = (b * scale).to(torch.float8_e4m3fn).float() / scale
    q = q.reshape(shp)[..., :n]
    return q.movedim(-1, dim)


class Emul:
    """mode: one of f16x3 | fp8corr | f16x1 | w16 (weights as ONE fp16 plane: hi.hi + lo_x.hi_w, two MFMAs) | x16 (activations as one plane:
    hi.hi + hi_x.lo_w) for every convolution -- or a PLAN "prefix=mode,prefix=mode,..." over the state-dict names of the weights
    (e.g. "fnet.=w16,cnet.=w16"), every convolution the plan does not name staying f16x3."""

    def __init__(self, mode, sd=None):
        self.calls, self.names = 0, {}
        self.plan = None
        if "=" in mode:
            self.plan = [tuple(kv.split("=")) for kv in mode.split(",")]
            self.names = {id(v): k for k, v in sd.items()}
            self.mode = "f16x3"
        else:
            self.mode = mode

    def conv2d(self, x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        self.calls += 1
        mode = self.mode
        if self.plan is not None:
            name = self.names.get(id(w), "?")
            for pre, m in self.plan:
                if name.startswith(pre):
                    mode = m
        kw = dict(stride=stride, padding=padding, dilation=dilation, groups=groups)
        hx, lx = split16(x)
        hw, lw = split16(w)
        y = F.conv2d(hx, hw, None, **kw)
        if mode == "f16x3":
            y = y + F.conv2d(lx, hw, None, **kw) + F.conv2d(hx, lw, None, **kw)
        elif mode == "fp8corr":
            y = y + F.conv2d(q8_blocks(lx, 1), q8_blocks(hw, 1), None, **kw) + F.conv2d(q8_blocks(hx, 1), q8_blocks(lw, 1), None, **kw)
        elif mode == "w16":
            y = y + F.conv2d(lx, hw, None, **kw)
        elif mode == "x16":
            y = y + F.conv2d(hx, lw, None, **kw)
        elif mode != "f16x1":
            raise ValueError(mode)
        return y if b is None else y + b.view(1, -1, 1, 1)


class _FProxy:
    """torch.nn.functional with conv2d replaced (the oracle module's `F`)."""

    def __init__(self, emul):
        self._e = emul

    def __getattr__(self, k):
        return self._e.conv2d if k == "conv2d" else getattr(F, k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=448)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--modes", default="f16x3;fp8corr;f16x1", help="';'-separated: a mode for every convolution, or a plan prefix=mode,prefix=mode")
    ap.add_argument("--seed", type=int, default=100)
    a = ap.parse_args()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    from craft_amd import CRAFT, default_args
    from craft_amd.synth import synth_pair, synth_state_dict
    from oracle import craft_oracle as O
    sd = synth_state_dict(CRAFT(default_args()).state_dict(), seed=1234)
    im1, im2, _ = synth_pair(1, a.height, a.width, seed=a.seed)
    t0 = time.time()
    with torch.no_grad():
        _, ref = O.craft_forward(sd, O.OracleConfig(), im1, im2, iters=a.iters, test_mode=1)
    print(f"# fp32 oracle: {time.time() - t0:.1f} s; |flow_up| mean {ref.abs().mean():.3f} px, max {ref.abs().max():.2f} px", flush=True)
    real_F = O.F
    for mode in a.modes.split(";"):
        e = Emul(mode, sd)
        O.F = _FProxy(e)
        try:
            t0 = time.time()
            with torch.no_grad():
                _, up = O.craft_forward(sd, O.OracleConfig(), im1, im2, iters=a.iters, test_mode=1)
        finally:
            O.F = real_F
        epe = (up - ref).pow(2).sum(1).sqrt()
        print(f"{mode:28s} {e.calls:4d} convolutions emulated, {time.time() - t0:6.1f} s: EPE vs fp32 oracle mean {epe.mean():.3e} px  max {epe.max():.3e} px",
              flush=True)


if __name__ == "__main__":
    main()
