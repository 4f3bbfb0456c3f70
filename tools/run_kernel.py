#!/usr/bin/env python3
"""Launch one hot-path kernel a few times at the bench shapes (for rocprofv3 --pmc passes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from craft_amd import ops  # noqa: E402
from craft_amd.hip import PROB_DTYPE, Precision, pick  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "pv"
    prec = Precision.parse(sys.argv[2] if len(sys.argv) > 2 else "mixed")
    B, H8, W8 = int(os.environ.get("B", 4)), int(os.environ.get("H8", 56)), int(os.environ.get("W8", 128))
    N, M, Dv = H8 * W8, 4, 128
    dev = torch.device("cuda")
    if which == "pv":
        ldp = ops.round_up(N, 32)
        pv = pick(prec, "pv")
        P = torch.rand(B, M, N, ldp, device=dev).div_(N / 2).to(PROB_DTYPE[pv])
        if not os.environ.get("CRAFT_P_ROWMAJOR"):       # the layout the forward pass runs: 32 x 64 tiles (CRAFT_P_TILED)
            P = ops.probs_tiled(P)
        vT = torch.randn(B, M * Dv, ldp, device=dev).to(PROB_DTYPE[pv])
        O = torch.empty(B, M, N, Dv, device=dev)
        for _ in range(5):
            ops.attn_apply(P, vT, Dv, pv, out=O)
        reps = int(os.environ.get("REPS", 0))
        if reps:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record()
            for _ in range(reps):
                ops.attn_apply(P, vT, Dv, pv, out=O)
            e.record()
            torch.cuda.synchronize()
            print(f"attn_apply (k_pv16): {s.elapsed_time(e) / reps * 1e3:.1f} us per launch")
    elif which == "rows":
        # the short-K nn.Linear / 1x1 products of the forward pass: raw weights (k_gemm_rows) vs packed weights (k_gemm_rows_wf)
        pp = pick(prec, "proj")
        reps = int(os.environ.get("REPS", 20))
        for name, cin, cout, vt in (("convc1 324->256", 324, 256, False), ("V^T 128->512", 128, 512, True), ("q/k 256->256", 256, 256, False),
                                    ("q/k 128->128", 128, 128, False)):
            x = torch.randn(B, N, cin + 4, device=dev)[..., :cin]
            w = torch.randn(cout, cin, device=dev) / cin ** 0.5
            bias = torch.randn(cout, device=dev)
            pk = ops.pack_linear_weight(w, pp)
            for label, pkd in (("raw   ", None), ("packed", pk)):
                def run():
                    if vt:
                        return ops.linear_t(x, w, N, prec, Dv=128, packed=pkd)
                    return ops.linear(x, w, bias, prec, packed=pkd)
                for _ in range(3):
                    y = run()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(reps):
                    y = run()
                e.record()
                torch.cuda.synchronize()
                us = s.elapsed_time(e) / reps * 1e3
                fl = 2.0 * B * N * cin * cout * (3 if pp == 3 else 1)
                by = 4.0 * B * N * cin + (2.0 if vt else 4.0) * B * N * cout
                print(f"{name:18s} {label} {us:7.1f} us   {fl / us / 1e6:6.1f} TF/s (MFMA)   {by / us / 1e3:6.0f} GB/s", flush=True)
    elif which in ("gru", "grustep", "menc", "head"):
        from craft_amd import CRAFT, default_args
        from craft_amd.synth import synth_state_dict
        m = CRAFT(default_args())
        m.load_state_dict(synth_state_dict(m.state_dict(), seed=1234))
        ub = m.cuda().eval().update_block
        hx = torch.randn(B, N, 512, device=dev)
        ws = ub.workspace(B, N, dev)
        corr = torch.randn(B, N, 324, device=dev)
        c0, c1, fl = ops.coords_init(None, B, H8, W8, dev)
        reps = int(os.environ.get("REPS", 0))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fields = ub.gru.context_tokens(hx[..., 128:256], (H8, W8), prec) if which == "grustep" else None
        for i in range(3 + reps):
            if i == 3:
                s.record()
            if which == "gru":
                ub.gru.forward_tokens(hx, (H8, W8), ws, prec)
            elif which == "grustep":          # the forward pass's form: context channels hoisted into per-pixel bias fields (K 1920 per gate conv)
                ub.gru.step_tokens(hx, (H8, W8), ws, prec, fields, 128, 256)
            elif which == "menc":
                ub.encoder.forward_tokens(fl, corr, (H8, W8), hx[..., 256:384], ws, prec)
            else:
                ub.flow_head_tokens(hx, (H8, W8), c1, c0, fl, None, ws, prec)
        if reps:
            e.record()
            torch.cuda.synchronize()
            print(f"{which} B={B} {H8}x{W8}: {s.elapsed_time(e) / reps * 1e3:.1f} us per call ({B * ((H8 + 7) // 8) * ((W8 + 15) // 16)} patches)")
    elif which == "convtok":
        # the GRU's z|r convolution shape (1x5, 384 -> 256) on fp32 tokens (k_conv_halo_wf)
        from craft_amd.hip import call, ACT_NONE, W_PACKED, PREC_F16X3
        KH, KW, cin, cout = int(os.environ.get("KH", 1)), int(os.environ.get("KW", 5)), int(os.environ.get("CIN", 384)), int(os.environ.get("COUT", 256))
        x = torch.randn(B, N, cin, device=dev)
        w = torch.randn(cout, cin, KH, KW, device=dev) / (cin * KH * KW) ** 0.5
        wp = ops.pack_conv_weights(w, PREC_F16X3)
        zb = torch.zeros(cout, device=dev)
        y = torch.empty(B, N, cout, device=dev)
        for _ in range(3 + int(os.environ.get("REPS", 4))):
            call("craft_conv2d_nhwc", x, cin, cin, wp, zb, cout, KH, KW, ACT_NONE, y, cout, B, H8, W8, PREC_F16X3 | W_PACKED)
    elif which in ("fnet", "cnet"):
        from craft_amd import CRAFT, default_args
        from craft_amd.synth import synth_state_dict, synth_pair
        m = CRAFT(default_args())
        m.load_state_dict(synth_state_dict(m.state_dict(), seed=1234))
        m = m.cuda().eval()
        im1, im2, _ = synth_pair(B, 448, 1024, seed=0)
        raw = torch.cat([im1, im2]).cuda() if which == "fnet" else im1.cuda()
        enc = m._henc_f if which == "fnet" else m._henc_c
        for _ in range(int(os.environ.get("REPS", 2))):
            enc.forward_tokens(raw, prec)
    elif which in ("wgrad", "gemm_tt"):
        from craft_amd import hip as H
        Bq, h8, w8, cin, cout = 8, 46, 62, 512, 256
        npix = Bq * h8 * w8
        x = torch.randn(npix, cin, device=dev)
        dy = torch.randn(npix, cout, device=dev)
        cp = pick(prec, "conv")
        if which == "wgrad":
            dw = torch.zeros(cout, 1, 5, cin, device=dev)
            for _ in range(3):
                H.call("craft_conv2d_wgrad", x, cin, cin, dy, cout, cout, 1, 5, Bq, h8, w8, dw, None, None, 0, cp)
        else:       # the same contraction without taps: dW = dY^T X (k-major x k-major, split-K)
            dw = torch.zeros(cout, cin, device=dev)
            for _ in range(3):
                H.call("craft_gemm", dy, 1, cout, 0, 0, x, 1, cin, 0, 0, dw, cin, 0, 0, 1, 1, cout, cin, npix, 1.0, 1, 0, cp)
    elif which == "corr":
        import math
        C, Mm = 256, 4
        q = torch.randn(B, N, C, device=dev)
        k = torch.randn(B, N, C, device=dev)
        tab = torch.randn(15, 15, device=dev)
        pyr = ops.CorrPyramid(B, H8, W8, 4, dev)
        for _ in range(int(os.environ.get("REPS", 2))):
            ops.corr_build(q, k, H8, W8, Mm, 1 / math.sqrt(C // Mm), tab, 0.5, 0.7, None, pyr, True, prec)
    elif which in ("probs", "probs_norm"):
        import math
        C, Mm = 128, 4
        q = torch.randn(B, N, C, device=dev)
        k = torch.randn(B, N, C, device=dev)
        tab = torch.randn(15, 15, device=dev)
        reps = int(os.environ.get("REPS", 2))
        fn = lambda: ops.attn_probs(q, k, H8, W8, Mm, 1 / math.sqrt(C // Mm), tab, 1.0, -1, None, prec, defer=which == "probs")  # noqa: E731
        out = fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        print(f"attn_probs ({which}, fused={not os.environ.get('CRAFT_NO_FUSED_PROBS')}): {s.elapsed_time(e) / reps:.3f} ms per call, "
              f"{out.numel() * out.element_size() / 1e9:.3f} GB of P")
    elif which == "flash":
        import math
        C, Mm, Dvf = 256, 4, 256
        q = torch.randn(B, N, C, device=dev)
        k = torch.randn(B, N, C, device=dev)
        x = torch.randn(B, N, C, device=dev)
        Wv = torch.randn(Mm * Dvf, C, device=dev) / 16
        tab = torch.randn(15, 15, device=dev)
        ldt = ops.round_up(N, 32)
        vT = ops.linear_t(x, Wv, ldt, prec, Dv=Dvf, acc_order=True)
        O = torch.empty(B, Mm, N, Dvf, device=dev)
        reps = int(os.environ.get("REPS", 5))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.flash_attention(q, k, vT, H8, W8, Mm, Dvf, 1 / math.sqrt(C // Mm), tab, 0.5, -1, None, prec, out=O)
        s.record()
        for _ in range(reps):
            ops.flash_attention(q, k, vT, H8, W8, Mm, Dvf, 1 / math.sqrt(C // Mm), tab, 0.5, -1, None, prec, out=O)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        flop = 2.0 * B * Mm * N * N * (3 * 64 + Dvf)
        print(f"flash attention (pack + kernel): {ms:.3f} ms  = {flop / ms / 1e9:.0f} TFLOP/s executed (f16x3 scores)")
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
