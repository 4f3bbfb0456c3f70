"""Packed-operand weight gradients (craft_pack_operand + craft_wgrad_pk, csrc/kernels_gemm_pk.hip) against torch: the pack layout
bit for bit, and dW / db of convolutions (every tap geometry of the update block and the encoders, channel counts that hit every tile
instantiation, ragged images) and of nn.Linear against float64 autograd of the reference operator (F.conv2d / F.linear:
update.py:49-64, :79-87; extractor.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from craft_amd import autograd as AG
from craft_amd.hip import PREC_BF16, PREC_F16, PREC_F16X3, round_up

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda")


def _unpack(pk: AG.Packed, planes: int):
    """[plane][C/32][rows_p][32] int16 -> float tensor [planes][rows_p][C_p]."""
    dt = torch.bfloat16 if pk.prec == PREC_BF16 else torch.float16
    t = pk.buf.view(dt).view(planes, pk.C_p // 32, pk.rows_p, 32).permute(0, 2, 1, 3).reshape(planes, pk.rows_p, pk.C_p)
    return t.float()


@pytest.mark.parametrize("prec", [PREC_F16X3, PREC_F16, PREC_BF16])
def test_pack_layout_spatial_and_plain(device, prec):
    g = torch.Generator().manual_seed(0)
    B, H, W, C, ph, pw = 2, 5, 7, 40, 1, 2
    x = torch.randn(B, H * W, C, generator=g).to(device)
    planes = 2 if prec == PREC_F16X3 else 1
    cs = torch.zeros(C, device=device)
    pk = AG.Packed(x, prec, (B, H, W, ph, pw), colsum=cs)
    assert pk.C_p == 64 and pk.Wp == W + 2 * pw and pk.guard == ph * pk.Wp + pw and pk.K % 32 == 0 and pk.rows_p >= 2 * pk.guard + pk.K
    u = _unpack(pk, planes)
    val = u.sum(0)                                           # hi + lo
    ref = torch.zeros(pk.rows_p, pk.C_p, device=device)
    grid = F.pad(x.view(B, H, W, C), (0, 0, pw, pw, ph, ph)).reshape(-1, C)
    ref[pk.guard:pk.guard + grid.shape[0], :C] = grid
    dt = torch.bfloat16 if prec == PREC_BF16 else torch.float16
    if prec == PREC_F16X3:
        hi = ref.to(dt).float()
        assert torch.equal(u[0], hi)                          # hi plane = RNE fp16 of x, lo = fp16 of the remainder
        assert torch.equal(u[1], (ref - hi).to(dt).float())
        assert (val - ref).abs().max().item() <= 2.0 ** -21 * ref.abs().max().item()
    else:
        assert torch.equal(u[0], ref.to(dt).float())
    assert torch.allclose(cs, x.sum((0, 1)), rtol=1e-5, atol=1e-4)
    # plain rows, C % 4 == 0 but not % 8, strided source (a column slice)
    wide = torch.randn(3, 50, 64, generator=g).to(device)
    xs = wide[..., 8:44]
    pk2 = AG.Packed(xs, prec)
    u2 = _unpack(pk2, planes).sum(0)
    assert pk2.guard == 0 and pk2.K == round_up(150, 32)
    want = torch.zeros(pk2.rows_p, 64, device=device)
    want[:150, :36] = xs.reshape(150, 36)
    tol = 0.0 if prec == PREC_F16X3 else None
    if prec == PREC_F16X3:
        assert (u2 - want).abs().max().item() <= 2.0 ** -21 * want.abs().max().item()
    else:
        assert torch.equal(u2, want.to(dt).float())


CONV_CASES = [  # (B, H, W, cin, cout, KH, KW)
    (2, 12, 20, 512, 256, 1, 5),      # SepConvGRU z|r horizontal: 256 x 256 tiles
    (2, 12, 20, 512, 128, 5, 1),      # q vertical: 128 x 256
    (1, 9, 13, 128, 256, 3, 3),       # flow head / mask head conv1 (ragged image)
    (2, 10, 12, 256, 192, 3, 3),      # convc2: cout 192 -> 256-row tile with a masked group
    (1, 16, 24, 64, 64, 3, 3),        # encoder layer1: several taps per N tile
    (1, 8, 8, 32, 128, 7, 7),         # convf1 (2 -> 32 padded input channels), 49 taps
    (3, 6, 10, 96, 32, 3, 3),         # heads' padded 2-channel outputs: 64 x 256 tile
    (1, 8, 16, 256, 32, 1, 1),
]


@pytest.mark.parametrize("prec,tol", [(PREC_F16X3, 2e-5), ("f16x3_x16", 6e-4), (PREC_F16, 4e-3), (PREC_BF16, 3e-2)])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_weight_and_bias_gradient(device, case, prec, tol):
    """"f16x3_x16": dY in hi / lo planes, the activation operand X as ONE fp16 plane (CRAFT_WGRAD_X_PREC: two MFMAs per product, the
    `wgx=fp16` role of the "mixed" training policy) -- X's rounding (2^-12 relative, random sign) bounds the error at ~2e-4."""
    B, H, W, cin, cout, KH, KW = case
    xprec = prec
    if prec == "f16x3_x16":
        prec, xprec = PREC_F16X3, PREC_F16
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, H * W, cin, generator=g)
    dy = torch.randn(B, H * W, cout, generator=g)
    w = torch.zeros(cout, cin, KH, KW, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    xn = x.double().view(B, H, W, cin).permute(0, 3, 1, 2)
    y = F.conv2d(xn, w, bias, padding=(KH // 2, KW // 2))
    y.backward(dy.double().view(B, H, W, cout).permute(0, 3, 1, 2))
    ref = w.grad.permute(0, 2, 3, 1).float()                 # [cout][KH][KW][cin]
    acc = torch.full((cout, KH, KW, cin), 0.5, device=device)             # ACCUMULATED into
    db = torch.zeros(cout, device=device)
    geom = (B, H, W, KH // 2, KW // 2)
    AG.wgrad_pk((AG.Packed(dy.to(device), prec, geom, colsum=db), AG.Packed(x.to(device), xprec, geom)), KH, KW, acc)
    got = (acc - 0.5).cpu()
    err = (got - ref).norm() / ref.norm()
    assert err < tol, f"relative L2 {err:.2e}"
    assert (got - ref).abs().max() < 40 * tol * ref.abs().max()
    assert torch.allclose(db.cpu(), bias.grad.float(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("prec,tol", [(PREC_F16X3, 2e-5), (PREC_BF16, 3e-2)])
@pytest.mark.parametrize("rows,cin,cout", [(5704, 324, 256), (1000, 128, 512), (333 * 4, 256, 576), (64, 36, 20)])
def test_linear_weight_gradient(device, rows, cin, cout, prec, tol):
    g = torch.Generator().manual_seed(rows + cin)
    x = torch.randn(1, rows, cin, generator=g)
    dy = torch.randn(1, rows, cout, generator=g)
    ref = (dy[0].double().t() @ x[0].double()).float()
    co_p, ci_p = round_up(cout, 32), round_up(cin, 32)
    acc = torch.zeros(co_p, ci_p, device=device)
    AG.wgrad_pk((AG.Packed(dy.to(device), prec), AG.Packed(x.to(device), prec)), 1, 1, acc)
    got = acc[:cout, :cin].cpu()
    assert (got - ref).norm() / ref.norm() < tol
    assert float(acc[cout:].abs().max() if co_p > cout else 0.0) == 0.0 and float(acc[:, cin:].abs().max() if ci_p > cin else 0.0) == 0.0


def test_large_k_full_training_shape(device):
    """The SepConvGRU z|r weight gradient at BASELINE configs[3]'s shape (8 x 46 x 62 pixels, 512 -> 256, 1 x 5) against torch on the
    GPU in float64 (K = 23 760 padded pixels, 25 K splits): the split-K / XCD block map and the guard rows at full size."""
    B, H, W, cin, cout, KH, KW = 8, 46, 62, 512, 256, 1, 5
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(B, H * W, cin, generator=g).to(device)
    dy = (torch.randn(B, H * W, cout, generator=g) * 1e-3).to(device)
    xn = x.double().view(B, H, W, cin).permute(0, 3, 1, 2)
    ref = torch.nn.grad.conv2d_weight(xn, (cout, cin, KH, KW), dy.double().view(B, H, W, cout).permute(0, 3, 1, 2), padding=(0, 2))
    ref = ref.permute(0, 2, 3, 1).float()
    acc = torch.zeros(cout, KH, KW, cin, device=device)
    geom = (B, H, W, 0, 2)
    AG.wgrad_pk((AG.Packed(dy, PREC_F16X3, geom), AG.Packed(x, PREC_F16X3, geom)), KH, KW, acc)
    assert ((acc - ref).norm() / ref.norm()).item() < 2e-5


@pytest.mark.parametrize("nseg", [2, 12, 19])
def test_segments_concatenate_k(device, nseg):
    """The calls of one layer in a pass as ONE launch over the concatenated K (12 refinement iterations; 19 > the 16 segments one
    launch carries): equal to the sum of the per-call products."""
    B, H, W, cin, cout, KH, KW = 1, 10, 14, 128, 96, 3, 3
    g = torch.Generator().manual_seed(nseg)
    geom = (B, H, W, 1, 1)
    pairs, ref = [], torch.zeros(cout, KH, KW, cin, dtype=torch.float64)
    for _ in range(nseg):
        x = torch.randn(B, H * W, cin, generator=g)
        dy = torch.randn(B, H * W, cout, generator=g)
        xn = x.double().view(B, H, W, cin).permute(0, 3, 1, 2)
        ref += torch.nn.grad.conv2d_weight(xn, (cout, cin, KH, KW), dy.double().view(B, H, W, cout).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
        pairs.append((AG.Packed(dy.to(device), PREC_F16X3, geom), AG.Packed(x.to(device), PREC_F16X3, geom)))
    acc = torch.zeros(cout, KH, KW, cin, device=device)
    AG.wgrad_pk(pairs, KH, KW, acc)
    assert ((acc.cpu().double() - ref).norm() / ref.norm()).item() < 2e-5


@pytest.mark.parametrize("xprec", [PREC_F16X3, PREC_F16])
@pytest.mark.parametrize("KH,KW", [(1, 5), (5, 1), (3, 3)])
def test_two_pack_x_operand(device, KH, KW, xprec):
    """The X operand as the channel concatenation of two packs (cat([h, x]) of SepConvGRU: x packed once per pass, h / r*h separately;
    the first from a column slice of a wider buffer) equals the product over the materialised cat."""
    from craft_amd.train_update import _CatPack
    B, H, W, c0, c1, cout = 2, 10, 12, 128, 384, 256
    g = torch.Generator().manual_seed(KH * 7 + KW)
    wide = torch.randn(B, H * W, 640, generator=g).to(device)
    h, x = wide[..., 128:256], wide[..., 256:640]
    dy = torch.randn(B, H * W, cout, generator=g).to(device)
    geom = (B, H, W, KH // 2, KW // 2)
    gp = AG.Packed(dy, PREC_F16X3, geom)
    acc = torch.zeros(cout, KH, KW, c0 + c1, device=device)
    AG.wgrad_pk([(gp, _CatPack(AG.Packed(h, xprec, geom), AG.Packed(x, xprec, geom)))] * 3, KH, KW, acc)
    ref = torch.zeros_like(acc)
    AG.wgrad_pk([(gp, AG.Packed(torch.cat([h, x], -1).contiguous(), xprec, geom))] * 3, KH, KW, ref)
    for lo, hi in ((0, 128), (128, 512)):
        e = ((acc[..., lo:hi] - ref[..., lo:hi]).norm() / ref[..., lo:hi].norm()).item()
        assert e < 1e-6, (lo, hi, e)
    # and a list of sources packed into ONE pack
    one = AG.Packed([h, x], xprec, geom)
    acc2 = torch.zeros_like(acc)
    AG.wgrad_pk([(gp, one)] * 3, KH, KW, acc2)
    assert ((acc2 - ref).norm() / ref.norm()).item() < 1e-6
