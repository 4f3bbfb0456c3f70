"""craft_gemm_pk (batched GEMM over packed operands, csrc/gemm_pkb.inc.hpp) against float64 matmul: the three operand-kind pairs in the
roles they play in an attention layer (O = P V, dV = P^T dO, dP = dO V^T; autograd of setrans.py:373-384, :520-557), batch strides
(outer / inner), ragged M / N / K that hit every tile instantiation and the clamped / masked tile borders, and the three operand modes."""
import pytest
import torch

from craft_amd import autograd as AG
from craft_amd import hip
from craft_amd.hip import PREC_BF16, PREC_F16, PREC_F16X3, round_up

pytestmark = pytest.mark.gpu
PRECS = [(PREC_F16X3, 2e-5), (PREC_F16, 4e-3), (PREC_BF16, 3e-2)]


@pytest.fixture(scope="module")
def device():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda")


def _rel(got, ref):
    return ((got.double().cpu() - ref).norm() / ref.norm()).item()


# (B, Mh, N tokens, C per mode): P [B, Mh, N, N], V [B, N, Mh*C]
ATTN_CASES = [(2, 3, 77, 32), (1, 4, 300, 128), (2, 1, 130, 64), (1, 2, 520, 256), (3, 2, 40, 160)]


@pytest.mark.parametrize("prec,tol", PRECS)
@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention_products(device, case, prec, tol):
    B, Mh, N, C = case
    g = torch.Generator().manual_seed(sum(case))
    ld = round_up(N, 32)
    P = torch.zeros(B, Mh, N, ld)
    P[..., :N] = torch.softmax(torch.randn(B, Mh, N, N, generator=g) * 2.0, dim=-1)
    V = torch.randn(B, N, Mh * C, generator=g)
    dO = torch.randn(B, Mh, N, C, generator=g)
    Pd, Vd, dOd = P.to(device), V.to(device), dO.to(device)
    Vm = V.view(B, N, Mh, C).permute(0, 2, 1, 3).double()                      # [B, Mh, N, C]
    Ppk = AG.PkMat(B * Mh, N, ld, prec, device).fill(Pd)                       # rows i, channels j
    Vpk = AG.PkMat(B, N, Mh * C, prec, device).fill(Vd)                        # rows j, channels (m, c)
    dOpk = AG.PkMat(B * Mh, N, C, prec, device).fill(dOd)                      # rows i, channels c
    cg = C // 32
    # O = P V  (CH, ROWS)
    O = torch.full((B, Mh, N, C), 7.0, device=device)
    AG.gemm_pk(Ppk, Ppk.desc(AG.PK_CH, Mh, 1), Vpk, Vpk.desc(AG.PK_ROWS, 1, 0, 0, 0, cg), O, C, Mh * N * C, N * C, Mh, B * Mh, N, C, ld)
    assert _rel(O, P[..., :N].double() @ Vm) < tol
    # dV = P^T dO  (ROWS, ROWS) into the [B, N, Mh, C] layout of V
    dV = torch.full((B, N, Mh, C), 7.0, device=device)
    AG.gemm_pk(Ppk, Ppk.desc(AG.PK_ROWS, Mh, 1), dOpk, dOpk.desc(AG.PK_ROWS, Mh, 1), dV, Mh * C, N * Mh * C, C, Mh, B * Mh, N, C, N)
    ref = (P[..., :N].double().transpose(-1, -2) @ dO.double()).permute(0, 2, 1, 3)
    assert _rel(dV, ref) < tol
    # dP = dO V^T  (CH, CH), padded columns untouched
    dP = torch.full((B, Mh, N, ld), 7.0, device=device)
    AG.gemm_pk(dOpk, dOpk.desc(AG.PK_CH, Mh, 1), Vpk, Vpk.desc(AG.PK_CH, 1, 0, 0, 0, cg), dP, ld, Mh * N * ld, N * ld, Mh, B * Mh, N, N, C)
    assert _rel(dP[..., :N], dO.double() @ Vm.transpose(-1, -2)) < tol
    if ld > N:
        assert float((dP[..., N:] - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("prec,tol", PRECS)
def test_column_block_output(device, prec, tol):
    """craft_gemm_pk with CRAFT_PK_CBLK: the (ROWS, ROWS) product dV = P^T [dO_1 .. dO_T] of all T iterations written as [B, N, T, Mh, C] (the
    modes of a batch entry interleave inside every iteration's column block) holds the same values as the plain [B, N, Mh, T*C] store,
    and an iteration's [B*N, Mh*C] slice is a strided view (train_update._phase2)."""
    B, Mh, N, C, T = 2, 4, 200, 128, 3
    g = torch.Generator().manual_seed(12)
    ld = round_up(N, 32)
    P = torch.zeros(B, Mh, N, ld)
    P[..., :N] = torch.softmax(torch.randn(B, Mh, N, N, generator=g) * 2.0, dim=-1)
    dO = torch.randn(B, Mh, N, T * C, generator=g)
    Ppk = AG.PkMat(B * Mh, N, ld, prec, device).fill(P.to(device))
    dOpk = AG.PkMat(B * Mh, N, T * C, prec, device).fill(dO.to(device))
    TC = T * C
    plain = torch.full((B, N, Mh, TC), 7.0, device=device)
    AG.gemm_pk(Ppk, Ppk.desc(AG.PK_ROWS, Mh, 1), dOpk, dOpk.desc(AG.PK_ROWS, Mh, 1), plain, Mh * TC, N * Mh * TC, TC, Mh, B * Mh, N, TC, N)
    blk = torch.full((B, N, T, Mh, C), 7.0, device=device)
    AG.gemm_pk(Ppk, Ppk.desc(AG.PK_ROWS, Mh, 1), dOpk, dOpk.desc(AG.PK_ROWS, Mh, 1), blk, Mh * TC, N * Mh * TC, C, Mh, B * Mh, N, TC, N, c_blk_shift=7)
    assert torch.equal(blk, plain.view(B, N, Mh, T, C).permute(0, 1, 3, 2, 4))
    ref = (P[..., :N].double().transpose(-1, -2) @ dO.double()).permute(0, 2, 1, 3)            # [B, N, Mh, T*C]
    assert _rel(plain, ref) < tol
    with pytest.raises(hip.CraftHipError):                                                      # c_inner must be the block width
        AG.gemm_pk(Ppk, Ppk.desc(AG.PK_ROWS, Mh, 1), dOpk, dOpk.desc(AG.PK_ROWS, Mh, 1), blk, Mh * TC, N * Mh * TC, TC, Mh, B * Mh, N, TC, N, c_blk_shift=7)


@pytest.mark.parametrize("M,N,K", [(257, 129, 96), (31, 65, 700), (512, 256, 64), (300, 20, 2852)])
def test_ragged_single_batch(device, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    Kc = round_up(K, 4)
    A, Bm = torch.randn(M, Kc, generator=g), torch.randn(N, Kc, generator=g)
    A[:, K:], Bm[:, K:] = 0.0, 0.0
    ref = A.double() @ Bm.double().t()
    Ad, Bd = A.to(device), Bm.to(device)
    # CH x CH: rows = m / n
    a1, b1 = AG.PkMat(1, M, round_up(Kc, 32), PREC_F16X3, device).fill(Ad), AG.PkMat(1, N, round_up(Kc, 32), PREC_F16X3, device).fill(Bd)
    C = torch.zeros(M, N, device=device)
    AG.gemm_pk(a1, a1.desc(AG.PK_CH, 0, 0), b1, b1.desc(AG.PK_CH, 0, 0), C, N, 0, 0, 1, 1, M, N, Kc)
    assert _rel(C, ref) < 2e-5
    # ROWS x ROWS: the transposed sources (rows = k); M / N must be channel counts % 4
    M4, N4 = M // 4 * 4, N // 4 * 4
    if M4 and N4:
        At, Bt = A[:M4].t().contiguous().to(device), Bm[:N4].t().contiguous().to(device)
        a0, b0 = AG.PkMat(1, Kc, round_up(M4, 32), PREC_F16X3, device).fill(At), AG.PkMat(1, Kc, round_up(N4, 32), PREC_F16X3, device).fill(Bt)
        C0 = torch.zeros(M4, N4, device=device)
        AG.gemm_pk(a0, a0.desc(AG.PK_ROWS, 0, 0), b0, b0.desc(AG.PK_ROWS, 0, 0), C0, N4, 0, 0, 1, 1, M4, N4, Kc)
        assert _rel(C0, ref[:M4, :N4]) < 2e-5
        # CH x ROWS
        C1 = torch.zeros(M, N4, device=device)
        AG.gemm_pk(a1, a1.desc(AG.PK_CH, 0, 0), b0, b0.desc(AG.PK_ROWS, 0, 0), C1, N4, 0, 0, 1, 1, M, N4, Kc)
        assert _rel(C1, ref[:, :N4]) < 2e-5


def test_softmax_writes_the_packed_probabilities(device):
    """craft_attn_softmax_fwd's Ppk output == craft_pack_operand of its (dropped) fp32 output, bit for bit, padding rows zero."""
    B, M, H8, W8 = 2, 3, 7, 9
    N = H8 * W8
    ld = round_up(N, 32)
    g = torch.Generator().manual_seed(1)
    S0 = (torch.randn(B, M, N, ld, generator=g) * 3.0).to(device)
    for prec in (PREC_F16X3, PREC_F16, PREC_BF16):
        for p in (0.0, 0.2):
            S = S0.clone()
            pk = AG.PkMat(B * M, N, ld, prec, device)
            pk.buf.fill_(0x7e00)                                        # NaN patterns: every element must be written
            AG.call("craft_attn_softmax_fwd", S, ld, B, M, H8, W8, None, 0, 0.0, -1, None, None, None, float(p), 99, pk.buf, pk.rows_total, pk.np_, prec)
            S2 = S0.clone()
            Pd = torch.empty_like(S2) if p > 0 else None
            AG.call("craft_attn_softmax_fwd", S2, ld, B, M, H8, W8, None, 0, 0.0, -1, None, None, Pd, float(p), 99, None, 0, 0, 0)
            assert torch.equal(S, S2)
            ref = AG.PkMat(B * M, N, ld, prec, device).fill(Pd if p > 0 else S2)
            planes = 2 if prec == PREC_F16X3 else 1
            a = pk.buf.view(planes, pk.ncg, pk.rows_total, 32)[:, :, :B * M * pk.np_]
            b = ref.buf.view(planes, pk.ncg, pk.rows_total, 32)[:, :, :B * M * pk.np_]
            assert torch.equal(a, b)


def test_softmax_backward_writes_dS_packed(device):
    """craft_attn_softmax_bwd's dSpk output == craft_pack_operand of its fp32 dS (bit for bit, padding rows zero), with the dropout mask
    and the positional-table gradient unchanged."""
    from craft_amd import hip
    B, M, H8, W8 = 2, 3, 7, 9
    N = H8 * W8
    ld = round_up(N, 32)
    g = torch.Generator().manual_seed(2)
    P = torch.zeros(B, M, N, ld)
    P[..., :N] = torch.softmax(torch.randn(B, M, N, N, generator=g) * 2.0, dim=-1)
    G = torch.randn(B, M, N, ld, generator=g)
    P, G = P.to(device), G.to(device)
    for prec in (PREC_F16X3, PREC_F16, PREC_BF16):
        d1, rep1 = G.clone(), torch.zeros(hip.STATS_REPLICAS, 225, device=device)
        AG.call("craft_attn_softmax_bwd", P, d1, ld, B, M, H8, W8, 7, 0.5, None, None, rep1, 0.2, 9, None, 0, 0, 0)
        d2, rep2 = G.clone(), torch.zeros(hip.STATS_REPLICAS, 225, device=device)
        pk = AG.PkMat(B * M, N, ld, prec, device)
        pk.buf.fill_(0x7e00)
        AG.call("craft_attn_softmax_bwd", P, d2, ld, B, M, H8, W8, 7, 0.5, None, None, rep2, 0.2, 9, pk.buf, pk.rows_total, pk.np_, prec)
        assert torch.equal(d2, G)                                   # the fp32 gradient buffer is left alone
        assert torch.allclose(rep1.sum(0), rep2.sum(0), rtol=1e-4, atol=1e-5)
        ref = AG.PkMat(B * M, N, ld, prec, device).fill(d1)
        planes = 2 if prec == PREC_F16X3 else 1
        a = pk.buf.view(planes, pk.ncg, pk.rows_total, 32)[:, :, :B * M * pk.np_]
        b = ref.buf.view(planes, pk.ncg, pk.rows_total, 32)[:, :, :B * M * pk.np_]
        assert torch.equal(a, b)


@pytest.mark.parametrize("prec,tol", PRECS)
def test_scores_chain_on_packed_operands(device, prec, tol):
    """Scores -> AttnSoftmax with a ScoreLink (Q K^T, dQ = dS K, dK = dS^T Q on craft_gemm_pk, dS handed over as a pack) against the same
    chain on the fp32-source engine: values and gradients."""
    B, M, H8, W8, C = 2, 4, 6, 10, 128
    N = H8 * W8
    g = torch.Generator().manual_seed(4)
    q0, k0 = torch.randn(B, N, C, generator=g).to(device), torch.randn(B, N, C, generator=g).to(device)
    tab0 = (torch.randn(15, 15, generator=g) * 0.5).to(device)
    gout = torch.randn(B, M, N, round_up(N, 32), generator=g).to(device)
    res = []
    for link in (None, AG.ScoreLink()):
        q, k, tab = q0.clone().requires_grad_(True), k0.clone().requires_grad_(True), tab0.clone().requires_grad_(True)
        S = AG.Scores.apply(q, k, M, 0.2, prec, link)
        Pm = AG.AttnSoftmax.apply(S, tab, 0.5, -1, None, (H8, W8), 0.1, 5, None, link)
        Pm.backward(gout.clone())
        res.append((Pm.detach()[..., :N].clone(), q.grad, k.grad, tab.grad))
    # f16x3 scores keep the fp32-source engine (packed three-term products measured neutral-to-negative): the link is
    # declined unless the policy gives the score-gradient products a single-plane role (sbw: test_train_backward's mixed cases)
    assert link.want == (prec != PREC_F16X3) and link.dS is None
    for a, b in zip(res[0], res[1]):
        assert ((a - b).norm() / a.norm()).item() < 2 * tol


def test_benchmark_shape_against_the_fp32_source_engine(device):
    """The three attention products at BASELINE configs[3]'s shape (8 x 4 modes x 2852 x 2852 probabilities, 128 channels per mode: batch
    offsets of hundreds of MB inside the packs, 32 batches over the 8 XCD queues, the 4-stage 128-row tile and the 256 x 256 tile) equal
    craft_gemm on the same fp32 tensors -- which tests/test_train_backward.py::test_gemm_layouts holds to float64."""
    B, Mh, N, C = 8, 4, 46 * 62, 128
    ld = round_up(N, 32)
    g = torch.Generator(device="cpu").manual_seed(9)
    P = torch.zeros(B, Mh, N, ld, device=device)
    P[..., :N] = torch.softmax(torch.randn(B, Mh, N, N, device=device) * 2.0, dim=-1)
    V = torch.randn(B, N, Mh * C, generator=g).to(device)
    dO = torch.randn(B, Mh, N, C, generator=g).to(device)
    prec = PREC_F16X3
    Ppk, Vpk, dOpk = AG.PkMat(B * Mh, N, ld, prec, device).fill(P), AG.PkMat(B, N, Mh * C, prec, device).fill(V), AG.PkMat(B * Mh, N, C, prec, device).fill(dO)
    cg = C // 32
    O, O2 = torch.empty(B, Mh, N, C, device=device), torch.empty(B, Mh, N, C, device=device)
    AG.gemm_pk(Ppk, Ppk.desc(AG.PK_CH, Mh, 1), Vpk, Vpk.desc(AG.PK_ROWS, 1, 0, 0, 0, cg), O, C, Mh * N * C, N * C, Mh, B * Mh, N, C, ld)
    AG.gemm(P, ld, 1, Mh * N * ld, N * ld, V, 1, Mh * C, N * Mh * C, C, O2, C, Mh * N * C, N * C, Mh, B * Mh, N, C, N, prec=prec)
    assert ((O - O2).norm() / O2.norm()).item() < 1e-6
    dV, dV2 = torch.empty(B, N, Mh, C, device=device), torch.empty(B, N, Mh, C, device=device)
    AG.gemm_pk(Ppk, Ppk.desc(AG.PK_ROWS, Mh, 1), dOpk, dOpk.desc(AG.PK_ROWS, Mh, 1), dV, Mh * C, N * Mh * C, C, Mh, B * Mh, N, C, N)
    AG.gemm(P, 1, ld, Mh * N * ld, N * ld, dO, 1, C, Mh * N * C, N * C, dV2, Mh * C, N * Mh * C, C, Mh, B * Mh, N, C, N, prec=prec)
    assert ((dV - dV2).norm() / dV2.norm()).item() < 1e-6
    dP, dP2 = torch.empty(B, Mh, N, ld, device=device), torch.zeros(B, Mh, N, ld, device=device)
    AG.gemm_pk(dOpk, dOpk.desc(AG.PK_CH, Mh, 1), Vpk, Vpk.desc(AG.PK_CH, 1, 0, 0, 0, cg), dP, ld, Mh * N * ld, N * ld, Mh, B * Mh, N, N, C)
    Vm = V.view(B, N, Mh, C).permute(0, 2, 1, 3).contiguous()
    AG.gemm(dO, C, 1, Mh * N * C, N * C, Vm, C, 1, Mh * N * C, N * C, dP2, ld, Mh * N * ld, N * ld, Mh, B * Mh, N, N, C, prec=prec)
    assert ((dP[..., :N] - dP2[..., :N]).norm() / dP2[..., :N].norm()).item() < 1e-6
