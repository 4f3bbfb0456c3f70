/* libcraft_hip.so — C ABI of the MI355X-native CRAFT hot path (gfx950).
 *
 * The reference (askerlee/craft @ 2024-10-20) has no FFI of its own: its hot path is PyTorch op call
 * sites inside core/{corr,setrans,gma,update,network}.py.  Each entry point below replaces one of
 * those Python functions (cited as file:line under the reference root) and is what a maintainer of
 * the reference would bind with ctypes (INTEGRATION.md shows the stub).  Conventions:
 *
 *   - plain `extern "C"`, raw DEVICE pointers + sizes, no torch types; `stream` is a hipStream_t
 *     (NULL = default stream).  Every call only enqueues work on `stream`; no internal allocation,
 *     no synchronisation, no global state -> re-entrant per stream and hipGraph-capturable.
 *   - the caller owns every buffer, including workspaces (sizes documented per call);
 *   - weights are passed on every call (no hidden copies: optimizer / load_state_dict updates are seen);
 *   - return 0 on success, a hipError_t value, or a CRAFT_ERR_* code (craft_hip_error_string()).
 *
 * Data layout ("tokens"): every activation on the hot path is channels-last fp32,
 *   T[b][n][c] at  base + (b*N + n)*ld + c,   n = y*W8 + x  (H8 = H/8, W8 = W/8, N = H8*W8);
 * `ld` (row stride, in floats) lets a tensor be a column slice of a wider buffer, which is how the
 * reference's torch.cat()s are made free.  Row strides and channel counts must be multiples of 4.
 * Coordinates / flow are tokens with 2 channels in (x, y) order (utils.py:82-85).
 *
 * Precision codes: 0 = fp32 (v_mfma_f32_32x32x2_f32, exact fp32 products), 1 = bf16 MFMA with fp32
 * accumulate, 2 = fp16 MFMA with fp32 accumulate (operands are converted while staged into LDS).
 */
#ifndef CRAFT_HIP_H
#define CRAFT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4 (round 6): craft_conv2d_pk (round 5's halo convolution over plane-packed activations, which no product code called) was REMOVED, and
 * craft_attn_apply accepts CRAFT_PV_ROWS(8 | 10 | 12 | 14) (the 8-wave kernel).  No signature changed. */
#define CRAFT_HIP_ABI_VERSION 4

#define CRAFT_PREC_F32 0
#define CRAFT_PREC_BF16 1
#define CRAFT_PREC_F16 2
#define CRAFT_PREC_F16X3 3 /* fp32 operands split into two fp16 planes, 3 fp16 MFMAs per product: fp32-class results */

#define CRAFT_ACT_NONE 0
#define CRAFT_ACT_TANH 1
#define CRAFT_ACT_RELU 2

int craft_hip_abi_version(void);
const char* craft_hip_error_string(int code);

/* SETransInputFeatEncoder.forward, pos_code_type='bias' (setrans.py:763-800) and the tanh/relu split of
 * cnet's output (network.py:209-212): channels [c_off, c_off+C) of `src` -> act -> optional LayerNorm over
 * C (no affine, eps 1e-12) -> tokens.  src_nchw=1: src is [B, Ctot, HW]; 0: src is tokens with row stride
 * src_ld.  C <= 256. */
int craft_tokens(const float* src, int src_nchw, int B, int Ctot, int c_off, int C, int HW, long src_ld,
                 int act, int do_ln, float* dst, long dst_ld, void* stream);

/* tokens (first C columns) -> NCHW [B, C, HW]   (the reshape at setrans.py:617 / network.py return values) */
int craft_tokens_to_nchw(const float* src, long ld, int B, int C, int HW, float* dst, void* stream);

/* nn.Linear (setrans.py:507-508, :373):  y[r][o] = sum_c x[r][c]*w[o][c] + bias[o]   (bias may be NULL) */
int craft_linear(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy, long rows,
                 int cin, int cout, int prec, void* stream);
/* the same projection written transposed per sample, yT[b][o][n] with row stride ldt >= N (used for V^T,
 * setrans.py:373-378), stored as float (out_prec 0), bf16 (1) or fp16 (2).  The caller zero-fills columns
 * [N, ldt) once.  frag_rows = 0: plain row-major [cout][ldt].  frag_rows = Dv > 0 (16-bit out_prec only; Dv % 32 == 0,
 * cout % Dv == 0, ldt % 16 == 0): every group of Dv rows (one attention mode) is stored in MFMA B-fragment order,
 * yT_group[((g*(Dv/32) + nb)*64 + lane)*8 + j] = yT[nb*32 + (lane & 31)][g*16 + (lane >> 5)*8 + j] -- the layout the
 * 16-bit craft_attn_apply streams from L2 straight into MFMA operand registers (same footprint: Dv * ldt values). */
int craft_linear_t(const float* x, long ldx, const float* w, void* yT, long ldt, int B, int N, int cin,
                   int cout, int out_prec, int frag_rows, int prec, void* stream);
/* or-ed into frag_rows: the 16 keys of a k-group are enumerated in MFMA ACCUMULATOR order (position 8*h + j holds key
 * 8*(j >> 2) + 4*h + (j & 3)) -- the V^T operand of craft_flash_attention, whose P operand is a score accumulator. */
#define CRAFT_FRAG_ACC_ORDER 0x10000

/* Global max of the raw scaled scores Q_m K_m^T * scale over batch, modes, i, j (the .max().item() of
 * setrans.py:520-521) as an order-preserving uint in *max_ord; consumers clamp to [-100, 100] iff that max
 * is > 100 (setrans.py:524-529) without a host round trip.  q,k: projected tokens [B][N][ld] (ld == the tokens'
 * row stride for all samples), mode m = columns [m*d, m*d+d).  max_ord points to 32 unsigned words: [0] is
 * the result (0 = "no score exceeds the threshold"), the rest is scratch for a Cauchy-Schwarz norm bound that
 * lets the exact N^2 pass exit immediately when no score can reach 100. */
int craft_score_max(const float* q, long ldq, const float* k, long ldk, int B, int H8, int W8, int M, int d,
                    float scale, unsigned* max_ord, int prec, void* stream);

/* TransCorrBlock.corr up to (not including) the global LayerNorm (corr.py:191-199; setrans.py:507-550):
 * c(i,j) = sum_m s_m softmax_m(w_aggr*s_m) + pos_w*pb(i,j) with s_m = clamp?(Q_m(i).K_m(j)*scale), written
 * to pyramid level 0 [B*N][H8][W8]; sums[b] = (sum c, sum c^2) in double for the lazy LayerNorm.
 * pos_tab: SlidingPosBiases2D.biases [(2R+1)^2] (setrans.py:644-708), NULL = none; clamp_ord from
 * craft_score_max (NULL = never clamp).  M=1, pos_tab=NULL, scale=1/sqrt(C) gives CorrBlock.corr
 * (corr.py:73-81).
 * ws (or NULL): scratch of 2 * (2 * B*N*M*d) 16-bit values.  With it, prec = F16X3, M = 4 and d = 64 the operands are split
 * into fp16 hi/lo planes ONCE (Q pre-multiplied by scale) and the build streams them with pure copies -- the fp32 -> hi/lo
 * conversion inside the K loop was a third of the kernel's instructions.  Same result up to fp32 rounding order. */
int craft_corr_build(const float* q, long ldq, const float* k, long ldk, int B, int H8, int W8, int M, int d,
                     float scale, const float* pos_tab, int R, float pos_w, float w_aggr,
                     const unsigned* clamp_ord, float* pyr0, double* sums, void* ws, int prec, void* stream);

/* craft_corr_build with the pyramid fused in (corr.py:186-189): the keys of a wave tile are an 8x8 cell of the key image held in
 * registers per query, so levels 1..3 (2x2 / 4x4 / 8x8 averages, floor sizes) are register sums of the tile that produced
 * level 0, which is never read back.  Implemented for prec = F16X3, M = 4, d = 64 with the ws of craft_corr_build, H8, W8 >= 8
 * and all four levels; anything else returns CRAFT_ERR_UNSUPPORTED (10003) and the caller uses craft_corr_build +
 * craft_corr_finish.  Follow it with craft_corr_finish(pyr0, NULL, NULL, NULL, ...) for the (mean, rstd) of the lazy LayerNorm.
 * prec | CRAFT_PYR_TILED: levels 0 and 1 are written in the tiled layout described at craft_corr_lookup (pyr0 / pyr1 then hold
 * ceil(H8/8)*ceil(W8/16)*128 and ceil(h1/4)*ceil(w1/8)*32 floats per query). */
#define CRAFT_PYR_TILED 0x200
int craft_corr_build_pyramid(const float* q, long ldq, const float* k, long ldk, int B, int H8, int W8, int M, int d, float scale,
                             const float* pos_tab, int R, float pos_w, float w_aggr, const unsigned* clamp_ord, float* pyr0,
                             float* pyr1, float* pyr2, float* pyr3, double* sums, void* ws, int prec, void* stream);

/* corr.py:186-189 + :200-204: levels 1..3 by 2x2 average pooling (floor sizes; pass NULL to stop early) and
 * mu_rstd[b] = (mean, 1/sqrt(var+1e-12)) over all N*N entries (do_norm=0: (0,1)). */
int craft_corr_finish(const float* pyr0, float* pyr1, float* pyr2, float* pyr3, const double* sums,
                      float* mu_rstd, int B, int H8, int W8, int do_norm, void* stream);

/* CorrBlock.__call__ + bilinear_sampler (corr.py:47-71, utils.py:65-79): out[q][l*lvl_stride + col_off + a*(2r+1) + b] =
 * bilinear_zero_pad(LN(pyr_l)[q], x/2^l + a - r, y/2^l + b - r), q = b*N + n, coords tokens (x,y).  lvl_stride = 0 means
 * (2r+1)^2 (one volume).  The two-way correlation of --f1 (corr.py:164-171: two volumes concatenated on the channel axis
 * of every level) is two calls with lvl_stride = 2*(2r+1)^2 and col_off = 0 / (2r+1)^2.
 * levels | CRAFT_PYR_TILED: the pyramid is in craft_corr_build_pyramid's tiled layout (prec | CRAFT_PYR_TILED there): per query,
 * level 0 as ceil(H8/8) x ceil(W8/16) tiles of 8 x 16 keys (128 floats, row-major inside the tile, tiles row-major) and level 1
 * (h1 = H8/2, w1 = W8/2) as ceil(h1/4) x ceil(w1/8) tiles of 4 x 8; levels 2 and 3 stay row-major.  A tile is the patch one
 * workgroup of the build kernel owns, so its stores fill whole 128-byte lines (row-major: 64-byte half lines completed by another
 * workgroup -> 1.3 x the bytes written plus read-modify-write fetches, profiles/r3/pmc_corr_build_levels.txt). */
int craft_corr_lookup(const float* pyr0, const float* pyr1, const float* pyr2, const float* pyr3, int levels,
                      const float* mu_rstd, const float* coords, int B, int H8, int W8, int radius,
                      float* out, long ldo, int lvl_stride, int col_off, void* stream);

/* CrossAttFeatTrans up to the softmax (setrans.py:507-557): P[b][m][i][j] = softmax_j(clamp?(Q_m(i).K_m(j)*
 * scale) + pos_w*pb(i,j) + mask), mask = -1e9 where Chebyshev distance > mask_radius (setrans.py:580-584,
 * <=0: none).  P has row stride ldp (multiple of 32, >= N); columns [N, ldp) are written as zeros.
 * Element type of P by p_prec: float (0), bf16 (1), fp16 (2); prec selects the MFMA path of Q K^T.
 * relpos_h [B*M*N][2*H8-1], relpos_w [B*M*N][2*W8-1] (or both NULL): per-query relative-position scores of gma.RelPosEmb
 * (gma.py:21-50), logit += relpos_weight * (relpos_h[q][kh - qh + H8 - 1] + relpos_w[q][kw - qw + W8 - 1]); they are two
 * small GEMMs of the query against the embedding rows (craft_linear); scale = 0 gives the position_only variant.
 * rowsum = NULL: P is the normalised softmax.  rowsum != NULL ((2 + ceil(N / CRAFT_ATTN_CHUNK_KEYS)) * B*M*N floats:
 * [B][M][N] row sums out, then scratch for the row maxima and the per-key-chunk partial sums): DEFERRED normalisation -- P holds
 * exp(logit - rowmax) in (0, 1] and rowsum the row sums; craft_attn_apply given the same rowsum divides its output rows
 * by them, which is the same O.  The first pass of the kernel then needs no exponentials (it is VALU-bound). */
int craft_attn_probs(const float* q, long ldq, const float* k, long ldk, int B, int H8, int W8, int M, int d,
                     float scale, const float* pos_tab, int R, float pos_w, int mask_radius,
                     const unsigned* clamp_ord, const float* relpos_h, const float* relpos_w, float relpos_weight, void* P,
                     long ldp, float* rowsum, int p_prec, int prec, void* stream);

/* The DEFERRED form of craft_attn_probs (rowsum != NULL) as ONE launch of independent waves (round 4; setrans.py:507-557 for the
 * intra-frame attention, network.py:214): P holds 2^(logit2 - rowmax) in (0, 1] (16-bit, p_prec = fp16 or bf16), rowsum [B][M][N] the
 * row sums -- the same contract as craft_attn_probs with rowsum, consumed by craft_attn_apply.  A wave owns 32 queries for the whole
 * key range: exact row maxima in a first sweep, exponentials + whole 256-byte row segments of P in a second; the keys are pre-split
 * once into fp16 hi / lo planes in MFMA fragment order and stream from L2 straight into MFMA registers (no operand staging, no block
 * barrier).  ws: B*M*ceil(N/128)*16384 bytes of scratch (the packed keys).  Implemented for d = 32, prec = f16x3 or fp16, no
 * relative-position scores, N < 65536; anything else returns CRAFT_ERR_UNSUPPORTED (10003) and the caller uses craft_attn_probs.
 * p_prec | CRAFT_P_TILED: P is written in 32-query x 64-key tiles -- element (i, j) of entry (b, m) at
 *   ((b*M + m) * ceil(N/32) + (i >> 5)) * 32 * ldp + (j >> 6) * 2048 + (i & 31) * 64 + (j & 63),   ldp a multiple of 64, >= N
 * (buffer: B*M*ceil(N/32)*32*ldp elements; rows >= N are not written, columns [N, ldp) are zeros).  The probabilities are written
 * once and streamed 12 times by craft_attn_apply (network.py:214-230: one aggregation per refinement iteration); in this layout both
 * sides move whole 4 KiB runs instead of 128-byte row segments 2*ldp bytes apart.  Only craft_attn_apply (prec | CRAFT_P_TILED) reads it. */
#define CRAFT_P_TILED 0x400
int craft_attn_probs_fused(const float* q, long ldq, const float* k, long ldk, int B, int H8, int W8, int M, int d, float scale,
                           const float* pos_tab, int R, float pos_w, int mask_radius, const unsigned* clamp_ord, void* P, long ldp,
                           float* rowsum, void* ws, int p_prec, int prec, void* stream);

/* ExpandedFeatTrans.forward, matmul part (setrans.py:384): O[b][m][i][:] = sum_j P[b][m][i][j] * V_m[j][:],
 * with vT[b][m*Dv + c][j] (row stride ldp, zero beyond N) from craft_linear_t.  O: [B][M][N][Dv] fp32.
 * prec (0 fp32, 1 bf16, 2 fp16) is the element type of BOTH P (craft_attn_probs with p_prec = prec) and vT
 * (craft_linear_t with out_prec = prec), and the MFMA path.  For prec 1 / 2, vT must be in fragment order
 * (craft_linear_t with frag_rows = Dv) and Dv % 128 == 0; for prec 0 it is plain row-major.  rowsum: NULL for a
 * normalised P, else the row sums craft_attn_probs produced with it (O rows are divided by them).
 * 16-bit path: a block owns 32*r query rows x 128 value columns; r (4..7) is chosen from the grid size unless the caller
 * or-s CRAFT_PV_ROWS(r) into prec (tests pin every instantiation that way); r = 8 / 10 / 12 / 14 selects the 8-wave kernel (a block owns
 * 2 x 32*(r/2) rows: one V^T fetch per two row halves).
 * prec | CRAFT_P_TILED (16-bit only): P is in craft_attn_probs_fused's tiled layout, ldp its tiled row extent (multiple of 64); vT
 * keeps the row stride N rounded up to 32. */
#define CRAFT_PV_ROWS_SHIFT 20
#define CRAFT_PV_ROWS(r) ((r) << CRAFT_PV_ROWS_SHIFT)
int craft_attn_apply(const void* P, long ldp, const float* rowsum, const void* vT, int B, int N, int M, int Dv, float* O,
                     int prec, void* stream);

/* CrossAttFeatTrans + the matmul of ExpandedFeatTrans in one pass (setrans.py:507-557 + :384) for a layer whose
 * probabilities are used once (the F2 feature transformer): O[b][m][i][:] = sum_j softmax_j(clamp?(Q_m(i).K_m(j)*scale) +
 * pos_w*pb(i,j) + mask) V_m[j][:] with an online softmax -- the N x N probabilities never exist in memory.  Arguments as
 * craft_attn_probs (no relative-position scores) and craft_attn_apply; vT from craft_linear_t(out_prec = pv_prec,
 * frag_rows = Dv | CRAFT_FRAG_ACC_ORDER), row stride ldt = N rounded up to 32.  ws: scratch for the pre-split Q / K
 * fragments, B*M * (8*ceil(N/256) + ceil(N/32)) * (d/16) * planes * 1024 bytes (planes = 2 for score_prec 3, else 1).  Supported: d = 64, Dv = 256, score_prec 2 (fp16) or 3 (f16x3), pv_prec 2, N < 65536;
 * anything else returns CRAFT_ERR_UNSUPPORTED (10003) and the caller uses craft_attn_probs + craft_attn_apply. */
int craft_flash_attention(const float* q, long ldq, const float* k, long ldk, const void* vT, long ldt, int B, int H8, int W8,
                          int M, int d, int Dv, float scale, const float* pos_tab, int R, float pos_w, int mask_radius,
                          const unsigned* clamp_ord, float* O, void* ws, int score_prec, int pv_prec, void* stream);

/* Warm start, forward_interpolate (core/utils/utils.py:34-62): flow, out [B][2][H][W] (NCHW, dx then dy).  Every pixel takes
 * the flow of the source pixel whose forward-warped position is nearest (float64 distances, like the reference's
 * griddata(..., 'nearest') on float64 coordinates), among sources landing strictly inside (0, W) x (0, H); none: zeros. */
int craft_forward_interpolate(const float* flow, int B, int H, int W, float* out, void* stream);

/* ExpandedFeatTrans.forward tail (setrans.py:395-407): a_m = softmax_m(<O_m, w_agg>), out = LayerNorm(
 * skip_coeff * x + sum_m a_m O_m).  C = Dv in {64,128,192,256}. */
int craft_mode_pool_ln(const float* O, const float* x, long ldx, const float* w_agg, const float* skip_coeff,
                       int B, int N, int M, int C, float* out, long ldo, void* stream);

/* gma.Aggregate.forward tail (gma.py:138): out = mf + gamma * O */
int craft_gma_residual(const float* mf, long ldm, const float* O, const float* gamma, int B, int N, int C,
                       float* out, long ldo, void* stream);

/* Pre-pack a conv weight matrix w [rows][K] (rows = Cout, K = KH*KW*Cin laid out [KH][KW][Cin], K % 32 == 0) into the
 * operand layout the KxK convolution kernels stream without any LDS staging.  fp32 -> plain copy.  bf16 / fp16 / F16X3 ->
 * MFMA fragment order: out[((((kt*NB + nb)*PL + pl)*2 + kk)*64 + lane)*8 + j] = plane_pl(w[nb*32 + (lane&31)][kt*32 + kk*16 +
 * (lane>>5)*8 + j]) with NB = ceil(rows/32) (missing rows zero) and PL = 2 planes for F16X3 (hi = fp16(w), lo = fp16(w - hi)),
 * else 1; `out` holds PL * NB*32 * K 16-bit values.  Operators below accept such buffers for their KxK convolutions (NOT the
 * 1x1 / tiny convs: wc1, wf1, flow-head w2, mask-head w2 stay raw fp32) when CRAFT_W_PACKED is or-ed into prec: a wave then
 * fetches the B operand of each 32x32x16 MFMA with one coalesced 1 KiB load and the K loop runs without per-tile barriers. */
#define CRAFT_W_PACKED 0x100
/* craft_conv2d_nhwc / craft_conv2d_nhwc2, prec = F16X3 | CRAFT_W_PACKED | CRAFT_CONV_W16: only the hi plane of the packed weights is
 * used (the weights rounded to fp16, the input keeps both planes): two MFMAs per product instead of three.  For the INPUT-gradient
 * convolutions of a training policy that asks for it (craft_amd.hip.Precision role `wgx`); stride-1 KxK with 5 or 9 taps, ignored
 * elsewhere. */
#define CRAFT_CONV_W16 0x400
/* craft_linear / craft_linear_t (fragment form), prec | CRAFT_W_PACKED: w is craft_pack_weights(w, rows = cout, K = cin rounded up to 32 with zero
 * columns, prec) instead of fp32 [cout][cin] -- the short-K products of the refinement loop then run on k_gemm_rows_wf (weights L2 ->
 * MFMA registers, activations converted once per 64-wide K chunk).  craft_motion_encoder, prec | CRAFT_W1X1_PACKED: wc1 (the 1x1
 * convc1, update.py:80) handed over the same way: craft_pack_weights(wc1, 256, round_up(cor_planes, 32), prec). */
#define CRAFT_W1X1_PACKED 0x800
/* craft_conv2d_nhwc2 with a bias_field, prec | CRAFT_CONV_FIELD_COL0(c) (c a multiple of 32, < 8192): the field is added to output
 * columns >= c only; columns below get the bare convolution.  The backward of SepConvGRU's second pass adds the first pass's
 * gradient of [motion features | aggregate] (columns 128..383 of a 384-wide field) while its own hidden-state columns start fresh. */
#define CRAFT_CONV_FIELD_COL0(c) ((((c) / 32) & 0xff) << 16)
#define CRAFT_STATS_REPLICAS 64
#define CRAFT_ATTN_CHUNK_KEYS 1024
int craft_pack_weights(const float* w, int rows, int K, int prec, void* out, void* stream);

/* nn.Conv2d (stride 1, "same" zero padding KH/2, KW/2) + bias + optional ReLU on tokens: x [B*H*W][cin] (row
 * stride ldx, cin % 32 == 0), w packed [cout][KH][KW][cin] (raw fp32, or craft_pack_weights output with
 * CRAFT_W_PACKED or-ed into prec when KH*KW > 1), y [B*H*W][cout].  The building block of the operators below. */
int craft_conv2d_nhwc(const float* x, long ldx, int cin, const float* w, const float* bias, int cout, int KH, int KW,
                      int act, float* y, long ldy, int B, int H, int W, int prec, void* stream);

/* ---- CNN encoder building blocks (BasicEncoder / ResidualBlock, extractor.py:6-64, 124-196) ----
 * craft_conv2d_nhwc_ex: craft_conv2d_nhwc plus (a) stride 2 (x is [B*Hin*Win][cin], y [B*Hout*Wout][cout], padding
 * KH/2); (b) in_norm [B][cin][2] = (mean, rstd): the input is read as relu((x - mean)*rstd), i.e. the InstanceNorm
 * + ReLU of the producing conv applied lazily (stride 1, KH*KW > 1 only); (c) stats [CRAFT_STATS_REPLICAS][B][cout][2]
 * doubles: += (sum, sum^2) of the biased output per (image, channel) for the consumer's lazy InstanceNorm (zero them
 * first).  The table is replicated (a block adds into replica blockIdx % CRAFT_STATS_REPLICAS) because thousands of
 * blocks hit the same B*cout cells at once and same-address atomics serialise in L2.
 * craft_stats_finalize: n = B*cout populations of `count` samples: replicas summed, (sum, sum^2) -> (mean,
 * 1/sqrt(var + eps)).
 * craft_residual_relu: out = relu(fx(x) + fy(y)), fx = identity or (x-mean)*rstd (xnorm), fy = identity / relu /
 * relu((y-mean)*rstd) (ynorm, y_relu bit 0; bit 1 = ReLU on fx): the tail of ResidualBlock.forward with both
 * norms applied lazily.
 * craft_stem_conv7x7: BasicEncoder.conv1 (7x7, stride 2, pad 3, 3 -> 64; extractor.py:139,181) fused with the input
 * normalisation 2*(x/255)-1 (network.py:169-173): image NCHW [B][3][H][W] raw 0..255, w packed [147 = (ky*7+kx)*3
 * + c][64] (weight.permute(2,3,1,0)), out tokens [B][(H/2)*(W/2)][64] = act(conv + bias); stats as above. */
/* craft_pack_weights straight from nn.Conv2d's weight layout [Cout][Cin][KH][KW] (16-bit / f16x3 precisions): w0 (cout0 rows) and
 * optionally w1 (cout1 rows) concatenated along Cout; input channels [a0, a1) u [b0, b1) selected (b0 == b1: one range).
 * transposed = 0: the forward operand (rows = output channels, K = KH*KW*selected channels; selected count % 32 == 0).
 * transposed = 1: the operand of the INPUT-gradient convolution (rows = selected input channels, K = KH*KW*round_up(Cout, 32),
 * taps flipped): W'[ci][KH-1-ky][KW-1-kx][co] = W[co][ci][ky][kx].  out: planes * round_up(rows, 32) * K 16-bit elements. */
int craft_pack_conv_weights(const float* w0, int cout0, const float* w1, int cout1, int Cin, int KH, int KW, int a0, int a1, int b0, int b1,
                            int transposed, int prec, void* out, void* stream);
/* The same re-layout for MANY weights in one launch (round 5): a training step re-packs every convolution weight -- forward and
 * transposed forms -- after each optimizer update (what `optimizer.step()` + the next forward's cuDNN filter transforms are in the
 * reference, train.py:231-236).  The job table lives in device memory: craft_pack_conv_job_fill writes one record (craft_pack_conv_job_bytes
 * each) into HOST memory and returns the job's block count (< 0: -error); the caller copies the records and the exclusive prefix sums
 * of the block counts (n + 1 ints) to the device once and calls craft_pack_conv_weights_batch after every update. */
int craft_pack_conv_job_bytes(void);
int craft_pack_conv_job_fill(void* job, const float* w0, int cout0, const float* w1, int cout1, int Cin, int KH, int KW, int a0, int a1, int b0,
                             int b1, int transposed, int prec, void* out);
int craft_pack_conv_weights_batch(const void* jobs_dev, const int* first_block_dev, int n, int total_blocks, void* stream);
/* craft_conv2d_nhwc over the virtual channel concatenation [x0 (c0 channels, row stride ld0) | x1 (c1, ld1)] (c1 = 0: x0 alone):
 * the conv input of SepConvGRU's q gate, cat([r*h, x]) (update.py:54, :61), without materialising the cat.  c0, c1 multiples of 32.
 * Exactly one of bias [cout] / bias_field [B*H*W][ld_bf] (a per-pixel bias: the hoisted, iteration-invariant share of a convolution,
 * as craft_sepconv_gru_context produces it) is non-NULL. */
int craft_conv2d_nhwc2(const float* x0, long ld0, int c0, const float* x1, long ld1, int c1, const float* w, const float* bias,
                       const float* bias_field, long ld_bf, int cout, int KH, int KW, int act, float* y, long ldy, int B, int H, int W, int prec,
                       void* stream);
/* craft_conv2d_nhwc with the tail of a ResidualBlock fused into the epilogue (round 5): y = relu(res + act(conv(x) + bias)), res fp32
 * tokens [B*H*W][cout] with row stride ldr -- `self.relu(x + y)` of extractor.py:56-63 for the encoder whose BatchNorm is folded into the
 * weights (cnet, eval): the standalone craft_residual_relu pass (read x, read y, write out) disappears.  Stride 1. */
int craft_conv2d_nhwc_res(const float* x, long ldx, int cin, const float* w, const float* bias, int cout, int KH, int KW, int act, const float* res,
                          long ldr, float* y, long ldy, int B, int H, int W, int prec, void* stream);
/* craft_conv2d_nhwc2 (no activation) whose epilogue also applies the ReLU backward of the layer BELOW: y = mask > 0 ? conv + bias(_field) : 0,
 * mask fp32 tokens [B*H*W][cout] (row stride ldm) = that layer's saved forward output.  The input-gradient convolutions of a chain
 * conv -> ReLU -> conv (autograd of update.py:80-87, :12-16) hand their result straight to the next one: no separate craft_act_bwd pass. */
int craft_conv2d_nhwc2_mask(const float* x0, long ld0, int c0, const float* x1, long ld1, int c1, const float* w, const float* bias,
                            const float* bias_field, long ld_bf, int cout, int KH, int KW, const float* mask, long ldm, float* y, long ldy, int B,
                            int H, int W, int prec, void* stream);
int craft_conv2d_nhwc_ex(const float* x, long ldx, int cin, int Hin, int Win, const float* in_norm, const float* w,
                         const float* bias, int cout, int KH, int KW, int stride, int act, float* y, long ldy, int B,
                         int Hout, int Wout, double* stats, int prec, void* stream);
int craft_stem_conv7x7(const float* image, const float* w, const float* bias, int act, int B, int H, int W,
                       float* out, double* stats, void* stream);
/* The same stem on the matrix cores (bf16 / fp16 / F16X3): w_packed = craft_pack_weights(rows 64, K 192, prec) of the
 * weight matrix re-ordered to k = (ky*3 + c)*8 + kx (kx = 7 and k >= 168 zero), so that a k-group of 8 is one run of
 * 8 consecutive pixels of an input row.  Same outputs / statistics as craft_stem_conv7x7. */
int craft_stem_conv7x7_mfma(const float* image, const void* w_packed, const float* bias, int act, int B, int H, int W,
                            float* out, double* stats, int prec, void* stream);
/* The same with the batch taken from TWO image tensors: images [0, Ba) from image_a [Ba][3][H][W], images [Ba, B) from image_b -- fnet
 * runs both frames of every pair as one batch (network.py:176-180: self.fnet([image1, image2]) concatenates them inside the encoder,
 * extractor.py:171-176) without a concatenated copy. */
int craft_stem_conv7x7_mfma_pair(const float* image_a, int Ba, const float* image_b, const void* w_packed, const float* bias, int act, int B, int H,
                                 int W, float* out, double* stats, int prec, void* stream);
int craft_stats_finalize(const double* sums, long n, double count, float eps, float* mean_rstd, void* stream);
int craft_residual_relu(const float* x, long ldx, const float* xnorm, const float* y, long ldy, const float* ynorm,
                        int y_relu, int B, int HW, int C, float* out, long ldo, void* stream);

/* BasicMotionEncoder.forward (update.py:79-87).  corr tokens [B*N][cor_planes] (row stride ldc), flow tokens
 * [B*N][2].  Conv weights are packed [Cout][KH][KW][Cin] (weight.permute(0,2,3,1)); wf1 is packed
 * [7*7*2][128] (weight.permute(2,3,1,0)) -- or, with CRAFT_W_PACKED and a 16-bit / F16X3 precision, craft_pack_weights(rows 128,
 * K 128, prec) of the matrix re-ordered to k = ky*16 + kx*2 + c (kx = 7 and k >= 112 zero), which runs the layer on the matrix
 * cores (a k-group of 8 is then one run of 8 consecutive floats of a flow-patch row).  Output: 128 channels (126 conv + 2 flow)
 * at out (row stride ldo).  ws: 640 floats per pixel.
 * flow_stream / flow_done (both NULL: everything runs in order on `stream`): a caller-owned hipStream_t and hipEvent_t.  The
 * flow branch (convf1 -> convf2) is independent of the correlation branch (convc1 -> convc2) until the last convolution, so
 * it is enqueued on flow_stream, flow_done is recorded behind it and `stream` waits for that event before the last
 * convolution.  The CALLER orders flow_stream behind the producers of `flow` and `ws` (an event wait on its own stream) --
 * the library creates no stream, event or other state of its own. */
int craft_motion_encoder(const float* corr, long ldc, int cor_planes, const float* flow, const float* wc1,
                         const float* bc1, const float* wc2, const float* bc2, const float* wf1,
                         const float* bf1, const float* wf2, const float* bf2, const float* wcv,
                         const float* bcv, int B, int H8, int W8, float* out, long ldo, float* ws, int prec,
                         void* flow_stream, void* flow_done, void* stream);

/* SepConvGRU.forward (update.py:49-64) in place on hx = [h (128) | x (cx)] (row stride ldhx): 1x5 pass then
 * 5x1 pass; wzr*: convz and convr stacked on the output axis and packed [256][KH][KW][128+cx], wq*:
 * [128][KH][KW][128+cx].  ws: 256 floats per pixel (z and r*h). */
int craft_sepconv_gru(float* hx, long ldhx, int cx, const float* wzr1, const float* bzr1, const float* wq1,
                      const float* bq1, const float* wzr2, const float* bzr2, const float* wq2,
                      const float* bq2, int B, int H8, int W8, float* ws, int prec, void* stream);

/* SepConvGRU with its iteration-invariant part hoisted.  The context features `inp` (cc channels of x) are the same
 * in every refinement iteration and a convolution is linear in its input channels, so their contribution to the six
 * gate convolutions is computed once per forward pass by craft_sepconv_gru_context into
 *   fields [B*N][768] = [z|r pass 1 (256) | q pass 1 (128) | z|r pass 2 (256) | q pass 2 (128)]   (biases included)
 * from the const-channel slices of the weights ([cout][KH][KW][cc], optionally packed).  craft_sepconv_gru_step then
 * runs one GRU update in place on hx = [h (128) | ... | v (cv channels at column voff)] with weights over the
 * [h | v] channels only ([cout][KH][KW][128+cv]) and `fields` as per-pixel bias: 25 % fewer MACs per iteration,
 * same result up to fp32 summation order.  ws: 256 floats per pixel. */
int craft_sepconv_gru_context(const float* inp, long ldi, int cc, const float* wzr1, const float* bzr1,
                              const float* wq1, const float* bq1, const float* wzr2, const float* bzr2,
                              const float* wq2, const float* bq2, int B, int H8, int W8, float* fields, int prec,
                              void* stream);
int craft_sepconv_gru_step(float* hx, long ldhx, int voff, int cv, const float* wzr1, const float* wq1,
                           const float* wzr2, const float* wq2, const float* fields, int B, int H8, int W8, float* ws,
                           int prec, void* stream);

/* FlowHead.forward (update.py:15-16) fused with coords1 += delta (network.py:247) and flow = coords1 - coords0.
 * h: hidden state tokens (row stride ldh); w1 packed [256][3][3][128]; w2 packed [2][3][3][256];
 * delta may be NULL.  ws: 256 floats per pixel. */
int craft_flow_head(const float* h, long ldh, const float* w1, const float* b1, const float* w2,
                    const float* b2, int B, int H8, int W8, float* coords1, const float* coords0, float* flow,
                    float* delta, float* ws, int prec, void* stream);

/* mask head (update.py:124-127, :161): mask = 0.25 * conv1x1(relu(conv3x3(h))) -> tokens [B*N][576].
 * w0 packed [256][3][3][128], w2 [576][256].  ws: 256 floats per pixel. */
int craft_mask_head(const float* h, long ldh, const float* w0, const float* b0, const float* w2,
                    const float* b2, int B, int H8, int W8, float* mask, float* ws, int prec, void* stream);

/* Evaluation metrics of the harness (evaluate.py:529 EPE, :578-598 1/3/5 px and magnitude bins, :833-841 KITTI Fl
 * outliers): pred, gt NCHW [B][2][H][W] fp32; valid [B][H][W] (>= 0.5 counts) or NULL; (gt_off_x, gt_off_y) is added to
 * gt only for the magnitude (the shift experiments, evaluate.py:534).  out16 += { sum epe, #valid, #epe<1, #epe<3,
 * #epe<5, #(epe>3 && epe/mag>0.05), sum epe per magnitude bin [0,1) [1,10) [10,20) [20,30) [30,inf), counts per bin }
 * as doubles (zero it first; accumulates over calls).  max_gt_mag > 0 also drops pixels with |gt| >= max_gt_mag (the
 * MAX_FLOW rule of the training metrics, train.py:53). */
int craft_flow_metrics(const float* pred, const float* gt, const float* valid, int B, int H, int W, float gt_off_x,
                       float gt_off_y, float max_gt_mag, double* out16, void* stream);

/* ---- training step around the path (the backward kernels of the model itself do not exist yet) ----
 * craft_flow_l1_loss: one term of sequence_loss (train.py:44-61): *loss += weight * mean(valid * |pred - gt|) over all
 *   B*2*H*W elements, valid = (valid >= 0.5) & (|gt| < max_flow); grad_pred (or NULL) = d(term)/d(pred).  The caller
 *   loops over the T predictions with weight = gamma^(T-1-i).
 * craft_sumsq: *out += sum x^2 (double): the global gradient norm of clip_grad_norm_ (train.py:234).
 * craft_adamw_step: torch.optim.AdamW semantics (decoupled decay, bias correction with `step` >= 1) over flat buffers:
 *   g = grad * grad_mul (e.g. 1/world_size after a summed all-reduce), then if grad_sumsq != NULL and max_norm > 0:
 *   g *= min(1, max_norm / (sqrt(*grad_sumsq) * grad_mul + 1e-6)) -- the clip coefficient, computed on the device.  With grad_sumsq
 *   given, a non-finite norm skips the update (whether or not max_norm > 0).
 * craft_loss_scale_update + craft_adamw_step_dyn: torch.cuda.amp.GradScaler (train.py:215, 231-238) without a host read-back.
 *   `state` is a 32-byte device record, 8 x 32-bit words:
 *     {float scale; int growth_tracker; int opt_step; int skipped; int found_inf; float gm; float bc1; float bc2_sqrt}
 *   (the caller initialises scale and opt_step, zeroes the rest).  The flat gradient holds scale x (sum over ranks) the gradient and
 *   *grad_sumsq its sum of squares.  craft_loss_scale_update: non-finite -> found_inf = 1, skipped += 1, scale *= backoff, tracker = 0;
 *   finite -> found_inf = 0, opt_step += 1, gm = grad_mul / scale * min(1, max_norm / (norm + 1e-6)) (max_norm <= 0: no clipping),
 *   bc1 / bc2_sqrt = AdamW's bias corrections for opt_step, tracker += 1 and scale *= growth when tracker reaches growth_interval
 *   (growth_interval <= 0: static scale).  craft_adamw_step_dyn: craft_adamw_step reading {found_inf, gm, bc1, bc2_sqrt} from the
 *   record -- a skipped step leaves weights, moments and the optimizer's step count untouched, as GradScaler.step does. */
int craft_flow_l1_loss(const float* pred, const float* gt, const float* valid, int B, int H, int W, float weight, float max_flow,
                       double* loss, float* grad_pred, void* stream);
int craft_sumsq(const float* x, long n, double* out, void* stream);
int craft_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int step, float grad_mul, const double* grad_sumsq,
                     float max_norm, void* stream);
int craft_loss_scale_update(const double* grad_sumsq, void* state, float grad_mul, float max_norm, float beta1, float beta2, float growth,
                            float backoff, int growth_interval, void* stream);
int craft_adamw_step_dyn(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, const void* state, void* stream);

/* ==== training: backward of the hot path (train.py:228-236 `loss.backward()` through network.py:164-267) =================
 * The reference gets its backward from autograd over PyTorch ops; here every operator's gradient is a kernel, bound as the
 * backward of an autograd.Function by craft_amd/autograd.py.  The dense contractions are ONE general strided batched GEMM
 * (dX = dY W, dW = dY^T X, dP = dO V^T, dV = P^T dO, dQ = dS K, dK = dS^T Q are all instances), the convolution weight
 * gradient is its sibling with the tap shift in the loader, and the convolution input gradient is the FORWARD convolution
 * (craft_conv2d_nhwc) with flipped, transposed weights.  Everything else is element-wise / row-wise and HBM-bound.
 *
 * craft_gemm:  C[z][m][n] = alpha * sum_k A(z,m,k) * B(z,n,k)  (+ C[z][m][n] when accumulate), z = z0*zdiv + z1 < batch,
 *   A(z,m,k) at A + z0*a_bs0 + z1*a_bs1 + m*a_sm + k*a_sk with a_sk == 1 (k contiguous: rows operand) or a_sm == 1 (k-major:
 *   a transposed operand read in place); likewise B with (b_sn, b_sk).  C row-major with row stride ldc.  A rows operand has
 *   its leading dimension and batch strides in multiples of 4 floats, a 16-byte aligned base and K % 4 == 0; a k-major operand has
 *   no alignment requirement (batch strides of one element are legal: shifted views of one buffer); M, N arbitrary.  ksplit: K is cut into that many
 *   ranges whose partial products are added with atomics (needs accumulate = 1 and a zero-filled or running C); 0 = choose
 *   so that the grid fills the chip (weight gradients: M x N is tiny, K = all rows).  prec as craft_linear.
 * craft_conv2d_wgrad: dW[co][ky][kx][ci] += sum_pix dY[pix][co] * X[pix + (ky-KH/2, kx-KW/2)][ci]  (stride 1, zero padding;
 *   x [B*H*W][cin] row stride ldx, dy [B*H*W][cout] row stride ldy; dW in the packed [cout][KH][KW][cin] layout, ACCUMULATED).
 *   db (or NULL): db[co] += sum_pix dY[pix][co], the bias gradient, added by the same launch.
 *   ws (or NULL) / ws_floats: scratch for the split-K partial sums; with >= 32 * cout*KH*KW*cin floats every split stores its
 *   partial tile with plain writes and one pass folds them into dW, else the partial sums are added with fp32 atomics.
 * craft_colsum: out[c] += sum_r x[r][c]   (bias gradients).
 * craft_act_fwd / craft_act_bwd: y = scale * act(x);  dx = scale * dy * act'(.) evaluated from the UNSCALED output y / scale the
 *   caller kept (relu: y > 0, tanh: 1 - y^2, sigmoid (CRAFT_ACT_SIGMOID = 3): y (1 - y)); C % 4 == 0.
 * craft_dropout: y[i] = x[i] * keep_i / (1 - p), keep_i = (hash(seed, i) >= p): nn.Dropout in training mode (setrans.py:553-557,
 *   :791-795) with a counter-based generator, so the backward is the same call on the gradient. */
#define CRAFT_ACT_SIGMOID 3
int craft_gemm(const float* A, long a_sm, long a_sk, long a_bs0, long a_bs1, const float* B, long b_sn, long b_sk, long b_bs0,
               long b_bs1, float* C, long ldc, long c_bs0, long c_bs1, int zdiv, int batch, int M, int N, int K, float alpha,
               int accumulate, int ksplit, int prec, void* stream);
int craft_conv2d_wgrad(const float* x, long ldx, int cin, const float* dy, long ldy, int cout, int KH, int KW, int B, int H, int W,
                       float* dW, float* db, float* ws, long ws_floats, int prec, void* stream);
int craft_colsum(const float* x, long ld, long rows, int C, float* out, void* stream);
/* count contiguous fp32 device tensors src[i] (n[i] elements) -> dst + dst_off[i], in ONE launch (src / n / dst_off are HOST arrays): the
 * parameter gradients of a training step into the optimizer's flat gradient buffer -- what the reference's optimizer.step() reads
 * parameter by parameter (train.py:231-236) and torch._foreach_copy_ turns into one copy launch per tensor.  chlast (HOST array or NULL):
 * chlast[i] = cin * 1024 + taps marks src[i] as a convolution weight gradient in the layout the weight-gradient kernels write,
 * [cout][taps][cin]; it is stored as nn.Conv2d's [cout][cin][taps] (0: plain copy). */
int craft_multi_copy(const void* const* src, const long* n, const long* dst_off, const long* chlast, int count, float* dst, void* stream);
/* Packed-operand weight gradients (round 3; craft_amd/csrc/kernels_gemm_pk.hip): the same contraction as craft_conv2d_wgrad /
 * craft_gemm's dW = dY^T X for the 16-bit MFMA modes, with the fp32 -> fp16-plane split taken OUT of the K loop.
 * craft_pack_operand: tokens x [rows][C] (fp32, row stride ldx, C % 4 == 0) -> out[plane][ceil(C/32)][rows_p][32] 16-bit
 *   (prec = CRAFT_PREC_F16X3: two fp16 planes hi, lo; CRAFT_PREC_F16 / CRAFT_PREC_BF16: one plane; channels >= C are zero).
 *   Plain form (B = 0): pack row r holds source row r - guard (zero outside [0, rows)).  Spatial form (B > 0, rows = B*H*W pixels):
 *   pack row r holds pixel (b, y, x) of the ZERO-PADDED grid [B][H + 2 padH][W + 2 padW] at index r - guard, so that a convolution
 *   tap is one constant row shift dy * (W + 2 padW) + dx.  rows_p >= the rows the consumer reads, caller-allocated
 *   (planes * ncg_total * rows_p * 64 bytes).  cg_off / ncg_total: this source fills the 32-channel groups [cg_off, cg_off + ceil(C/32))
 *   of a pack of ncg_total groups -- several calls build the pack of a channel concatenation without a torch.cat.  colsum (or NULL): colsum[c] += sum_r x[r][c] (a convolution's bias gradient rides
 *   on the pack of its dY).  tail (spatial form, else 0): extra zero columns behind every grid row, Wp = W + 2 padW + tail -- with H = 1,
 *   padW = 0 the pack holds B batches of W rows, each padded to W + tail rows (the per-batch operands of craft_gemm_pk).
 * craft_wgrad_pk: dW[co][tap][ci] += sum_{s < nseg} sum_{k < K} dYp[s][guard + k][co] * Xp[s][guard + k + shift(tap)][ci], shift(tap) =
 *   (tap / KW - KH/2) * Wp + (tap % KW - KW/2); cout, cin multiples of 32 (the packs' channel groups), K % 32 == 0, guard >=
 *   the largest |shift|, both packs built with the same geometry and prec.  dW in the [cout][KH][KW][cin] layout, ACCUMULATED
 *   (split-K partial sums are added with fp32 atomics).  KH = KW = 1: nn.Linear's weight gradient.  dYp / Xp: HOST arrays of nseg device
 *   pointers -- the packs of the nseg calls of one layer in a pass (the update block runs every layer once per refinement iteration:
 *   their weight gradient is ONE launch over the concatenated K, one atomic epilogue per pass instead of one per iteration).
 *   Xp1 (or NULL) / cin0: the X operand as the channel concatenation of TWO packs over the same rows -- input channels [0, cin0) from
 *   Xp[s], [cin0, cin) from Xp1[s] (cat([h, x]) / cat([r*h, x]) of SepConvGRU: x is packed once per pass and shared by both gates).
 *   prec | CRAFT_WGRAD_X_PREC(CRAFT_PREC_F16) with prec = F16X3: the X packs hold ONE fp16 plane (X rounded to fp16) while dY keeps
 *   its hi / lo planes -- two MFMAs per product instead of three (dW to ~2e-4 relative instead of ~2e-5). */
#define CRAFT_WGRAD_X_PREC(p) (((p) + 1) << 8)
int craft_pack_operand(const float* x, long ldx, int C, long rows, int B, int H, int W, int padH, int padW, long guard, long rows_p,
                       int prec, void* out, int cg_off, int ncg_total, float* colsum, int tail, void* stream);
/* n craft_pack_operand calls as ONE launch (a training iteration packs ~14 convolution inputs of a few MB each: separate launches are
 * overhead-bound).  descs: HOST array of n x 17 longs per tensor, the arguments of craft_pack_operand in order with pointers as integers:
 * (x, ldx, C, rows, B, H, W, padH, padW, guard, rows_p, prec, out, cg_off, ncg_total, colsum, tail). */
int craft_pack_operands(const long* descs, int n, void* stream);
int craft_wgrad_pk(const void* const* dYp, const void* const* Xp, const void* const* Xp1, int cin0, int nseg, long dy_rows_p, int cout, long x_rows_p,
                   int cin, long guard, long K, int KH, int KW, int Wp, float* dW, int prec, void* stream);

/* Batched GEMM over packed operands (round 3; craft_amd/csrc/gemm_pkb.inc.hpp) -- the attention products of the training pass
 * (autograd of setrans.py:373-384, :520-557; update.py:143-149) with nothing but copies and MFMAs in the K loop:
 *   C[z][m][n] = alpha * sum_{k < K} A_z[m, k] * B_z[n, k],   z = outer * inner + inner_index < nbatch,  C_z = C + outer * c_outer + inner_index * c_inner
 * A / B: packs [plane][channel group][row][32] (craft_pack_operand(s), craft_attn_softmax_fwd's Ppk), described by 9 longs each:
 *   {kind, rows_p, ncg, row0, row_outer, row_inner, cg0, cg_outer, cg_inner}: kind 0 = K runs down the ROWS of the pack and M (N) over
 *   its channels, kind 1 = K runs over the CHANNELS and M (N) down the rows; rows_p / ncg: rows and channel groups of the pack (its
 *   strides); batch z reads from row row0 + outer * row_outer + inner_index * row_inner and channel group cg0 + ...
 *   With P packed as (rows i, channels j):  O = P V -> (P kind 1, V kind 0);  dV = P^T dO -> (0, 0);  dP = dO V^T -> (1, 1).
 * K % 32 == 0 with ZEROS in the K padding of both packs; the M / N padding may hold anything finite or not (never stored).
 * Kind pairs (0,0), (1,0), (1,1); prec CRAFT_PREC_F16X3 / F16 / BF16 = the mode the packs were written in. */
/* prec | CRAFT_PK_CBLK(s) (s >= 5, c_inner == 2^s): the N output columns are stored in blocks of 2^s, consecutive blocks inner * 2^s elements
 * apart -- C[z][m][n] at outer * c_outer + inner_index * 2^s + m * ldc + (n >> s) * inner * 2^s + (n & (2^s - 1)): the batch's `inner` entries
 * interleave inside every column block, e.g. dV of all T iterations as [B][N][T][M][Cv] from batches (b, m) and columns (t, c), so that one
 * iteration's [B*N][M*Cv] slice is a strided view instead of a copy. */
#define CRAFT_PK_CBLK_SHIFT 8
#define CRAFT_PK_CBLK(s) ((s) << CRAFT_PK_CBLK_SHIFT)
int craft_gemm_pk(const void* A, const long* a_desc, const void* B, const long* b_desc, float* C, long ldc, long c_outer, long c_inner,
                  int inner, int nbatch, int M, int N, int K, float alpha, int prec, void* stream);

int craft_act_fwd(const float* x, long ldx, float* y, long ldy, long rows, int C, int act, float scale, void* stream);
int craft_act_bwd(const float* dy, long lddy, const float* y, long ldy, float* dx, long lddx, long rows, int C, int act, float scale,
                  void* stream);
/* craft_act_bwd with two more inputs: dy2 (optional, row stride lddy2) -- the gradient that flows back is dy + dy2 -- and zero_tail: the
 * last zero_tail of the C channels carry no gradient (dx = 0 there: the two pass-through flow channels of BasicMotionEncoder's output,
 * update.py:93-94).  One launch for what a training step did as add_ + craft_act_bwd + a slice fill per refinement iteration. */
int craft_act_bwd2(const float* dy, long lddy, const float* dy2, long lddy2, const float* y, long ldy, float* dx, long lddx, long rows, int C,
                   int act, float scale, int zero_tail, void* stream);
int craft_dropout(const float* x, float* y, long n, float p, unsigned long long seed, void* stream);

/* backward of craft_tokens with a token-major source: y = LayerNorm?(act(x)), x / dy / dx rows of C <= 256 channels. */
int craft_tokens_bwd(const float* x, long ldx, const float* dy, long lddy, float* dx, long lddx, long rows, int C, int act, int do_ln,
                     void* stream);

/* Attention probabilities from MATERIALISED scores (training form of craft_attn_probs; setrans.py:520-551), in place:
 *   S [B][M][N][ld] (scaled Q K^T from craft_gemm; ld % 32 == 0) -> P = softmax_j(clamp?(S) + pos_w*pb + mask), columns [N, ld) = 0.
 *   clampbits (or NULL): [B*M*N][ld/32] words, bit j = "score (i, j) was clamped" -- written only when the clamp is active.
 * craft_attn_softmax_bwd: dP (gradient w.r.t. P, overwritten with dS) -> dS = P (dP - sum_j dP P), zero where clamped;
 *   dtab_rep [CRAFT_STATS_REPLICAS][(2R+1)^2] += pos_w * dS over the positional window (zero it first; craft_reduce_replicas
 *   folds the replicas into the table's gradient).
 * Pdrop / drop_p / seed (Pdrop NULL: none): the dropout of the probabilities (setrans.py:553-557) fused into both passes -- the forward
 *   also writes Pdrop = craft_dropout(P, drop_p, seed) (same mask: the flat element index), the backward (drop_p > 0) takes dP as the
 *   gradient w.r.t. Pdrop and applies the mask while it reads the row (two 1 GB passes less per attention at 368x496, batch 8).
 * Ppk (or NULL) / pk_rows / pk_np / pk_prec: the (dropped, if drop_p > 0) probabilities also as a packed operand of craft_gemm_pk,
 *   [plane][ld / 32][pk_rows][32] in mode pk_prec with batch z = b * M + m in rows [z * pk_np, (z + 1) * pk_np), pk_np >= N a multiple
 *   of 32, rows >= N of a batch zero: rows = query i, channels = key j.
 * dSpk (craft_attn_softmax_bwd; or NULL): dS leaves as such a pack INSTEAD of fp32 over dP (same layout and arguments as Ppk) -- the
 *   operand of dQ = dS K and dK = dS^T Q on craft_gemm_pk. */
int craft_attn_softmax_fwd(float* S, long ld, int B, int M, int H8, int W8, const float* pos_tab, int R, float pos_w, int mask_radius,
                           const unsigned* clamp_ord, unsigned* clampbits, float* Pdrop, float drop_p, unsigned long long seed,
                           void* Ppk, long pk_rows, int pk_np, int pk_prec, void* stream);
int craft_attn_softmax_bwd(const float* P, float* dP, long ld, int B, int M, int H8, int W8, int R, float pos_w,
                           const unsigned* clamp_ord, const unsigned* clampbits, float* dtab_rep, float drop_p, unsigned long long seed,
                           void* dSpk, long pk_rows, int pk_np, int pk_prec, void* stream);
int craft_reduce_replicas(const float* rep, int nrep, int n, float* out, void* stream);

/* gma.Attention's relative-position scores (RelPosEmb, gma.py:21-50, :84-98) added to materialised scores S [BZ][N][ld], BZ = B*heads:
 *   S[z][(x,y)][(u,v)] += w * (Hs[z][(x,y)][u - x + H8 - 1] + Ws[z][(x,y)][v - y + W8 - 1]), x / u rows, y / v columns;
 *   Hs [BZ][N][>= 2 H8 - 1] (row stride ldh) = q . E_h rows of the offsets -(H8-1) .. H8-1, Ws likewise (products of the caller).
 * craft_relpos_bwd: dHs[z][i][d] = w * sum_v dS[z][i][(u, v)] for d = u - x + H8 - 1, 0 for the nh - H8 unreachable offsets; dWs likewise. */
int craft_relpos_add(float* S, long ld, int BZ, int H8, int W8, const float* Hs, long ldh, const float* Ws, long ldw, float w, void* stream);
int craft_relpos_bwd(const float* dS, long ld, int BZ, int H8, int W8, float* dHs, long ldh, int nh, float* dWs, long ldw, int nw, float w,
                     void* stream);

/* Correlation volume from MATERIALISED scores (training form of craft_corr_build; corr.py:191-199, setrans.py:520-550):
 *   S [B][M][N][ld] -> c0 [B*N][N] = sum_m s_m softmax_m(w s_m), s_m = clamp?(S_m) + pos_w*pb; sums[b] += (sum c, sum c^2).
 *   w: DEVICE pointer to attn_softaggr.feat2score.weight (no host read-back).  craft_corr_finish then builds the pyramid and
 *   (mean, rstd) exactly as in inference, and craft_corr_lookup samples it.
 * craft_corr_lookup_bwd: gradient of the lookup output (tokens, same column layout as craft_corr_lookup) scattered into
 *   G0..G3, the gradients of the NORMALISED pyramid levels (same shapes as the pyramid, zero them first; atomics).
 * craft_corr_pyramid_bwd: folds G1..G3 into G0 (2x2 average pooling backward, floor sizes) and reduces gstats[b] += (sum G0,
 *   sum G0 * c_hat) for the global-LayerNorm backward (zero gstats first).
 * craft_corr_pool_bwd: (G0, gstats) -> dS written over S (LayerNorm backward over all N*N entries, mode-pooling backward, clamp
 *   mask); dtab_rep as above; *dw += gradient of the pooling weight (double). */
int craft_corr_pool_fwd(const float* S, long ld, int B, int M, int H8, int W8, const float* pos_tab, int R, float pos_w, const float* w,
                        const unsigned* clamp_ord, float* c0, double* sums, void* stream);
int craft_corr_lookup_bwd(const float* dout, long ldo, const float* coords, float* G0, float* G1, float* G2, float* G3, int levels, int B,
                          int H8, int W8, int radius, int lvl_stride, int col_off, void* stream);
int craft_corr_pyramid_bwd(float* G0, const float* G1, const float* G2, const float* G3, const float* c0, const float* mu_rstd, int B,
                           int H8, int W8, double* gstats, void* stream);
int craft_corr_pool_bwd(float* S, long ld, int B, int M, int H8, int W8, const float* pos_tab, int R, float pos_w, const float* w,
                        const unsigned* clamp_ord, const float* c0, const float* G0, const float* mu_rstd, const double* gstats,
                        int do_norm, float* dtab_rep, double* dw, void* stream);

/* backward of craft_mode_pool_ln: dy -> dO [B][M][N][C], dx [B*N][C] (row stride lddx), and dw_rep [CRAFT_STATS_REPLICAS][C+1] +=
 * (d w_agg [C], d skip_coeff) (zero it first; craft_reduce_replicas).  C = 128 or 256, M <= 4 (as craft_mode_pool_ln). */
int craft_mode_pool_ln_bwd(const float* O, const float* x, long ldx, const float* w_agg, const float* skip_coeff, const float* dy,
                           long lddy, int B, int N, int M, int C, float* dO, float* dx, long lddx, float* dw_rep, void* stream);

/* backward of craft_convex_upsample: dup NCHW [B][2][8*H8][8*W8] -> dmask [B*N][576] (row stride lddm) and dflow [B*N][>= 2] +=
 * (row stride lddf: 2 for a dense gradient, 32 when it is written straight into the flow head's padded output gradient). */
int craft_convex_upsample_bwd(const float* mask, long ldm, const float* flow, const float* dup, int B, int H8, int W8, float* dmask,
                              long lddm, float* dflow, long lddf, void* stream);

/* The coordinate bookkeeping of one refinement iteration of the training forward (network.py:232-234, :247) in one launch:
 * flow [rows][2] = coords1 - coords0; flow32 (or NULL) [rows][32] the same, zero-padded (the operand of convf1's weight gradient);
 * coords1_copy (or NULL) [rows][2] = coords1 (craft_flow_head updates it in place). */
int craft_flow_tokens(const float* coords1, const float* coords0, long rows, float* flow, float* flow32, float* coords1_copy, void* stream);

/* SepConvGRU gates as separate stages (update.py:55-63), C hidden channels (C % 4 == 0); z, r, rh, q, dz, dq_pre, dh are dense
 * [rows][C], zr_pre / dzr_pre dense [rows][2C]:
 *   zr_fwd : z = sigmoid(zr_pre[:, :C]), r = sigmoid(zr_pre[:, C:]), rh = r*h       out_fwd: q = tanh(q_pre), h' = (1-z) h + z q
 *   out_bwd: dq_pre = dh' z (1-q^2), dz = dh' (q-h), dh = dh' (1-z)                 zr_bwd : dzr_pre = [dz z(1-z) | drh h r(1-r)], dh += drh r */
int craft_gru_zr_fwd(const float* zr_pre, long ldzr, const float* h, long ldh, float* z, float* r, float* rh, long rows, int C, void* stream);
int craft_gru_out_fwd(const float* q_pre, long ldq, const float* z, const float* h, long ldh, float* q, float* h_new, long ldhn, long rows,
                      int C, void* stream);
/* dq_pre_sum / dzr_pre_sum (or NULL): += the gate gradients just written -- their sum over the refinement iterations is what the hoisted
 *   context share of the gate convolutions differentiates once per pass.  dh_out / lddho (NULL: dh in place): where zr_bwd writes
 *   dh + drh r; it may be drh's own buffer (the first C columns of the q convolution's input gradient), which makes that buffer the
 *   per-pixel field the z|r input-gradient convolution accumulates into (craft_conv2d_nhwc2's bias_field, in place). */
int craft_gru_out_bwd(const float* dh_new, long lddhn, const float* z, const float* q, const float* h, long ldh, float* dq_pre, float* dz,
                      float* dh, long rows, int C, float* dq_pre_sum, void* stream);
int craft_gru_zr_bwd(const float* dz, const float* drh, long lddrh, const float* z, const float* r, const float* h, long ldh, float* dzr_pre,
                     float* dh, long rows, int C, float* dzr_pre_sum, float* dh_out, long lddho, void* stream);

/* ---- CNN encoders in training (BasicEncoder / ResidualBlock with autograd on: extractor.py:6-64, 124-196) ----
 * The inference path folds the encoders' normalisation layers away; training needs them as operators with a backward.
 * craft_norm_act_fwd: out = tail(act((x - mean) * rstd * gamma + beta)) on tokens [B][N][C] (C % 4 == 0);
 *   mean_rstd [B][C][2] (mr_per_image = 1: nn.InstanceNorm2d, statistics from craft_conv2d_nhwc_ex + craft_stats_finalize) or
 *   [C][2] (mr_per_image = 0: nn.BatchNorm2d, batch statistics in training / running statistics under freeze_bn);
 *   gamma / beta may be NULL (InstanceNorm2d is not affine); act = 0 / CRAFT ReLU (2); res != NULL: tail(t) = relu(res + t), the
 *   end of ResidualBlock.forward (extractor.py:56-64).
 * craft_norm_act_bwd_reduce: sums [B][C][2] (doubles, zeroed by the caller) += (sum dz, sum dz * x^) per image and channel, where dz
 *   is dy masked by the tail's ReLU (has_res: out > 0) and the inner ReLU, x^ = (x - mean) * rstd.  dbeta / dgamma are these sums
 *   (summed over the batch), and mean_P(dz), mean_P(dz * x^) over the normalisation population P feed the next call.
 * craft_norm_act_bwd_apply: dx = gamma * rstd * (dz - red[..][0] - x^ * red[..][1]) with red [B][C][2] / [C][2] (red_per_image) or
 *   NULL (running statistics: no mean terms); has_res: dres = dy masked by out > 0 (the gradient of the residual input).
 * craft_stem_im2col: the 7x7 / stride-2 stem's input patches as rows of 160 floats (147 = (ky*7+kx)*3 + c, then zeros) with the
 *   input normalisation 2*(x/255)-1 applied, one row per output pixel: the stem's weight gradient is craft_gemm(dY^T . cols).
 * craft_zero_stuff2: gf [B][Hin*Win][C] = g [B][(Hin/2)*(Win/2)][C] at the even positions, 0 elsewhere: the backward of a
 *   stride-2 convolution is the backward of the stride-1 convolution it subsamples, applied to gf. */
int craft_norm_act_fwd(const float* x, long ldx, const float* mean_rstd, int mr_per_image, const float* gamma, const float* beta, int act,
                       const float* res, long ldr, float* out, long ldo, int B, int N, int C, void* stream);
int craft_norm_act_bwd_reduce(const float* dy, long ldg, const float* out, long ldo, const float* x, long ldx, const float* mean_rstd,
                              int mr_per_image, const float* gamma, const float* beta, int act, int has_res, double* sums, int B, int N,
                              int C, void* stream);
int craft_norm_act_bwd_apply(const float* dy, long ldg, const float* out, long ldo, const float* x, long ldx, const float* mean_rstd,
                             int mr_per_image, const float* gamma, const float* beta, int act, int has_res, const float* red,
                             int red_per_image, float* dx, long lddx, float* dres, long lddr, int B, int N, int C, void* stream);
/* craft_bn_finalize: nn.BatchNorm2d statistics for craft_norm_act_fwd.  stats != NULL (training): the producing conv's
 *   [CRAFT_STATS_REPLICAS][B][C][2] sums over `count` pixels per image -> batch mean / biased variance -> mean_rstd [C][2]; running_mean
 *   / running_var (may be NULL) get the momentum update with the unbiased variance.  stats == NULL: mean_rstd from the running
 *   statistics (eval mode, CRAFT.freeze_bn()).
 * craft_norm_bwd_finalize: sums [B][C][2] of craft_norm_act_bwd_reduce -> red (population means for craft_norm_act_bwd_apply: [B][C][2]
 *   when per_image, else [C][2]; untouched when population == 0) and dgamma / dbeta [C] (sums over the batch; may be NULL). */
int craft_bn_finalize(const double* stats, int B, int C, double count, float eps, float momentum, float* mean_rstd, float* running_mean,
                      float* running_var, void* stream);
int craft_norm_bwd_finalize(const double* sums, int B, int C, double population, int per_image, float* red, float* dgamma, float* dbeta,
                            void* stream);
int craft_stem_im2col(const float* image, int B, int H, int W, float* cols, void* stream);
int craft_zero_stuff2(const float* g, long ldg, int B, int Hin, int Win, int C, float* gf, long ldf, void* stream);

/* ==== input pipeline on the GPU (core/utils/augmentor.py) ========================================
 * All images are float HWC in 0..255 ([H][W][3]), flow [H][W][2] (x, y); the host draws the random parameters.
 * craft_aug_spatial: FlowAugmentor.spatial_transform (augmentor.py:141-193) as one gather: out [ch][cw][C] = crop at (y0, x0) of
 *   v-flip(h-flip(resize(src))) where resize is cv2.INTER_LINEAR's mapping src = (dst + 0.5) / f - 0.5 with replicated borders on
 *   a (round(H*fy), round(W*fx)) grid (do_resize = 0: no resize).  is_flow: C = 2, values scaled by (fx, fy) and negated by the
 *   flips; else resized values are rounded to integer levels in 0..255.
 * craft_aug_photo: one ColorJitter step on img [npix][3] (integer levels) in place -- torchvision.transforms.ColorJitter on a PIL image
 *   (augmentor.py:104, :111-123) = Pillow's 8-bit arithmetic, bit for bit (tests/golden/photo_pil.npz, produced by Pillow): op 0 brightness
 *   / 1 contrast / 2 saturation = ImagingBlend(degenerate, image, factor) with degenerate = black / the grey level `mean` (= int(mean of
 *   the "L" image + 0.5), which the caller reduces) / the pixel's "L"; op 3 hue = RGB -> 8-bit HSV, hue + `mean` mod 256 (`mean` carries
 *   the integer shift int32(hue_factor * 255) & 255 torchvision adds to the hue plane; `factor` unused), HSV -> RGB.
 * craft_aug_erase: FlowAugmentor.eraser_transform (augmentor.py:125-139): rects [nrect][4] = (x0, y0, dx, dy) (device ints) filled
 *   with (mr, mg, mb).
 * craft_aug_shift: random_shift (augmentor.py:16-78) for even (dx, dy): the two frames cropped against each other by the shift,
 *   flow - (dx, dy), zero-padded back to [H][W], valid [H][W] = 1 inside the remaining area.
 * craft_aug_blur: cv2.GaussianBlur(img, (K, K), sigma) of FlowAugmentor.__call__ (augmentor.py:195-198; blur_sigma > 0): out [H][W][C] =
 *   src filtered with the separable Gaussian exp(-(i - (K-1)/2)^2 / (2 sigma^2)) / sum (cv2.getGaussianKernel for sigma > 0), border
 *   BORDER_REFLECT_101, rounded to integer levels.  K odd, 1 <= K <= 31; out must not alias src. */
int craft_aug_spatial(const float* src, int H, int W, int C, int do_resize, float fx, float fy, int hflip, int vflip, int y0, int x0, int ch,
                      int cw, int is_flow, float* out, void* stream);
int craft_aug_photo(float* img, long npix, int op, float factor, float mean, void* stream);
/* SparseFlowAugmentor (KITTI; augmentor.py:249-316): resize_sparse_flow_map -- every valid source pixel moves to (round(x fx), round(y
 * fy)) when strictly inside the resized frame, the last source pixel in row-major order winning a contested target -- followed by
 * the h-flip and the crop at (y0, x0): out_flow [ch][cw][2] (scaled by (fx, fy), u negated by the flip), out_valid [ch][cw].
 * owner: scratch of round(H fy) * round(W fx) ints.  fx = fy = 1 is the no-resize case. */
int craft_aug_sparse(const float* flow, const float* valid, int H, int W, float fx, float fy, int hflip, int y0, int x0, int ch, int cw,
                     int* owner, float* out_flow, float* out_valid, void* stream);
int craft_aug_erase(float* img, int H, int W, const int* rects, int nrect, float mr, float mg, float mb, void* stream);
int craft_aug_shift(const float* img1, const float* img2, const float* flow, int H, int W, int dx, int dy, float* out1, float* out2,
                    float* out_flow, float* valid, void* stream);
int craft_aug_blur(const float* src, int H, int W, int C, int K, float sigma, float* out, void* stream);

/* Host helper of the evaluation harness (frame_utils.py:70-120 reads KITTI's 16-bit PNGs through cv2): PNG scan-line unfiltering,
 * filter types 0-4; rows [h][1 + stride] -> out [h][stride], bpp = bytes per pixel.  HOST pointers, no stream, no device work. */
int craft_png_unfilter(const unsigned char* rows, int h, int stride, int bpp, unsigned char* out);

/* CRAFT.upsample_flow (network.py:151-162): mask tokens [B*N][576], flow tokens [B*N][2] -> up NCHW
 * [B][2][8*H8][8*W8]. */
int craft_convex_upsample(const float* mask, const float* flow, int B, int H8, int W8, float* up, void* stream);

/* coords_grid + flow_init (utils.py:82-85, network.py:219-222): coords0 = grid, coords1 = grid + flow_init
 * (NCHW [B][2][H8][W8], may be NULL), flow = coords1 - coords0; all tokens [B*N][2]. */
int craft_coords_init(const float* flow_init_nchw, int B, int H8, int W8, float* coords0, float* coords1,
                      float* flow, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CRAFT_HIP_H */
