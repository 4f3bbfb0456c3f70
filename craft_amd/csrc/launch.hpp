// Host-visible parameter blocks and launcher prototypes shared by the .hip translation units and the
// C-ABI layer (craft_hip.hip).  Nothing here is part of the public ABI (see include/craft_hip.h).
#pragma once
#include "common.hpp"
#include "gemm_engine.hpp"

#define CRAFT_OK 0
#define CRAFT_ERR_ARG 10001
#define CRAFT_ERR_ALIGN 10002
#define CRAFT_ERR_UNSUPPORTED 10003

namespace craft {

// Developer A/B overrides from the environment, read ONCE when the library is loaded (never in a launch path):
// CRAFT_HALO_BN (64 | 128), CRAFT_NO_C64, CRAFT_WF_DYNAMIC_TAPS.  Everything a caller may legitimately vary per call is an
// argument of the C ABI instead (e.g. CRAFT_PV_ROWS in craft_attn_apply's prec).
struct Tuning { int halo_bn; bool no_c64, wf_dynamic_taps, no_wgrad64, wgrad_sb; int pk_mode, corr_dbg; bool pv_wr2; int corr_ncp; int conv_xcd; };
const Tuning& tuning();

struct RowsGemmParams {
  const void* A; const void* B; void* C;
  int c_dtype;           // element type of C: 0 fp32, 1 bf16, 2 fp16
  long lda, ldb, ldc;
  long a_bs0, a_bs1, b_bs0, b_bs1, c_bs0, c_bs1;   // batch z -> (z / zdiv, z % zdiv) strides, in elements
  int zdiv, batch;
  int M, N, K;
  const float* bias;     // per output column, may be null
  float scale;
  int act;
  const float* row_div; long rd_bs;   // non-null: C row r of batch z is divided by row_div[z * rd_bs + r] (deferred softmax sums)
  int c_frag;            // > 0: C (16-bit) is stored in MFMA B-fragment order per group of c_frag rows (k_pv16's V^T operand)
  int c_frag_acc;        // with c_frag: the 16 rows (keys) of a k-group are enumerated in MFMA ACCUMULATOR order
                         // (position 8*h + j <-> key 8*(j >> 2) + 4*h + (j & 3)): k_flash_attn's V^T operand
  int b_packed;          // B (the weight operand) is craft_pack_weights' fragment order for `prec`, K padded to 32: k_gemm_rows_wf
  int a_tiled;           // k_pv16: A (= P) in 32-row x 64-key tiles (CRAFT_P_TILED): element (i, j) at
                         // ((i >> 5) * 32 * lda) + (j >> 6) * 2048 + (i & 31) * 64 + (j & 63); K = lda is a multiple of 64, ldb = V^T's key extent
};

enum { CONV_EPI_BIAS_ACT = 0, CONV_EPI_GRU_ZR = 1, CONV_EPI_GRU_Q = 2, CONV_EPI_MENC = 3 };

struct ConvGemmParams {
  ConvGeom g;
  const float* W;        // packed [cout][KH][KW][c0+c1]
  const float* bias;     // [cout]
  int cout;
  int epi, act;
  float scale;
  float* out; int ldo;
  const float* aux0; int ld0;
  float* aux1; int ld1;
  int force_generic;     // 1: always use the generic implicit GEMM (k_gemm_conv), for A/B tests
  int w_packed;          // 1: W was produced by craft_pack_weights for this precision (halo kernel only)
  int w16;               // 1 (f16x3, packed weights, static taps): use the weights' hi plane only -- CRAFT_CONV_W16, two MFMAs per product
  double* stats;         // optional [B][cout][2]: += (sum, sum^2) of the biased conv output per (image, channel)
  int bf_col0;                          // with bias_field: output columns < bf_col0 get no bias at all (CRAFT_CONV_FIELD_COL0)
  const float* bias_field; int ld_bf;   // optional per-pixel bias [npix][ld_bf] used INSTEAD of bias[col] (hoisted
                                        // iteration-invariant part of a conv: SepConvGRU context term)
  const float* res; int ld_res;         // optional (CONV_EPI_BIAS_ACT): out = relu(res[pix][col] + act(conv + bias) * scale) -- the tail of a
                                        // ResidualBlock (extractor.py:56-63) in the epilogue of its second convolution
  const float* mask; int ld_mask;       // optional (CONV_EPI_BIAS_ACT): out = mask[pix][col] > 0 ? out : 0 -- the ReLU backward of the layer
                                        // BELOW an input-gradient convolution (mask = that layer's saved output), fused into the epilogue
};

int launch_pack_operand(const float* x, long ldx, int C, long rows, int B, int H, int W, int padH, int padW, long guard, long rows_p,
                        int prec, void* out, int cg_off, int ncg_total, float* colsum, int tail, hipStream_t s);
int launch_pack_operands(const long* descs, int n, hipStream_t s);
int launch_gemm_pk(const void* A, const long* a_desc, const void* B, const long* b_desc, float* C, long ldc, long c_outer, long c_inner,
                   int inner, int nbatch, int M, int N, int K, float alpha, int prec, hipStream_t s);
int launch_wgrad_pk(const void* const* dYp, const void* const* Xp, const void* const* Xp1, int cin0, int nseg, long dy_rows_p, int cout, long x_rows_p,
                    int cin, long guard, long K, int KH, int KW, int Wp, float* dW, int prec, int prec_x, hipStream_t s);
int launch_pack_conv_weights(const float* w0, int cout0, const float* w1, int cout1, int Cin, int KH, int KW, int a0, int a1, int b0, int b1,
                             int transposed, int prec, void* out, hipStream_t s);
size_t pack_conv_job_bytes();
long fill_pack_conv_job(void* job, const float* w0, int cout0, const float* w1, int cout1, int Cin, int KH, int KW, int a0, int a1, int b0, int b1,
                        int transposed, int prec, void* out);
int launch_pack_conv_weights_batch(const void* jobs_dev, const int* first_dev, int n, int total_blocks, hipStream_t s);
int launch_relpos_add(float* S, long ld, int BZ, int H8, int W8, const float* Hs, long ldh, const float* Ws, long ldw, float w, hipStream_t s);
int launch_relpos_bwd(const float* dS, long ld, int BZ, int H8, int W8, float* dHs, long ldh, int nh, float* dWs, long ldw, int nw, float w,
                      hipStream_t s);
int launch_gemm_rows(const RowsGemmParams& p, int prec, bool a16, hipStream_t s);
int launch_pv16(const RowsGemmParams& p, int prec, int rows32, hipStream_t s);   // A and B both 16-bit (type = prec), C fp32;
                                                                                  // rows32: 32-row groups per block (4..7), 0 = auto
int launch_gemm_conv(const ConvGemmParams& p, int prec, hipStream_t s);
int launch_conv_halo(const ConvGemmParams& p, int prec, hipStream_t s);
int launch_conv_halo_wf(const ConvGemmParams& p, int prec, hipStream_t s);
bool conv3x3_c64_applies(const ConvGemmParams& p);
int launch_conv3x3_c64(const ConvGemmParams& p, int prec, hipStream_t s);
int launch_pack_weights(const float* w, int rows, int K, int prec, void* out, hipStream_t s);
int launch_stem7x7(const float* img, const float* w, const float* bias, int act, int B, int H, int W, float* out, double* stats,
                   hipStream_t s);
int launch_stats_finalize(const double* sums, long n, double count, float eps, float* mean_rstd, hipStream_t s);
int launch_residual_relu(const float* x, long ldx, const float* xnorm, const float* y, long ldy, const float* ynorm, int y_relu,
                         int B, int HW, int C, float* out, long ldo, hipStream_t s);

// ---- attention / correlation (kernels_attn.hip) ----
struct ScoreParams {
  const float* Q; const float* Kf;    // [B][N][ld] token-major features (already projected)
  long ldq, ldk, q_bs, k_bs;          // row strides and per-sample strides (elements)
  int B, H8, W8, N;
  int M, d;                           // modes, per-mode width (columns m*d .. m*d+d)
  float scale;                        // 1/sqrt(d)
  const float* pos_tab; int R; float pos_w;   // sliding bias table [(2R+1)^2] (null: no bias)
  int mask_radius;                    // Chebyshev mask radius (<=0: none)
  const unsigned* clamp_ord;          // ordered-uint global max of the raw scores (null: never clamp)
  // per-QUERY relative-position scores (gma.RelPosEmb, gma.py:21-50): logit += rb_w * (rb_h[q][kh - qh + H8 - 1] +
  // rb_w_[q][kw - qw + W8 - 1]), q = (b*M + m)*N + query; null: none
  const float* rb_h; const float* rb_wd; long ld_rbh, ld_rbw; float rb_w;
  float* rowsum;                      // k_attn_probs: non-null = deferred normalisation, row sums [B][M][N] out ...
  unsigned* rowmax;                   // ... followed by [B][M][N] ordered-uint row maxima (scratch): rowsum + B*M*N
  int tiled;                          // k_corr_build4t: levels 0 / 1 of the pyramid in the tiled layout (CRAFT_PYR_TILED)
  int dbg;                            // developer ablation of k_corr_build4t's stores (CRAFT_CORR_DBG: 1 no level 0, 2 no levels 1-3, 4 no mode-0 softmax), 0 in production
  int ncp;                            // k_corr_build4t: cell pairs of a cell row one block walks (the row band; <= 1: one pair per block as in rounds 2-5)
};

// deferred-normalisation probabilities as one launch of independent waves (kernels_attn_w.hip); ws: B*M*ceil(N/128)*16384 bytes
int launch_attn_probs_fused(const ScoreParams& p, void* P, long ldp, void* ws, int p_prec, int prec, int tiled, hipStream_t s);

// ---- flash-fused attention (kernels_flash.hip) ----
struct FlashParams {
  const uint16_t* Qf; const uint16_t* Kf;   // pre-split fragment order [z][32-row block][k-step][plane][lane][8]
  const uint16_t* Vf; long v_bs, v_ms;      // V^T fragments (accumulator key order): per-sample / per-mode strides (elements)
  float* O;                                 // [B][M][N][Dv]
  int B, M, N, H8, W8;
  int nkt, nqb, nkb;                        // key tiles; 32-row blocks allocated per z for Q (multiple of 8) and K
  unsigned w8_magic;                        // ceil(2^32 / W8): key / W8 = umulhi(key, magic) for key < 2^16
  const float* pos_tab; int R; float pos_w; int mask_radius; const unsigned* clamp_ord;
};
size_t flash_ws_bytes(int B, int M, int N, int d, int score_prec);
int launch_flash_attn(const ScoreParams& sp, const void* vT, long ldt, int Dv, float* O, void* ws, int score_prec, int pv_prec,
                      hipStream_t s);

int launch_score_max(const ScoreParams& p, unsigned* max_ord, int prec, hipStream_t s);
int launch_corr_build(const ScoreParams& p, float w_aggr, float* pyr0, double* sums, void* ws, int prec, hipStream_t s);
int launch_corr_build_pyramid(const ScoreParams& p, float w_aggr, float* pyr0, float* pyr1, float* pyr2, float* pyr3, double* sums,
                              void* ws, int prec, int tiled, hipStream_t s);
int launch_attn_probs(const ScoreParams& p, void* P, long ldp, int p_prec, int prec, hipStream_t s);
template <int D> int launch_attn_probs_d(const ScoreParams& p, void* P, long ldp, int p_prec, int prec, hipStream_t s);

// ---- element-wise / gather kernels (kernels_misc.hip) ----
int launch_tokens(const float* src, int src_nchw, int B, int Ctot, int c_off, int C, int HW, long src_ld,
                  int act, int do_ln, float* dst, long dst_ld, hipStream_t s);
int launch_mode_pool_ln(const float* O, const float* x, long ldx, const float* w_agg, const float* skip_coeff,
                        int B, int N, int M, int C, float* out, long ldo, hipStream_t s);
int launch_corr_pyramid(const float* l0, float* l1, float* l2, float* l3, long nimg, int H8, int W8, hipStream_t s);
int launch_corr_stats(const double* sums, float* mu_rstd, int B, double count, int do_norm, hipStream_t s);
int launch_corr_lookup(const float* l0, const float* l1, const float* l2, const float* l3, int levels,
                       const float* mu_rstd, const float* coords, int B, int H8, int W8, int radius,
                       float* out, long ldo, int lvl_stride, int col_off,
                       int tiled, hipStream_t s);
int launch_convf1(const float* flow, const float* w, const float* bias, int B, int H8, int W8, float* out, long ldo,
                  hipStream_t s);
int launch_convf1_mfma(const float* flow, const void* w_packed, const float* bias, int B, int H8, int W8, float* out, long ldo, int prec,
                       hipStream_t s);
int launch_flow_head2(const float* hid, const float* w, const float* bias, int B, int H8, int W8, float* coords1,
                      const float* coords0, float* flow, float* delta, hipStream_t s);
int launch_convex_upsample(const float* mask, const float* flow, int B, int H8, int W8, float* up, hipStream_t s);
int launch_stem_mfma(const float* img, const float* img2, int bsplit, const void* w_packed, const float* bias, int act, int B, int H, int W, float* out,
                     double* stats, int prec, hipStream_t s);
int launch_flow_metrics(const float* pred, const float* gt, const float* valid, int B, int H, int W, float offx, float offy,
                        float max_mag, double* out, hipStream_t s);
int launch_flow_l1(const float* pred, const float* gt, const float* valid, int B, int H, int W, float weight, float max_flow,
                   double* loss, float* grad, hipStream_t s);
int launch_sumsq(const float* x, long n, double* out, hipStream_t s);
int launch_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, float wd,
                 int step, float grad_mul, const double* sumsq, float max_norm, hipStream_t s);
int launch_scaler_update(const double* sumsq, void* state, float grad_mul, float max_norm, float beta1, float beta2, float growth,
                         float backoff, int growth_interval, hipStream_t s);
int launch_adamw_dyn(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, float wd,
                     const void* state, hipStream_t s);
int launch_forward_interpolate(const float* flow, int B, int H, int W, float* out, hipStream_t s);
int launch_coords_init(const float* flow_init_nchw, int B, int H8, int W8, float* coords0, float* coords1, float* flow,
                       hipStream_t s);
int launch_tokens_to_nchw(const float* src, long ld, int B, int C, int HW, float* dst, hipStream_t s);
int launch_gma_residual(const float* mf, long ldm, const float* O, const float* gamma, int B, int N, int C, float* out,
                        long ldo, hipStream_t s);

// ---- training: general GEMM / weight gradient (kernels_gemm_gen.hip), element-wise and row-wise backward (kernels_train.hip) ----
int launch_gemm_gen(const float* A, long a_sm, long a_sk, long a_bs0, long a_bs1, const float* B, long b_sn, long b_sk, long b_bs0,
                    long b_bs1, float* C, long ldc, long c_bs0, long c_bs1, int zdiv, int batch, int M, int N, int K, float alpha,
                    int accumulate, int ksplit, int prec, hipStream_t s);
int launch_conv_wgrad(const float* x, long ldx, int cin, const float* dy, long ldy, int cout, int KH, int KW, int B, int H, int W,
                      float* dW, float* db, float* ws, long ws_floats, int prec, hipStream_t s);
// kernels_enc_train.hip
struct NormActParams {
  const float* x; long ldx;
  const float* mr; int mr_bs;             // (mean, rstd) at mr[(b * mr_bs + c) * 2]; mr_bs = C (per image) or 0 (per channel)
  const float* gamma; const float* beta;  // may be null (1, 0)
  int act;                                // CRAFT_ACT_NONE / CRAFT_ACT_RELU on the normalised value
  const float* res; long ldr;             // forward: residual input (null: none)
  float* out; long ldo;                   // forward output; backward: the forward's output (read only when has_res)
  const float* dy; long ldg;
  int has_res;
  const float* red; int red_bs;           // backward apply: (mean dz, mean dz * x^) at red[(b * red_bs + c) * 2]; null = (0, 0)
  float* dx; long lddx;
  float* dres; long lddr;
  double* sums;                           // backward reduce: [B][C][2] += (sum dz, sum dz * x^)
  int B, N, C;
};
int launch_norm_act_fwd(const NormActParams& p, hipStream_t s);
int launch_norm_act_bwd_reduce(const NormActParams& p, hipStream_t s);
int launch_norm_act_bwd_apply(const NormActParams& p, hipStream_t s);
int launch_stem_im2col(const float* img, int B, int H, int W, float* cols, hipStream_t s);
int launch_zero_stuff2(const float* g, long ldg, int B, int Hin, int Win, int C, float* gf, long ldf, hipStream_t s);
int launch_bn_finalize(const double* stats, int B, int C, double count, float eps, float momentum, float* mr, float* rmean, float* rvar,
                       hipStream_t s);
int launch_norm_bwd_finalize(const double* sums, int B, int C, double population, int per_image, float* red, float* dgamma, float* dbeta,
                             hipStream_t s);
int launch_colsum(const float* x, long ld, long rows, int C, float* out, hipStream_t s);
int launch_multi_copy(const void* const* src, const long* n, const long* dst_off, const long* chlast, int count, float* dst, hipStream_t s);
int launch_act_fwd(const float* x, long ldx, float* y, long ldy, long rows, int C, int act, float scale, hipStream_t s);
int launch_act_bwd(const float* dy, long lddy, const float* dy2, long lddy2, const float* y, long ldy, float* dx, long lddx, long rows, int C,
                   int act, float scale, int ztail, hipStream_t s);
int launch_dropout(const float* x, float* y, long n, float p, unsigned long long seed, hipStream_t s);
int launch_tokens_bwd(const float* x, long ldx, const float* dy, long lddy, float* dx, long lddx, long rows, int C, int act, int do_ln,
                      hipStream_t s);
int launch_attn_softmax_fwd(float* S, long ld, int B, int M, int H8, int W8, const float* pos_tab, int R, float pos_w, int mask_radius,
                            const unsigned* clamp_ord, unsigned* clampbits, float* Pdrop, float drop_p, unsigned long long seed,
                            void* Ppk, long pk_rows, int pk_np, int pk_prec, hipStream_t s);
int launch_attn_softmax_bwd(const float* P, float* dP, long ld, int B, int M, int H8, int W8, int R, float pos_w,
                            const unsigned* clamp_ord, const unsigned* clampbits, float* dtab, float drop_p, unsigned long long seed,
                            void* dSpk, long pk_rows, int pk_np, int pk_prec, hipStream_t s);
int launch_reduce_replicas(const float* rep, int nrep, int n, float* out, hipStream_t s);
int launch_corr_pool_fwd(const float* S, long ld, int B, int M, int H8, int W8, const float* pos_tab, int R, float pos_w, const float* w,
                         const unsigned* clamp_ord, float* c0, double* sums, hipStream_t s);
int launch_corr_pool_bwd(float* S, long ld, int B, int M, int H8, int W8, const float* pos_tab, int R, float pos_w, const float* w,
                         const unsigned* clamp_ord, const float* c0, const float* G, const float* mu_rstd, const double* gstats,
                         int do_norm, float* dtab, double* dw, hipStream_t s);
int launch_corr_pyramid_bwd(float* G0, const float* G1, const float* G2, const float* G3, const float* c0, const float* mu_rstd, int B,
                            int H8, int W8, double* gstats, hipStream_t s);
int launch_corr_lookup_bwd(const float* dout, long ldo, const float* coords, float* G0, float* G1, float* G2, float* G3, int levels, int B,
                           int H8, int W8, int radius, int lvl_stride, int col_off, hipStream_t s);
int launch_mode_pool_ln_bwd(const float* O, const float* x, long ldx, const float* w_agg, const float* skip_coeff, const float* dy, long lddy,
                            int B, int N, int M, int C, float* dO, float* dx, long lddx, float* dw_rep, hipStream_t s);
int launch_flow_tokens(const float* c1, const float* c0, long rows, float* flow, float* flow32, float* c1copy, hipStream_t s);
int launch_convex_upsample_bwd(const float* mask, long ldm, const float* flow, const float* dup, int B, int H8, int W8, float* dmask,
                               long lddm, float* dflow, long lddf, hipStream_t s);
int launch_gru_zr_fwd(const float* zr, long ldzr, const float* h, long ldh, float* z, float* r, float* rh, long rows, int C, hipStream_t s);
int launch_gru_out_fwd(const float* qp, long ldq, const float* z, const float* h, long ldh, float* q, float* hn, long ldhn, long rows, int C,
                       hipStream_t s);
int launch_gru_out_bwd(const float* dhn, long lddhn, const float* z, const float* q, const float* h, long ldh, float* dqp, float* dz, float* dh,
                       long rows, int C, float* dqp_sum, hipStream_t s);
int launch_gru_zr_bwd(const float* dz, const float* drh, long lddrh, const float* z, const float* r, const float* h, long ldh, float* dzr,
                      float* dh, long rows, int C, float* dzr_sum, float* dh_out, long lddho, hipStream_t s);

// ---- input pipeline (kernels_augment.hip) ----
int launch_aug_spatial(const float* src, int H, int W, int C, int do_resize, float fx, float fy, int hflip, int vflip, int y0, int x0, int ch,
                       int cw, int is_flow, float* out, hipStream_t s);
int launch_aug_sparse(const float* flow, const float* valid, int H, int W, float fx, float fy, int hflip, int y0, int x0, int ch, int cw,
                      int* owner, float* oflow, float* ovalid, hipStream_t s);
int launch_aug_photo(float* img, long npix, int op, float factor, float mean, hipStream_t s);
int launch_aug_erase(float* img, int H, int W, const int* rects, int nrect, float mr, float mg, float mb, hipStream_t s);
int launch_aug_shift(const float* img1, const float* img2, const float* flow, int H, int W, int dx, int dy, float* o1, float* o2, float* oflow,
                     float* valid, hipStream_t s);

}  // namespace craft
