// C ABI of libcraft_hip.so (declared in include/craft_hip.h): thin argument marshalling over the kernel
// launchers; operator-level entry points (motion encoder, GRU, heads) compose several launches.
#include <cstdlib>
#include "launch.hpp"
#include "../../include/craft_hip.h"

using namespace craft;

#define S(stream) reinterpret_cast<hipStream_t>(stream)
#define TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

namespace craft {
const Tuning& tuning() {
  static const Tuning t = [] {
    Tuning v = {};
    if (const char* e = getenv("CRAFT_HALO_BN")) v.halo_bn = atoi(e);
    v.no_c64 = getenv("CRAFT_NO_C64") != nullptr;
    v.wf_dynamic_taps = getenv("CRAFT_WF_DYNAMIC_TAPS") != nullptr;
    v.wgrad_sb = getenv("CRAFT_NO_WGRAD_SB") == nullptr;     // weight gradient with one LDS tile buffer (3 resident blocks per CU instead of 2; off: developer A/B)
    v.no_wgrad64 = getenv("CRAFT_NO_WGRAD64") != nullptr;   // weight gradient of 64-channel layers on the generic 128-row tile (developer A/B)
    if (const char* e = getenv("CRAFT_CORR_DBG")) v.corr_dbg = atoi(e);    // store ablations of k_corr_build4t (developer, tools/corr_write_pmc.sh)
    if (const char* e = getenv("CRAFT_PK_MODE")) v.pk_mode = atoi(e);      // ablations of k_gemm_pk (developer): 1 no DMA after tile 0, 2 no epilogue, 4 no MFMA phase
    v.corr_ncp = 8;                                       // k_corr_build4t: cell pairs per block (the row band); CRAFT_CORR_NCP=1: one pair per block (developer A/B)
    if (const char* e = getenv("CRAFT_CORR_NCP")) v.corr_ncp = atoi(e);
    v.pv_wr2 = getenv("CRAFT_PV_NO_WR2") == nullptr;      // k_pv16: the launcher may pick the 8-wave (2 x 32 MT rows) instantiation (round 6; off: developer A/B)
    if (const char* e = getenv("CRAFT_CONV_XCD")) v.conv_xcd = atoi(e);     // k_conv_halo_wf: XCD x owns a contiguous eighth of the patch list (developer A/B, round 6)
    return v;
  }();
  return t;
}
}  // namespace craft
static const craft::Tuning& g_tuning_at_load = craft::tuning();     // evaluated by the dynamic loader, not by a launch

// (declared here rather than in launch.hpp, the header of every kernel translation unit)
namespace craft { int launch_aug_blur(const float* src, int H, int W, int C, int K, float sigma, float* out, hipStream_t s); }

extern "C" {

int craft_hip_abi_version(void) { return CRAFT_HIP_ABI_VERSION; }

const char* craft_hip_error_string(int code) {
  switch (code) {
    case 0: return "ok";
    case CRAFT_ERR_ARG: return "craft: invalid argument";
    case CRAFT_ERR_ALIGN: return "craft: alignment / divisibility requirement violated (strides %4, segment channels %32, d %32)";
    case CRAFT_ERR_UNSUPPORTED: return "craft: size outside supported range";
    default: return hipGetErrorString((hipError_t)code);
  }
}

int craft_tokens(const float* src, int src_nchw, int B, int Ctot, int c_off, int C, int HW, long src_ld, int act,
                 int do_ln, float* dst, long dst_ld, void* stream) {
  return launch_tokens(src, src_nchw, B, Ctot, c_off, C, HW, src_ld, act, do_ln, dst, dst_ld, S(stream));
}

int craft_tokens_to_nchw(const float* src, long ld, int B, int C, int HW, float* dst, void* stream) {
  return launch_tokens_to_nchw(src, ld, B, C, HW, dst, S(stream));
}

int craft_linear(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy, long rows, int cin,
                 int cout, int prec, void* stream) {
  RowsGemmParams p = {};
  p.A = x; p.lda = ldx; p.B = w; p.ldb = cin; p.C = y; p.ldc = ldy;
  p.zdiv = 1; p.batch = 1; p.M = (int)rows; p.N = cout; p.K = cin;
  p.bias = bias; p.scale = 1.f; p.act = CRAFT_ACT_NONE;
  p.b_packed = ((prec >> 8) & 1) && (prec & 0xff) != CRAFT_PREC_F32;          // prec | CRAFT_W_PACKED: w from craft_pack_weights(rows cout, K round_up(cin, 32))
  return launch_gemm_rows(p, prec & 0xff, false, S(stream));
}

int craft_linear_t(const float* x, long ldx, const float* w, void* yT, long ldt, int B, int N, int cin, int cout,
                   int out_prec, int frag_rows, int prec, void* stream) {
  if (out_prec < 0 || out_prec > 2) return CRAFT_ERR_ARG;
  const int frag_acc = (frag_rows & CRAFT_FRAG_ACC_ORDER) ? 1 : 0;
  frag_rows &= ~CRAFT_FRAG_ACC_ORDER;
  if (frag_acc && !frag_rows) return CRAFT_ERR_ARG;
  if (frag_rows && (out_prec == CRAFT_PREC_F32 || frag_rows % 32 || cout % frag_rows || ldt % 16)) return CRAFT_ERR_ALIGN;
  RowsGemmParams p = {};
  p.c_frag_acc = frag_acc;
  p.c_dtype = out_prec;
  p.zdiv = 1; p.batch = B; p.K = cin;
  p.bias = nullptr; p.scale = 1.f; p.act = CRAFT_ACT_NONE;
  if (frag_rows) {
    // fragment order: computed as y = x W^T (rows = keys) so that 4 consecutive keys of one V^T row -- 8 contiguous
    // bytes of the fragment layout -- sit in one lane of the accumulator
    p.c_frag = frag_rows;
    p.A = x; p.lda = ldx; p.a_bs0 = (long)N * ldx; p.B = w; p.ldb = cin; p.C = yT; p.ldc = ldt; p.c_bs0 = (long)cout * ldt;
    p.M = N; p.N = cout;
  } else {
    p.A = w; p.lda = cin; p.B = x; p.ldb = ldx; p.b_bs0 = (long)N * ldx; p.C = yT; p.ldc = ldt; p.c_bs0 = (long)cout * ldt;
    p.M = cout; p.N = N;
  }
  p.b_packed = ((prec >> 8) & 1) && (prec & 0xff) != CRAFT_PREC_F32;
  if (p.b_packed && !frag_rows) return CRAFT_ERR_UNSUPPORTED;                 // (the un-fragmented form has the weights as the A operand)
  return launch_gemm_rows(p, prec & 0xff, false, S(stream));
}

static ScoreParams make_score(const float* q, long ldq, const float* k, long ldk, int B, int H8, int W8, int M, int d,
                              float scale, const float* pos_tab, int R, float pos_w, int mask_radius,
                              const unsigned* clamp_ord) {
  ScoreParams p = {};
  p.Q = q; p.Kf = k; p.ldq = ldq; p.ldk = ldk;
  p.B = B; p.H8 = H8; p.W8 = W8; p.N = H8 * W8;
  p.q_bs = (long)p.N * ldq; p.k_bs = (long)p.N * ldk;
  p.M = M; p.d = d; p.scale = scale; p.pos_tab = pos_tab; p.R = R; p.pos_w = pos_w; p.mask_radius = mask_radius;
  p.clamp_ord = clamp_ord;
  return p;
}

int craft_score_max(const float* q, long ldq, const float* k, long ldk, int B, int H8, int W8, int M, int d, float scale,
                    unsigned* max_ord, int prec, void* stream) {
  return launch_score_max(make_score(q, ldq, k, ldk, B, H8, W8, M, d, scale, nullptr, 0, 0.f, -1, nullptr), max_ord, prec,
                          S(stream));
}

int craft_corr_build(const float* q, long ldq, const float* k, long ldk, int B, int H8, int W8, int M, int d, float scale,
                     const float* pos_tab, int R, float pos_w, float w_aggr, const unsigned* clamp_ord, float* pyr0,
                     double* sums, void* ws, int prec, void* stream) {
  return launch_corr_build(make_score(q, ldq, k, ldk, B, H8, W8, M, d, scale, pos_tab, R, pos_w, -1, clamp_ord), w_aggr,
                           pyr0, sums, ws, prec, S(stream));
}

int craft_corr_build_pyramid(const float* q, long ldq, const float* k, long ldk, int B, int H8, int W8, int M, int d, float scale,
                             const float* pos_tab, int R, float pos_w, float w_aggr, const unsigned* clamp_ord, float* pyr0,
                             float* pyr1, float* pyr2, float* pyr3, double* sums, void* ws, int prec, void* stream) {
  return launch_corr_build_pyramid(make_score(q, ldq, k, ldk, B, H8, W8, M, d, scale, pos_tab, R, pos_w, -1, clamp_ord), w_aggr,
                                   pyr0, pyr1, pyr2, pyr3, sums, ws, prec & ~CRAFT_PYR_TILED, (prec & CRAFT_PYR_TILED) ? 1 : 0, S(stream));
}

int craft_corr_finish(const float* pyr0, float* pyr1, float* pyr2, float* pyr3, const double* sums, float* mu_rstd, int B,
                      int H8, int W8, int do_norm, void* stream) {
  const long N = (long)H8 * W8;
  if (pyr1) TRY(launch_corr_pyramid(pyr0, pyr1, pyr2, pyr3, (long)B * N, H8, W8, S(stream)));
  return launch_corr_stats(sums, mu_rstd, B, (double)N * (double)N, do_norm, S(stream));
}

int craft_corr_lookup(const float* pyr0, const float* pyr1, const float* pyr2, const float* pyr3, int levels,
                      const float* mu_rstd, const float* coords, int B, int H8, int W8, int radius, float* out, long ldo,
                      int lvl_stride, int col_off, void* stream) {
  return launch_corr_lookup(pyr0, pyr1, pyr2, pyr3, levels & ~CRAFT_PYR_TILED, mu_rstd, coords, B, H8, W8, radius, out, ldo, lvl_stride,
                            col_off, (levels & CRAFT_PYR_TILED) ? 1 : 0, S(stream));
}

int craft_attn_probs(const float* q, long ldq, const float* k, long ldk, int B, int H8, int W8, int M, int d, float scale,
                     const float* pos_tab, int R, float pos_w, int mask_radius, const unsigned* clamp_ord,
                     const float* relpos_h, const float* relpos_w, float relpos_weight, void* P,
                     long ldp, float* rowsum, int p_prec, int prec, void* stream) {
  ScoreParams sp = make_score(q, ldq, k, ldk, B, H8, W8, M, d, scale, pos_tab, R, pos_w, mask_radius, clamp_ord);
  if ((relpos_h == nullptr) != (relpos_w == nullptr)) return CRAFT_ERR_ARG;
  sp.rb_h = relpos_h; sp.rb_wd = relpos_w; sp.ld_rbh = 2 * H8 - 1; sp.ld_rbw = 2 * W8 - 1; sp.rb_w = relpos_weight;
  sp.rowsum = rowsum;
  sp.rowmax = rowsum ? reinterpret_cast<unsigned*>(rowsum + (long)B * M * H8 * W8) : nullptr;
  if (mask_radius > 15 || (pos_tab && R > 15)) return CRAFT_ERR_UNSUPPORTED;
  return launch_attn_probs(sp, P, ldp, p_prec, prec, S(stream));
}

int craft_attn_probs_fused(const float* q, long ldq, const float* k, long ldk, int B, int H8, int W8, int M, int d, float scale,
                           const float* pos_tab, int R, float pos_w, int mask_radius, const unsigned* clamp_ord, void* P, long ldp,
                           float* rowsum, void* ws, int p_prec, int prec, void* stream) {
  ScoreParams sp = make_score(q, ldq, k, ldk, B, H8, W8, M, d, scale, pos_tab, R, pos_w, mask_radius, clamp_ord);
  sp.rowsum = rowsum;
  return launch_attn_probs_fused(sp, P, ldp, ws, p_prec & ~CRAFT_P_TILED, prec, (p_prec & CRAFT_P_TILED) ? 1 : 0, S(stream));
}

int craft_flash_attention(const float* q, long ldq, const float* k, long ldk, const void* vT, long ldt, int B, int H8, int W8,
                          int M, int d, int Dv, float scale, const float* pos_tab, int R, float pos_w, int mask_radius,
                          const unsigned* clamp_ord, float* O, void* ws, int score_prec, int pv_prec, void* stream) {
  ScoreParams sp = make_score(q, ldq, k, ldk, B, H8, W8, M, d, scale, pos_tab, R, pos_w, mask_radius, clamp_ord);
  return launch_flash_attn(sp, vT, ldt, Dv, O, ws, score_prec, pv_prec, S(stream));
}

int craft_attn_apply(const void* P, long ldp, const float* rowsum, const void* vT, int B, int N, int M, int Dv, float* O,
                     int prec, void* stream) {
  RowsGemmParams p = {};
  p.row_div = rowsum; p.rd_bs = N;
  const bool tiled = (prec & CRAFT_P_TILED) != 0;
  prec &= ~CRAFT_P_TILED;
  // tiled P: [B][M][ceil(N/32)] bands of 32 rows x ldp keys; V^T keeps its own key extent (N rounded up to 32)
  const long p_rows = tiled ? ((long)N + 31) / 32 * 32 : N, ldt = tiled ? ((long)N + 31) / 32 * 32 : ldp;
  if (tiled && (prec == CRAFT_PREC_F32 || ldp % 64 || ldp < N)) return CRAFT_ERR_ALIGN;
  p.a_tiled = tiled ? 1 : 0;
  p.A = P; p.lda = ldp; p.a_bs0 = (long)M * p_rows * ldp; p.a_bs1 = p_rows * ldp;
  p.B = vT; p.ldb = ldt; p.b_bs0 = (long)M * Dv * ldt; p.b_bs1 = (long)Dv * ldt;
  p.C = O; p.ldc = Dv; p.c_bs0 = (long)M * N * Dv; p.c_bs1 = (long)N * Dv;
  p.zdiv = M; p.batch = B * M; p.M = N; p.N = Dv; p.K = (int)ldp;
  p.bias = nullptr; p.scale = 1.f; p.act = CRAFT_ACT_NONE;
  const int rows32 = (prec >> CRAFT_PV_ROWS_SHIFT) & 15;
  prec &= (1 << CRAFT_PV_ROWS_SHIFT) - 1;
  if (prec == CRAFT_PREC_F32) return launch_gemm_rows(p, prec, false, S(stream));
  return launch_pv16(p, prec, rows32, S(stream));
}

int craft_forward_interpolate(const float* flow, int B, int H, int W, float* out, void* stream) {
  return launch_forward_interpolate(flow, B, H, W, out, S(stream));
}

int craft_mode_pool_ln(const float* O, const float* x, long ldx, const float* w_agg, const float* skip_coeff, int B, int N,
                       int M, int C, float* out, long ldo, void* stream) {
  return launch_mode_pool_ln(O, x, ldx, w_agg, skip_coeff, B, N, M, C, out, ldo, S(stream));
}

int craft_gma_residual(const float* mf, long ldm, const float* O, const float* gamma, int B, int N, int C, float* out,
                       long ldo, void* stream) {
  return launch_gma_residual(mf, ldm, O, gamma, B, N, C, out, ldo, S(stream));
}

#define PREC_OF(prec) ((prec) & 0xff)
#define PACKED_OF(prec) ((((prec) >> 8) & 1) && PREC_OF(prec) != CRAFT_PREC_F32)
#define W16_OF(prec) ((((prec) & CRAFT_CONV_W16) != 0) && PREC_OF(prec) == CRAFT_PREC_F16X3)

static ConvGemmParams conv_params(const float* in0, int ld0, int c0, const float* in1, int ld1, int c1, int B, int H8, int W8,
                                  int KH, int KW, const float* W, const float* bias, int cout, int epi, int act, float scale,
                                  float* out, int ldo) {
  ConvGemmParams p = {};
  p.g.seg0 = in0; p.g.ld0 = ld0; p.g.c0 = c0; p.g.seg1 = in1; p.g.ld1 = ld1; p.g.c1 = c1;
  p.g.H = H8; p.g.W = W8; p.g.KH = KH; p.g.KW = KW; p.g.padH = KH / 2; p.g.padW = KW / 2; p.g.npix = B * H8 * W8;
  p.g.stride = 1; p.g.Hin = H8; p.g.Win = W8; p.g.in_norm = nullptr;
  p.W = W; p.bias = bias; p.cout = cout; p.epi = epi; p.act = act; p.scale = scale; p.out = out; p.ldo = ldo;
  return p;
}

int craft_motion_encoder(const float* corr, long ldc, int cor_planes, const float* flow, const float* wc1, const float* bc1,
                         const float* wc2, const float* bc2, const float* wf1, const float* bf1, const float* wf2,
                         const float* bf2, const float* wcv, const float* bcv, int B, int H8, int W8, float* out, long ldo,
                         float* ws, int prec, void* flow_stream, void* flow_done, void* stream) {
  const long npix = (long)B * H8 * W8;
  float* cor1 = ws;                    // [npix][256]
  float* corflo = ws + npix * 256;     // [npix][256] = [cor (192) | flo (64)]
  float* flo1 = ws + npix * 512;       // [npix][128]
  hipStream_t s = S(stream);
  // The flow branch (convf1 -> convf2) and the correlation branch (convc1 -> convc2) are independent until `conv`:
  // with a caller-owned side stream + event the flow branch is enqueued there (the caller has already made flow_stream
  // wait for whatever produced `flow` and `ws`), the event is recorded behind it and `stream` waits for it before `conv`.
  const bool forked = flow_stream != nullptr && flow_done != nullptr && flow_stream != stream;
  hipStream_t sf = forked ? S(flow_stream) : s;
  {  // cor = relu(convc1(corr))  1x1, cor_planes -> 256   (update.py:80)
    RowsGemmParams p = {};
    p.A = corr; p.lda = ldc; p.B = wc1; p.ldb = cor_planes; p.C = cor1; p.ldc = 256;
    p.zdiv = 1; p.batch = 1; p.M = (int)npix; p.N = 256; p.K = cor_planes; p.bias = bc1; p.scale = 1.f; p.act = CRAFT_ACT_RELU;
    p.b_packed = (prec & CRAFT_W1X1_PACKED) != 0 && PREC_OF(prec) != CRAFT_PREC_F32;     // wc1 from craft_pack_weights(256, round_up(cor_planes, 32))
    TRY(launch_gemm_rows(p, PREC_OF(prec), false, s));
  }
  const int pk = PACKED_OF(prec);
  prec = PREC_OF(prec);
  // cor = relu(convc2(cor))  3x3, 256 -> 192   (update.py:81)
  {
    ConvGemmParams q = conv_params(cor1, 256, 256, nullptr, 0, 0, B, H8, W8, 3, 3, wc2, bc2, 192, CONV_EPI_BIAS_ACT,
                                   CRAFT_ACT_RELU, 1.f, corflo, 256);
    q.w_packed = pk;
    TRY(launch_gemm_conv(q, prec, s));
  }
  // flo = relu(convf1(flow))  7x7, 2 -> 128   (update.py:82)
  if (pk && prec != CRAFT_PREC_F32) TRY(launch_convf1_mfma(flow, wf1, bf1, B, H8, W8, flo1, 128, prec, sf));
  else TRY(launch_convf1(flow, wf1, bf1, B, H8, W8, flo1, 128, sf));
  // flo = relu(convf2(flo))  3x3, 128 -> 64   (update.py:83) -> columns 192..255 of corflo (the torch.cat of :85)
  {
    ConvGemmParams q = conv_params(flo1, 128, 128, nullptr, 0, 0, B, H8, W8, 3, 3, wf2, bf2, 64, CONV_EPI_BIAS_ACT,
                                   CRAFT_ACT_RELU, 1.f, corflo + 192, 256);
    q.w_packed = pk;
    TRY(launch_gemm_conv(q, prec, sf));
  }
  if (forked) {
    TRY((int)hipEventRecord(reinterpret_cast<hipEvent_t>(flow_done), sf));
    TRY((int)hipStreamWaitEvent(s, reinterpret_cast<hipEvent_t>(flow_done), 0));
  }
  // out = cat[relu(conv(cor_flo)) (126), flow (2)]  (update.py:86-87)
  ConvGemmParams p = conv_params(corflo, 256, 256, nullptr, 0, 0, B, H8, W8, 3, 3, wcv, bcv, 126, CONV_EPI_MENC,
                                 CRAFT_ACT_RELU, 1.f, out, (int)ldo);
  p.aux0 = flow; p.ld0 = 2;
  p.w_packed = pk;
  return launch_gemm_conv(p, prec, s);
}

int craft_sepconv_gru(float* hx, long ldhx, int cx, const float* wzr1, const float* bzr1, const float* wq1,
                      const float* bq1, const float* wzr2, const float* bzr2, const float* wq2, const float* bq2, int B,
                      int H8, int W8, float* ws, int prec, void* stream) {
  const long npix = (long)B * H8 * W8;
  float* z = ws;                 // [npix][128]
  float* rh = ws + npix * 128;   // [npix][128]
  hipStream_t s = S(stream);
  const float* wzr[2] = {wzr1, wzr2};
  const float* bzr[2] = {bzr1, bzr2};
  const float* wq[2] = {wq1, wq2};
  const float* bq[2] = {bq1, bq2};
  const int pk = PACKED_OF(prec);
  prec = PREC_OF(prec);
  for (int pass = 0; pass < 2; ++pass) {
    const int KH = pass == 0 ? 1 : 5, KW = pass == 0 ? 5 : 1;
    // z = sigmoid(convz(hx)), r = sigmoid(convr(hx)); rh = r*h   (update.py:51-53 / :58-60)
    ConvGemmParams a = conv_params(hx, (int)ldhx, 128 + cx, nullptr, 0, 0, B, H8, W8, KH, KW, wzr[pass], bzr[pass], 256,
                                   CONV_EPI_GRU_ZR, 0, 1.f, z, 128);
    a.aux0 = hx; a.ld0 = (int)ldhx; a.aux1 = rh; a.ld1 = 128; a.w_packed = pk;
    TRY(launch_gemm_conv(a, prec, s));
    // q = tanh(convq(cat[r*h, x])); h = (1-z)*h + z*q   (update.py:54-55 / :61-62)
    ConvGemmParams q = conv_params(rh, 128, 128, hx + 128, (int)ldhx, cx, B, H8, W8, KH, KW, wq[pass], bq[pass], 128,
                                   CONV_EPI_GRU_Q, 0, 1.f, hx, (int)ldhx);
    q.aux0 = hx; q.ld0 = (int)ldhx; q.aux1 = z; q.ld1 = 128; q.w_packed = pk;
    TRY(launch_gemm_conv(q, prec, s));
  }
  return 0;
}

int craft_sepconv_gru_context(const float* inp, long ldi, int cc, const float* wzr1, const float* bzr1, const float* wq1,
                              const float* bq1, const float* wzr2, const float* bzr2, const float* wq2, const float* bq2, int B,
                              int H8, int W8, float* fields, int prec, void* stream) {
  const int pk = PACKED_OF(prec);
  prec = PREC_OF(prec);
  const float* w[4] = {wzr1, wq1, wzr2, wq2};
  const float* bb[4] = {bzr1, bq1, bzr2, bq2};
  const int cout[4] = {256, 128, 256, 128}, off[4] = {0, 256, 384, 640};
  for (int i = 0; i < 4; ++i) {
    const int KH = i < 2 ? 1 : 5, KW = i < 2 ? 5 : 1;
    ConvGemmParams q = conv_params(inp, (int)ldi, cc, nullptr, 0, 0, B, H8, W8, KH, KW, w[i], bb[i], cout[i], CONV_EPI_BIAS_ACT,
                                   CRAFT_ACT_NONE, 1.f, fields + off[i], 768);
    q.w_packed = pk;
    TRY(launch_gemm_conv(q, prec, S(stream)));
  }
  return 0;
}

int craft_sepconv_gru_step(float* hx, long ldhx, int voff, int cv, const float* wzr1, const float* wq1, const float* wzr2,
                           const float* wq2, const float* fields, int B, int H8, int W8, float* ws, int prec, void* stream) {
  const long npix = (long)B * H8 * W8;
  float* z = ws;                 // [npix][128]
  float* rh = ws + npix * 128;   // [npix][128]
  hipStream_t s = S(stream);
  const float* wzr[2] = {wzr1, wzr2};
  const float* wq[2] = {wq1, wq2};
  const int pk = PACKED_OF(prec);
  prec = PREC_OF(prec);
  for (int pass = 0; pass < 2; ++pass) {
    const int KH = pass == 0 ? 1 : 5, KW = pass == 0 ? 5 : 1;
    // z, r from [h | v] + the hoisted context term (fields columns 0..255 / 384..639)
    ConvGemmParams a = conv_params(hx, (int)ldhx, 128, hx + voff, (int)ldhx, cv, B, H8, W8, KH, KW, wzr[pass], nullptr, 256,
                                   CONV_EPI_GRU_ZR, 0, 1.f, z, 128);
    a.aux0 = hx; a.ld0 = (int)ldhx; a.aux1 = rh; a.ld1 = 128; a.w_packed = pk;
    a.bias_field = fields + (pass == 0 ? 0 : 384); a.ld_bf = 768;
    TRY(launch_gemm_conv(a, prec, s));
    // q from [r*h | v] + context term (columns 256..383 / 640..767); h = (1-z)*h + z*q
    ConvGemmParams q = conv_params(rh, 128, 128, hx + voff, (int)ldhx, cv, B, H8, W8, KH, KW, wq[pass], nullptr, 128,
                                   CONV_EPI_GRU_Q, 0, 1.f, hx, (int)ldhx);
    q.aux0 = hx; q.ld0 = (int)ldhx; q.aux1 = z; q.ld1 = 128; q.w_packed = pk;
    q.bias_field = fields + (pass == 0 ? 256 : 640); q.ld_bf = 768;
    TRY(launch_gemm_conv(q, prec, s));
  }
  return 0;
}

int craft_flow_head(const float* h, long ldh, const float* w1, const float* b1, const float* w2, const float* b2, int B,
                    int H8, int W8, float* coords1, const float* coords0, float* flow, float* delta, float* ws, int prec,
                    void* stream) {
  ConvGemmParams q = conv_params(h, (int)ldh, 128, nullptr, 0, 0, B, H8, W8, 3, 3, w1, b1, 256, CONV_EPI_BIAS_ACT,
                                 CRAFT_ACT_RELU, 1.f, ws, 256);
  q.w_packed = PACKED_OF(prec);
  TRY(launch_gemm_conv(q, PREC_OF(prec), S(stream)));
  return launch_flow_head2(ws, w2, b2, B, H8, W8, coords1, coords0, flow, delta, S(stream));
}

int craft_mask_head(const float* h, long ldh, const float* w0, const float* b0, const float* w2, const float* b2, int B,
                    int H8, int W8, float* mask, float* ws, int prec, void* stream) {
  ConvGemmParams q = conv_params(h, (int)ldh, 128, nullptr, 0, 0, B, H8, W8, 3, 3, w0, b0, 256, CONV_EPI_BIAS_ACT,
                                 CRAFT_ACT_RELU, 1.f, ws, 256);
  q.w_packed = PACKED_OF(prec);
  TRY(launch_gemm_conv(q, PREC_OF(prec), S(stream)));
  return launch_gemm_conv(conv_params(ws, 256, 256, nullptr, 0, 0, B, H8, W8, 1, 1, w2, b2, 576, CONV_EPI_BIAS_ACT,
                                      CRAFT_ACT_NONE, 0.25f, mask, 576), PREC_OF(prec), S(stream));
}

int craft_conv2d_nhwc(const float* x, long ldx, int cin, const float* w, const float* bias, int cout, int KH, int KW, int act,
                      float* y, long ldy, int B, int H, int W, int prec, void* stream) {
  if (cin % 32) return CRAFT_ERR_ALIGN;
  ConvGemmParams q = conv_params(x, (int)ldx, cin, nullptr, 0, 0, B, H, W, KH, KW, w, bias, cout, CONV_EPI_BIAS_ACT, act, 1.f, y,
                                 (int)ldy);
  q.w_packed = PACKED_OF(prec);
  q.w16 = W16_OF(prec);
  return launch_gemm_conv(q, PREC_OF(prec), S(stream));
}

int craft_conv2d_nhwc_res(const float* x, long ldx, int cin, const float* w, const float* bias, int cout, int KH, int KW, int act, const float* res,
                          long ldr, float* y, long ldy, int B, int H, int W, int prec, void* stream) {
  if (cin % 32) return CRAFT_ERR_ALIGN;
  if (res == nullptr) return CRAFT_ERR_ARG;
  ConvGemmParams q = conv_params(x, (int)ldx, cin, nullptr, 0, 0, B, H, W, KH, KW, w, bias, cout, CONV_EPI_BIAS_ACT, act, 1.f, y, (int)ldy);
  q.w_packed = PACKED_OF(prec);
  q.res = res; q.ld_res = (int)ldr;
  return launch_gemm_conv(q, PREC_OF(prec), S(stream));
}

int craft_pack_conv_weights(const float* w0, int cout0, const float* w1, int cout1, int Cin, int KH, int KW, int a0, int a1, int b0, int b1,
                             int transposed, int prec, void* out, void* stream) {
  return launch_pack_conv_weights(w0, cout0, w1, cout1, Cin, KH, KW, a0, a1, b0, b1, transposed, prec, out, S(stream));
}

int craft_pack_conv_job_bytes(void) { return (int)pack_conv_job_bytes(); }
int craft_pack_conv_job_fill(void* job, const float* w0, int cout0, const float* w1, int cout1, int Cin, int KH, int KW, int a0, int a1, int b0,
                              int b1, int transposed, int prec, void* out) {
  return (int)fill_pack_conv_job(job, w0, cout0, w1, cout1, Cin, KH, KW, a0, a1, b0, b1, transposed, prec, out);
}
int craft_pack_conv_weights_batch(const void* jobs_dev, const int* first_block_dev, int n, int total_blocks, void* stream) {
  return launch_pack_conv_weights_batch(jobs_dev, first_block_dev, n, total_blocks, S(stream));
}

int craft_conv2d_nhwc2_mask(const float* x0, long ld0, int c0, const float* x1, long ld1, int c1, const float* w, const float* bias,
                            const float* bias_field, long ld_bf, int cout, int KH, int KW, const float* mask, long ldm, float* y, long ldy, int B,
                            int H, int W, int prec, void* stream) {
  if (c0 % 32 || c1 % 32 || c0 <= 0 || c1 < 0 || (c1 > 0 && x1 == nullptr)) return CRAFT_ERR_ALIGN;
  if ((bias == nullptr) == (bias_field == nullptr) || mask == nullptr) return CRAFT_ERR_ARG;
  ConvGemmParams q = conv_params(x0, (int)ld0, c0, c1 ? x1 : nullptr, (int)ld1, c1, B, H, W, KH, KW, w, bias, cout, CONV_EPI_BIAS_ACT, CRAFT_ACT_NONE,
                                 1.f, y, (int)ldy);
  q.bias_field = bias_field; q.ld_bf = (int)ld_bf;
  q.w_packed = PACKED_OF(prec);
  q.w16 = W16_OF(prec);
  q.mask = mask; q.ld_mask = (int)ldm;
  return launch_gemm_conv(q, PREC_OF(prec), S(stream));
}

int craft_conv2d_nhwc2(const float* x0, long ld0, int c0, const float* x1, long ld1, int c1, const float* w, const float* bias,
                       const float* bias_field, long ld_bf, int cout, int KH, int KW, int act, float* y, long ldy, int B, int H, int W, int prec,
                       void* stream) {
  if (c0 % 32 || c1 % 32 || c0 <= 0 || c1 < 0 || (c1 > 0 && x1 == nullptr)) return CRAFT_ERR_ALIGN;
  if ((bias == nullptr) == (bias_field == nullptr)) return CRAFT_ERR_ARG;          // exactly one of the two
  ConvGemmParams q = conv_params(x0, (int)ld0, c0, c1 ? x1 : nullptr, (int)ld1, c1, B, H, W, KH, KW, w, bias, cout, CONV_EPI_BIAS_ACT, act, 1.f, y,
                                 (int)ldy);
  q.bias_field = bias_field; q.ld_bf = (int)ld_bf;
  q.bf_col0 = bias_field ? ((prec >> 16) & 0xff) * 32 : 0;                 // CRAFT_CONV_FIELD_COL0
  q.w_packed = PACKED_OF(prec);
  q.w16 = W16_OF(prec);
  return launch_gemm_conv(q, PREC_OF(prec), S(stream));
}

int craft_conv2d_nhwc_ex(const float* x, long ldx, int cin, int Hin, int Win, const float* in_norm, const float* w,
                         const float* bias, int cout, int KH, int KW, int stride, int act, float* y, long ldy, int B, int Hout,
                         int Wout, double* stats, int prec, void* stream) {
  if (cin % 32 || (stride != 1 && stride != 2)) return CRAFT_ERR_ALIGN;
  ConvGemmParams q = conv_params(x, (int)ldx, cin, nullptr, 0, 0, B, Hout, Wout, KH, KW, w, bias, cout, CONV_EPI_BIAS_ACT, act, 1.f,
                                 y, (int)ldy);
  q.g.stride = stride; q.g.Hin = Hin; q.g.Win = Win; q.g.in_norm = in_norm;
  q.stats = stats;
  q.w_packed = PACKED_OF(prec);
  return launch_gemm_conv(q, PREC_OF(prec), S(stream));
}

int craft_stem_conv7x7(const float* image, const float* w, const float* bias, int act, int B, int H, int W, float* out,
                       double* stats, void* stream) {
  return launch_stem7x7(image, w, bias, act, B, H, W, out, stats, S(stream));
}

int craft_stem_conv7x7_mfma(const float* image, const void* w_packed, const float* bias, int act, int B, int H, int W,
                            float* out, double* stats, int prec, void* stream) {
  return launch_stem_mfma(image, nullptr, B, w_packed, bias, act, B, H, W, out, stats, PREC_OF(prec), S(stream));
}
int craft_stem_conv7x7_mfma_pair(const float* image_a, int Ba, const float* image_b, const void* w_packed, const float* bias, int act, int B, int H,
                                 int W, float* out, double* stats, int prec, void* stream) {
  return launch_stem_mfma(image_a, image_b, Ba, w_packed, bias, act, B, H, W, out, stats, PREC_OF(prec), S(stream));
}

int craft_stats_finalize(const double* sums, long n, double count, float eps, float* mean_rstd, void* stream) {
  return launch_stats_finalize(sums, n, count, eps, mean_rstd, S(stream));
}

int craft_residual_relu(const float* x, long ldx, const float* xnorm, const float* y, long ldy, const float* ynorm, int y_relu,
                        int B, int HW, int C, float* out, long ldo, void* stream) {
  return launch_residual_relu(x, ldx, xnorm, y, ldy, ynorm, y_relu, B, HW, C, out, ldo, S(stream));
}

int craft_pack_weights(const float* w, int rows, int K, int prec, void* out, void* stream) {
  return launch_pack_weights(w, rows, K, PREC_OF(prec), out, S(stream));
}

int craft_flow_metrics(const float* pred, const float* gt, const float* valid, int B, int H, int W, float gt_off_x,
                       float gt_off_y, float max_gt_mag, double* out16, void* stream) {
  return launch_flow_metrics(pred, gt, valid, B, H, W, gt_off_x, gt_off_y, max_gt_mag, out16, S(stream));
}

int craft_flow_l1_loss(const float* pred, const float* gt, const float* valid, int B, int H, int W, float weight, float max_flow,
                       double* loss, float* grad_pred, void* stream) {
  return launch_flow_l1(pred, gt, valid, B, H, W, weight, max_flow, loss, grad_pred, S(stream));
}

int craft_sumsq(const float* x, long n, double* out, void* stream) { return launch_sumsq(x, n, out, S(stream)); }

int craft_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int step, float grad_mul, const double* grad_sumsq,
                     float max_norm, void* stream) {
  return launch_adamw(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_mul, grad_sumsq,
                      max_norm, S(stream));
}

int craft_loss_scale_update(const double* grad_sumsq, void* state, float grad_mul, float max_norm, float beta1, float beta2, float growth,
                            float backoff, int growth_interval, void* stream) {
  return launch_scaler_update(grad_sumsq, state, grad_mul, max_norm, beta1, beta2, growth, backoff, growth_interval, S(stream));
}

int craft_adamw_step_dyn(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, const void* state, void* stream) {
  return launch_adamw_dyn(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, state, S(stream));
}

int craft_convex_upsample(const float* mask, const float* flow, int B, int H8, int W8, float* up, void* stream) {
  return launch_convex_upsample(mask, flow, B, H8, W8, up, S(stream));
}

int craft_coords_init(const float* flow_init_nchw, int B, int H8, int W8, float* coords0, float* coords1, float* flow,
                      void* stream) {
  return launch_coords_init(flow_init_nchw, B, H8, W8, coords0, coords1, flow, S(stream));
}

// ---- training (include/craft_hip.h, "training" section) -------------------------------------------------------------------
int craft_gemm(const float* A, long a_sm, long a_sk, long a_bs0, long a_bs1, const float* B, long b_sn, long b_sk, long b_bs0,
               long b_bs1, float* C, long ldc, long c_bs0, long c_bs1, int zdiv, int batch, int M, int N, int K, float alpha,
               int accumulate, int ksplit, int prec, void* stream) {
  return launch_gemm_gen(A, a_sm, a_sk, a_bs0, a_bs1, B, b_sn, b_sk, b_bs0, b_bs1, C, ldc, c_bs0, c_bs1, zdiv, batch, M, N, K, alpha,
                         accumulate, ksplit, prec, S(stream));
}
int craft_conv2d_wgrad(const float* x, long ldx, int cin, const float* dy, long ldy, int cout, int KH, int KW, int B, int H, int W,
                       float* dW, float* db, float* ws, long ws_floats, int prec, void* stream) {
  return launch_conv_wgrad(x, ldx, cin, dy, ldy, cout, KH, KW, B, H, W, dW, db, ws, ws_floats, prec, S(stream));
}
int craft_pack_operand(const float* x, long ldx, int C, long rows, int B, int H, int W, int padH, int padW, long guard, long rows_p,
                       int prec, void* out, int cg_off, int ncg_total, float* colsum, int tail, void* stream) {
  return launch_pack_operand(x, ldx, C, rows, B, H, W, padH, padW, guard, rows_p, prec, out, cg_off, ncg_total, colsum, tail, S(stream));
}
int craft_gemm_pk(const void* A, const long* a_desc, const void* B, const long* b_desc, float* C, long ldc, long c_outer, long c_inner,
                  int inner, int nbatch, int M, int N, int K, float alpha, int prec, void* stream) {
  return launch_gemm_pk(A, a_desc, B, b_desc, C, ldc, c_outer, c_inner, inner, nbatch, M, N, K, alpha, prec, S(stream));
}
int craft_pack_operands(const long* descs, int n, void* stream) { return launch_pack_operands(descs, n, S(stream)); }
int craft_wgrad_pk(const void* const* dYp, const void* const* Xp, const void* const* Xp1, int cin0, int nseg, long dy_rows_p, int cout, long x_rows_p,
                   int cin, long guard, long K, int KH, int KW, int Wp, float* dW, int prec, void* stream) {
  const int px = (prec >> 8) & 0xff;                      // CRAFT_WGRAD_X_PREC(p): the X packs' mode when it differs from dY's
  return launch_wgrad_pk(dYp, Xp, Xp1, cin0, nseg, dy_rows_p, cout, x_rows_p, cin, guard, K, KH, KW, Wp, dW, prec & 0xff, px ? px - 1 : (prec & 0xff), S(stream));
}
int craft_relpos_add(float* S, long ld, int BZ, int H8, int W8, const float* Hs, long ldh, const float* Ws, long ldw, float w, void* stream) {
  return launch_relpos_add(S, ld, BZ, H8, W8, Hs, ldh, Ws, ldw, w, S(stream));
}
int craft_relpos_bwd(const float* dS, long ld, int BZ, int H8, int W8, float* dHs, long ldh, int nh, float* dWs, long ldw, int nw, float w,
                     void* stream) {
  return launch_relpos_bwd(dS, ld, BZ, H8, W8, dHs, ldh, nh, dWs, ldw, nw, w, S(stream));
}
// ---- CNN encoders in training (kernels_enc_train.hip)
int craft_norm_act_fwd(const float* x, long ldx, const float* mean_rstd, int mr_per_image, const float* gamma, const float* beta, int act,
                       const float* res, long ldr, float* out, long ldo, int B, int N, int C, void* stream) {
  NormActParams p = {};
  p.x = x; p.ldx = ldx; p.mr = mean_rstd; p.mr_bs = mr_per_image ? C : 0; p.gamma = gamma; p.beta = beta; p.act = act;
  p.res = res; p.ldr = ldr; p.out = out; p.ldo = ldo; p.B = B; p.N = N; p.C = C;
  if (act != CRAFT_ACT_NONE && act != CRAFT_ACT_RELU) return CRAFT_ERR_UNSUPPORTED;
  return launch_norm_act_fwd(p, S(stream));
}
int craft_norm_act_bwd_reduce(const float* dy, long ldg, const float* out, long ldo, const float* x, long ldx, const float* mean_rstd,
                              int mr_per_image, const float* gamma, const float* beta, int act, int has_res, double* sums, int B, int N,
                              int C, void* stream) {
  NormActParams p = {};
  p.dy = dy; p.ldg = ldg; p.out = const_cast<float*>(out); p.ldo = ldo; p.x = x; p.ldx = ldx; p.mr = mean_rstd;
  p.mr_bs = mr_per_image ? C : 0; p.gamma = gamma; p.beta = beta; p.act = act; p.has_res = has_res; p.sums = sums; p.B = B; p.N = N; p.C = C;
  if (act != CRAFT_ACT_NONE && act != CRAFT_ACT_RELU) return CRAFT_ERR_UNSUPPORTED;
  return launch_norm_act_bwd_reduce(p, S(stream));
}
int craft_norm_act_bwd_apply(const float* dy, long ldg, const float* out, long ldo, const float* x, long ldx, const float* mean_rstd,
                             int mr_per_image, const float* gamma, const float* beta, int act, int has_res, const float* red,
                             int red_per_image, float* dx, long lddx, float* dres, long lddr, int B, int N, int C, void* stream) {
  NormActParams p = {};
  p.dy = dy; p.ldg = ldg; p.out = const_cast<float*>(out); p.ldo = ldo; p.x = x; p.ldx = ldx; p.mr = mean_rstd;
  p.mr_bs = mr_per_image ? C : 0; p.gamma = gamma; p.beta = beta; p.act = act; p.has_res = has_res; p.red = red;
  p.red_bs = red_per_image ? C : 0; p.dx = dx; p.lddx = lddx; p.dres = dres; p.lddr = lddr; p.B = B; p.N = N; p.C = C;
  if (act != CRAFT_ACT_NONE && act != CRAFT_ACT_RELU) return CRAFT_ERR_UNSUPPORTED;
  return launch_norm_act_bwd_apply(p, S(stream));
}
int craft_bn_finalize(const double* stats, int B, int C, double count, float eps, float momentum, float* mean_rstd, float* running_mean,
                      float* running_var, void* stream) {
  return launch_bn_finalize(stats, B, C, count, eps, momentum, mean_rstd, running_mean, running_var, S(stream));
}
int craft_norm_bwd_finalize(const double* sums, int B, int C, double population, int per_image, float* red, float* dgamma, float* dbeta,
                            void* stream) {
  return launch_norm_bwd_finalize(sums, B, C, population, per_image, red, dgamma, dbeta, S(stream));
}
int craft_stem_im2col(const float* image, int B, int H, int W, float* cols, void* stream) {
  return launch_stem_im2col(image, B, H, W, cols, S(stream));
}
int craft_zero_stuff2(const float* g, long ldg, int B, int Hin, int Win, int C, float* gf, long ldf, void* stream) {
  return launch_zero_stuff2(g, ldg, B, Hin, Win, C, gf, ldf, S(stream));
}
int craft_colsum(const float* x, long ld, long rows, int C, float* out, void* stream) { return launch_colsum(x, ld, rows, C, out, S(stream)); }
int craft_multi_copy(const void* const* src, const long* n, const long* dst_off, const long* chlast, int count, float* dst, void* stream) {
  if (count <= 0) return 0;
  return launch_multi_copy(src, n, dst_off, chlast, count, dst, S(stream));
}
int craft_act_fwd(const float* x, long ldx, float* y, long ldy, long rows, int C, int act, float scale, void* stream) {
  return launch_act_fwd(x, ldx, y, ldy, rows, C, act, scale, S(stream));
}
int craft_act_bwd(const float* dy, long lddy, const float* y, long ldy, float* dx, long lddx, long rows, int C, int act, float scale,
                  void* stream) {
  return launch_act_bwd(dy, lddy, nullptr, 0, y, ldy, dx, lddx, rows, C, act, scale, 0, S(stream));
}
int craft_act_bwd2(const float* dy, long lddy, const float* dy2, long lddy2, const float* y, long ldy, float* dx, long lddx, long rows, int C,
                   int act, float scale, int zero_tail, void* stream) {
  return launch_act_bwd(dy, lddy, dy2, lddy2, y, ldy, dx, lddx, rows, C, act, scale, zero_tail, S(stream));
}
int craft_dropout(const float* x, float* y, long n, float p, unsigned long long seed, void* stream) {
  return launch_dropout(x, y, n, p, seed, S(stream));
}
int craft_tokens_bwd(const float* x, long ldx, const float* dy, long lddy, float* dx, long lddx, long rows, int C, int act, int do_ln,
                     void* stream) {
  return launch_tokens_bwd(x, ldx, dy, lddy, dx, lddx, rows, C, act, do_ln, S(stream));
}
int craft_attn_softmax_fwd(float* Sc, long ld, int B, int M, int H8, int W8, const float* pos_tab, int R, float pos_w, int mask_radius,
                           const unsigned* clamp_ord, unsigned* clampbits, float* Pdrop, float drop_p, unsigned long long seed,
                           void* Ppk, long pk_rows, int pk_np, int pk_prec, void* stream) {
  return launch_attn_softmax_fwd(Sc, ld, B, M, H8, W8, pos_tab, R, pos_w, mask_radius, clamp_ord, clampbits, Pdrop, drop_p, seed, Ppk, pk_rows,
                                 pk_np, pk_prec, S(stream));
}
int craft_attn_softmax_bwd(const float* P, float* dP, long ld, int B, int M, int H8, int W8, int R, float pos_w,
                           const unsigned* clamp_ord, const unsigned* clampbits, float* dtab_rep, float drop_p, unsigned long long seed,
                           void* dSpk, long pk_rows, int pk_np, int pk_prec, void* stream) {
  return launch_attn_softmax_bwd(P, dP, ld, B, M, H8, W8, R, pos_w, clamp_ord, clampbits, dtab_rep, drop_p, seed, dSpk, pk_rows, pk_np, pk_prec,
                                 S(stream));
}
int craft_reduce_replicas(const float* rep, int nrep, int n, float* out, void* stream) {
  return launch_reduce_replicas(rep, nrep, n, out, S(stream));
}
int craft_corr_pool_fwd(const float* Sc, long ld, int B, int M, int H8, int W8, const float* pos_tab, int R, float pos_w, const float* w,
                        const unsigned* clamp_ord, float* c0, double* sums, void* stream) {
  return launch_corr_pool_fwd(Sc, ld, B, M, H8, W8, pos_tab, R, pos_w, w, clamp_ord, c0, sums, S(stream));
}
int craft_corr_lookup_bwd(const float* dout, long ldo, const float* coords, float* G0, float* G1, float* G2, float* G3, int levels, int B,
                          int H8, int W8, int radius, int lvl_stride, int col_off, void* stream) {
  return launch_corr_lookup_bwd(dout, ldo, coords, G0, G1, G2, G3, levels, B, H8, W8, radius, lvl_stride, col_off, S(stream));
}
int craft_corr_pyramid_bwd(float* G0, const float* G1, const float* G2, const float* G3, const float* c0, const float* mu_rstd, int B,
                           int H8, int W8, double* gstats, void* stream) {
  return launch_corr_pyramid_bwd(G0, G1, G2, G3, c0, mu_rstd, B, H8, W8, gstats, S(stream));
}
int craft_corr_pool_bwd(float* Sc, long ld, int B, int M, int H8, int W8, const float* pos_tab, int R, float pos_w, const float* w,
                        const unsigned* clamp_ord, const float* c0, const float* G0, const float* mu_rstd, const double* gstats,
                        int do_norm, float* dtab_rep, double* dw, void* stream) {
  return launch_corr_pool_bwd(Sc, ld, B, M, H8, W8, pos_tab, R, pos_w, w, clamp_ord, c0, G0, mu_rstd, gstats, do_norm, dtab_rep, dw,
                              S(stream));
}
int craft_mode_pool_ln_bwd(const float* O, const float* x, long ldx, const float* w_agg, const float* skip_coeff, const float* dy,
                           long lddy, int B, int N, int M, int C, float* dO, float* dx, long lddx, float* dw_rep, void* stream) {
  return launch_mode_pool_ln_bwd(O, x, ldx, w_agg, skip_coeff, dy, lddy, B, N, M, C, dO, dx, lddx, dw_rep, S(stream));
}
int craft_convex_upsample_bwd(const float* mask, long ldm, const float* flow, const float* dup, int B, int H8, int W8, float* dmask,
                              long lddm, float* dflow, long lddf, void* stream) {
  return launch_convex_upsample_bwd(mask, ldm, flow, dup, B, H8, W8, dmask, lddm, dflow, lddf, S(stream));
}

int craft_flow_tokens(const float* coords1, const float* coords0, long rows, float* flow, float* flow32, float* coords1_copy, void* stream) {
  return launch_flow_tokens(coords1, coords0, rows, flow, flow32, coords1_copy, S(stream));
}
int craft_gru_zr_fwd(const float* zr_pre, long ldzr, const float* h, long ldh, float* z, float* r, float* rh, long rows, int C, void* stream) {
  return launch_gru_zr_fwd(zr_pre, ldzr, h, ldh, z, r, rh, rows, C, S(stream));
}
int craft_gru_out_fwd(const float* q_pre, long ldq, const float* z, const float* h, long ldh, float* q, float* h_new, long ldhn, long rows,
                      int C, void* stream) {
  return launch_gru_out_fwd(q_pre, ldq, z, h, ldh, q, h_new, ldhn, rows, C, S(stream));
}
int craft_gru_out_bwd(const float* dh_new, long lddhn, const float* z, const float* q, const float* h, long ldh, float* dq_pre, float* dz,
                      float* dh, long rows, int C, float* dq_pre_sum, void* stream) {
  return launch_gru_out_bwd(dh_new, lddhn, z, q, h, ldh, dq_pre, dz, dh, rows, C, dq_pre_sum, S(stream));
}
int craft_gru_zr_bwd(const float* dz, const float* drh, long lddrh, const float* z, const float* r, const float* h, long ldh, float* dzr_pre,
                     float* dh, long rows, int C, float* dzr_pre_sum, float* dh_out, long lddho, void* stream) {
  return launch_gru_zr_bwd(dz, drh, lddrh, z, r, h, ldh, dzr_pre, dh, rows, C, dzr_pre_sum, dh_out, lddho, S(stream));
}

// ---- input pipeline ---------------------------------------------------------------------------------------------
int craft_aug_spatial(const float* src, int H, int W, int C, int do_resize, float fx, float fy, int hflip, int vflip, int y0, int x0, int ch,
                      int cw, int is_flow, float* out, void* stream) {
  return launch_aug_spatial(src, H, W, C, do_resize, fx, fy, hflip, vflip, y0, x0, ch, cw, is_flow, out, S(stream));
}
int craft_aug_sparse(const float* flow, const float* valid, int H, int W, float fx, float fy, int hflip, int y0, int x0, int ch, int cw,
                     int* owner, float* out_flow, float* out_valid, void* stream) {
  return launch_aug_sparse(flow, valid, H, W, fx, fy, hflip, y0, x0, ch, cw, owner, out_flow, out_valid, S(stream));
}
int craft_aug_photo(float* img, long npix, int op, float factor, float mean, void* stream) {
  return launch_aug_photo(img, npix, op, factor, mean, S(stream));
}
int craft_aug_erase(float* img, int H, int W, const int* rects, int nrect, float mr, float mg, float mb, void* stream) {
  return launch_aug_erase(img, H, W, rects, nrect, mr, mg, mb, S(stream));
}
int craft_aug_shift(const float* img1, const float* img2, const float* flow, int H, int W, int dx, int dy, float* out1, float* out2,
                    float* out_flow, float* valid, void* stream) {
  return launch_aug_shift(img1, img2, flow, H, W, dx, dy, out1, out2, out_flow, valid, S(stream));
}
int craft_aug_blur(const float* src, int H, int W, int C, int K, float sigma, float* out, void* stream) {
  return craft::launch_aug_blur(src, H, W, C, K, sigma, out, S(stream));
}

// ---- host-side helper of the evaluation harness (no device work) ----------------------------------------------------
// PNG scan-line unfiltering (filter types 0-4, PNG spec 9.2): rows [h][1 + stride] (filter byte + filtered bytes) -> out [h][stride].
// Average and Paeth are sequential along the line, so the Python reader (craft_amd/flow_io.py) needed one interpreter iteration per
// byte: seconds per KITTI frame.  Returns 0, or -1 on an unknown filter type.
int craft_png_unfilter(const unsigned char* rows, int h, int stride, int bpp, unsigned char* out) {
  for (int y = 0; y < h; ++y) {
    const unsigned char* in = rows + (long)y * (stride + 1);
    unsigned char* cur = out + (long)y * stride;
    const unsigned char* prev = y ? cur - stride : nullptr;
    const int ft = in[0];
    ++in;
    if (ft < 0 || ft > 4) return -1;
    for (int x = 0; x < stride; ++x) {
      const int a = x >= bpp ? cur[x - bpp] : 0, b = prev ? prev[x] : 0, c = (prev && x >= bpp) ? prev[x - bpp] : 0;
      int pr = 0;
      if (ft == 1) pr = a;
      else if (ft == 2) pr = b;
      else if (ft == 3) pr = (a + b) >> 1;
      else if (ft == 4) {
        const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
        pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
      }
      cur[x] = (unsigned char)(in[x] + pr);
    }
  }
  return 0;
}

}  // extern "C"
