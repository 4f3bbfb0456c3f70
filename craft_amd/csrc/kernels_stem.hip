// BasicEncoder.conv1 (7x7, stride 2, pad 3, 3 -> 64 channels; extractor.py:139,181) on the matrix cores, fused with the
// input normalisation 2*(x/255)-1 (network.py:169-173), bias (+folded BatchNorm + ReLU for cnet) and the InstanceNorm
// statistics for fnet.  The direct fp32 kernel (k_stem7x7, kept for the exact-fp32 policy) is VALU-bound: 147*64
// packed FMAs per pixel pair made it 0.4 ms for 8 images, more than a whole residual layer.
//
// Implicit GEMM with K ordered (ky, c, kx) and kx padded 7 -> 8: one k-group of 8 is then ONE run of 8 consecutive
// pixels of one input row and channel, i.e. 32 contiguous bytes of the staged patch -- the A fragment of a lane is four
// 8-byte LDS reads, no gather.  K = 21 groups (+3 zero groups) = 192 = 12 k-steps of 16.  Weights: [64][192] in that
// order, packed by craft_pack_weights (fragment order), streamed from L2 like every other conv here.
// Block = 8 x 16 output pixels x 64 channels; wave = 64 pixels x 32 channels.
#include "conv_epilogue.hpp"

namespace craft {

constexpr int STM_TH = 8, STM_TW = 16, STM_PH = 2 * STM_TH + 5, STM_PLD = 40;     // patch 21 rows x (37 -> 40) columns

template <int PREC>
__global__ __launch_bounds__(NTHREADS) void k_stem_mfma(const float* __restrict__ img, const float* __restrict__ img2, int bsplit, ConvGemmParams p, int H, int W) {
  typedef typename FragT<PREC>::t frag_t;
  typedef typename PrecT<PREC>::lds_t h_t;
  constexpr int PL = Planes<PREC>::N, MT = 2;
  // the patch lives in LDS as 16-bit planes (hi, and lo = x - hi for the split-fp16 product): converted ONCE while it is staged.  (Rounds
  // 2-5 kept it in fp32 and every wave converted its A fragments inside the K loop -- each patch value ~20 times per block: PMC round 6:
  // 19-22 VALU per MFMA, matrix pipe 22 % busy.)
  constexpr int NPIX = 3 * STM_PH * STM_PLD;
  __shared__ __attribute__((aligned(16))) h_t pat[PL * NPIX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H2 = p.g.H, W2 = p.g.W;
  const int tiles_x = (W2 + STM_TW - 1) / STM_TW, tiles_y = (H2 + STM_TH - 1) / STM_TH;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int oy0 = ty * STM_TH, ox0 = tx * STM_TW;
  const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
  // patch staging: all of a thread's pixels requested first, from clamped addresses, then normalised / zero-padded and published (a
  // load under the bounds test compiles to branch + load + vmcnt(0): ten dependent HBM round trips per block before its first MFMA)
  constexpr int NLD = (NPIX + NTHREADS - 1) / NTHREADS;
  float pv[NLD];
  unsigned pok = 0u;
  // images [0, bsplit) from `img`, the rest from `img2`: frames 1 and 2 of the pairs go through fnet as one batch without being
  // concatenated first (torch.cat of 2 x 22 MB: 40-58 us in front of the first kernel of the main stream, profiles/r6/timeline_fw.txt)
  const float* imb = b < bsplit ? img + (long)b * 3 * H * W : img2 + (long)(b - bsplit) * 3 * H * W;
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int i = min(tid + k * NTHREADS, NPIX - 1);
    const int c = i / (STM_PH * STM_PLD), rem = i - c * (STM_PH * STM_PLD);
    const int r = rem / STM_PLD, q = rem - r * STM_PLD;
    const int y = iy0 + r, x = ix0 + q;
    pok |= (y >= 0 && y < H && x >= 0 && x < W) ? 1u << k : 0u;
    pv[k] = imb[(c * H + min(max(y, 0), H - 1)) * W + min(max(x, 0), W - 1)];          // (one image: < 2^31 elements)
  }
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int i = tid + k * NTHREADS;
    if (i < NPIX) {
      const float x = ((pok >> k) & 1u) ? 2.f * (pv[k] / 255.f) - 1.f : 0.f;
      const h_t hi = (h_t)x;
      pat[i] = hi;
      if constexpr (PL == 2) pat[NPIX + i] = (h_t)(x - (float)hi);
    }
  }
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 32;
  const int hsel = lane >> 5;
  int base[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int r = wm0 + mt * 32 + (lane & 31);
    base[mt] = 2 * (r >> 4) * STM_PLD + 2 * (r & 15);
  }
  const uint16_t* wb = reinterpret_cast<const uint16_t*>(p.W) + (long)(wn0 / 32) * (PL * 1024) + lane * 8;
  const long kt_stride = 2L * (PL * 1024);                   // 64 output channels = 2 column blocks
  f32x16 acc[MT][1];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[mt][0][e] = 0.f;
  __syncthreads();
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int s = 0; s < 12; ++s) {
    // k-groups 2s (lanes 0-31) and 2s+1 (lanes 32-63): group g = (ky, c) = (g / 3, g % 3); groups 21..23 are zero padding -- their
    // weights (and the weight of the padding tap kx = 7) are zero, so the A fragment may hold any FINITE patch values there: group 0's
    // run is re-read (8 halves from an even column: 4-byte aligned, four ds_read_b32)
    constexpr int NG = 21;
    const int g0 = 2 * s, g1 = 2 * s + 1;
    const int off0 = g0 < NG ? ((g0 % 3) * STM_PH + g0 / 3) * STM_PLD : 0;
    const int off1 = g1 < NG ? ((g1 % 3) * STM_PH + g1 / 3) * STM_PLD : 0;
    const int off = hsel ? off1 : off0;
    frag_t bw[PL];
    {
      const uint16_t* q = wb + (long)(s >> 1) * kt_stride + (s & 1) * 512;
#pragma unroll
      for (int pl = 0; pl < PL; ++pl) bw[pl] = *reinterpret_cast<const frag_t*>(q + pl * 1024);
    }
    frag_t ah[MT], al[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const unsigned* sh = reinterpret_cast<const unsigned*>(&pat[off + base[mt]]);
      u32x4 t;
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = sh[j];
      ah[mt] = __builtin_bit_cast(frag_t, t);
      if constexpr (PL == 2) {
        const unsigned* sl = reinterpret_cast<const unsigned*>(&pat[NPIX + off + base[mt]]);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = sl[j];
        al[mt] = __builtin_bit_cast(frag_t, t);
      }
    }
    if constexpr (PL == 2) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][0] = mfma16<PREC>(al[mt], bw[0], acc[mt][0]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][0] = mfma16<PREC>(ah[mt], bw[1], acc[mt][0]);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] = mfma16<PREC>(ah[mt], bw[0], acc[mt][0]);
  }
  const long imgo = (long)b * H2 * W2;
  conv_epilogue_patch<CONV_EPI_BIAS_ACT, true, MT, 1>(p, acc, wm0, lane, wn0, imgo, oy0, ox0);
  if (p.stats) {
    const int rh4 = 4 * (lane >> 5);
    unsigned mlo = ~0u, mhi = ~0u;
    if (oy0 + STM_TH > H2 || ox0 + STM_TW > W2) {
      mlo = 0u; mhi = 0u;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = wm0 + mt * 32 + (e & 3) + 8 * (e >> 2) + rh4;
          const bool ok = (oy0 + (r >> 4)) < H2 && (ox0 + (r & 15)) < W2;
          const int bit = mt * 16 + e;
          if (ok) { if (bit < 32) mlo |= 1u << bit; else mhi |= 1u << (bit - 32); }
        }
    }
    conv_col_stats<MT, 1>(p, acc, lane, wn0, (long)b, mlo, mhi);
  }
}

// image NCHW [B][3][H][W] raw 0..255; w: craft_pack_weights(rows 64, K 192) of the (ky, c, kx8)-ordered matrix
int launch_stem_mfma(const float* img, const float* img2, int bsplit, const void* w_packed, const float* bias, int act, int B, int H, int W, float* out,
                     double* stats, int prec, hipStream_t s) {
  if ((H & 1) || (W & 1)) return CRAFT_ERR_ALIGN;
  if (bsplit < 0 || bsplit > B || (bsplit < B && !img2)) return CRAFT_ERR_ARG;
  ConvGemmParams p = {};
  p.g.H = H / 2; p.g.W = W / 2; p.g.npix = B * p.g.H * p.g.W;
  p.W = reinterpret_cast<const float*>(w_packed); p.bias = bias; p.cout = 64; p.epi = CONV_EPI_BIAS_ACT; p.act = act; p.scale = 1.f;
  p.out = out; p.ldo = 64; p.stats = stats;
  const int tiles = ((p.g.W + STM_TW - 1) / STM_TW) * ((p.g.H + STM_TH - 1) / STM_TH) * B;
  if (prec == CRAFT_PREC_BF16) hipLaunchKernelGGL((k_stem_mfma<CRAFT_PREC_BF16>), dim3(tiles), dim3(NTHREADS), 0, s, img, img2, bsplit, p, H, W);
  else if (prec == CRAFT_PREC_F16) hipLaunchKernelGGL((k_stem_mfma<CRAFT_PREC_F16>), dim3(tiles), dim3(NTHREADS), 0, s, img, img2, bsplit, p, H, W);
  else if (prec == CRAFT_PREC_F16X3) hipLaunchKernelGGL((k_stem_mfma<CRAFT_PREC_F16X3>), dim3(tiles), dim3(NTHREADS), 0, s, img, img2, bsplit, p, H, W);
  else return CRAFT_ERR_ARG;
  return (int)hipGetLastError();
}

}  // namespace craft
