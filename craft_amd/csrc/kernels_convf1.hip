// BasicMotionEncoder.convf1 (update.py:75,82): 7x7 convolution of the 2-channel flow field to 128 channels + ReLU, on the
// matrix cores.  Implicit GEMM  out[pix][co] = sum_k A[pix][k] W[co][k]  with k = ky*16 + kx*2 + c (kx = 7 is padding:
// K = 7 * 16 = 112, one MFMA k-step of 16 per kernel row), so the 8 consecutive k of an A fragment are 8 consecutive floats
// of one row of the flow patch: fragments are built straight from the LDS patch, no im2col.
//   block = 8 x 16 pixels x 128 channels, 4 waves; wave w owns tile rows 2w, 2w+1 (32 pixels) x all 128 channels.
//   LDS: flow patch (8+6) x (16+8) x 2 floats (row pitch 96 floats: the two pixel rows of a wave land on disjoint banks) and
//   the whole weight matrix in MFMA fragment order (craft_pack_weights(rows 128, K 128, prec): 57 KB for the two f16x3 planes).
// The VALU form of this layer (k_convf1) took 102 us per call at 448x1024 batch 4 for 1.4 GFLOP and was the critical path of the
// motion encoder's flow branch.
#include "launch.hpp"

namespace craft {

constexpr int F1_TH = 8, F1_TW = 16, F1_PH = F1_TH + 6, F1_LD = 96;

template <int PREC>
__global__ __launch_bounds__(256) void k_convf1_mfma(const float* __restrict__ flow, const uint16_t* __restrict__ wpk,
                                                     const float* __restrict__ bias, int H8, int W8, float* __restrict__ out, long ldo) {
  typedef typename PrecT<PREC>::lds_t h_t;
  typedef typename std::conditional<PREC == CRAFT_PREC_BF16, bf16x8, f16x8>::type frag_t;
  constexpr int PL = Planes<PREC>::N;
  constexpr int NFRAG = 7 * 4 * PL;                            // LDS: [ky 7][nb 4][pl][lane 64][8] (the 8th k-step of K = 128 is padding)
  constexpr int WELEMS = NFRAG * 512;
  __shared__ __attribute__((aligned(16))) float patch[F1_PH * F1_LD];
  __shared__ __attribute__((aligned(16))) uint16_t wl[WELEMS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tx_n = (W8 + F1_TW - 1) / F1_TW, ty_n = (H8 + F1_TH - 1) / F1_TH;
  int bid = blockIdx.x;
  const int tx0 = (bid % tx_n) * F1_TW; bid /= tx_n;
  const int ty0 = (bid % ty_n) * F1_TH;
  const int b = bid / ty_n;
  const long img = (long)b * H8 * W8;
  // weights: one 16-byte copy per thread and step
  // (all of a thread's weight pieces and patch pixels are REQUESTED first and stored to LDS afterwards: as a rolled copy loop each of the
  // 14 pieces was load -> s_waitcnt vmcnt(0) -> ds_write, 14 dependent L2 round trips in front of a block's 84 MFMAs)
  constexpr int NWP = NFRAG * 64 / 256;                        // 16-byte pieces per thread (14 for f16x3, 7 for one plane)
  static_assert(NFRAG * 64 % 256 == 0, "weight pieces per thread");
  uint4 wreg[NWP];
#pragma unroll
  for (int k = 0; k < NWP; ++k) {
    const int i = tid + 256 * k;
    const int f = i >> 6, pl = f % PL, n = (f / PL) & 3, ky = f / (4 * PL);
    const int src = (((ky >> 1) * 4 + n) * PL + pl) * 2 + (ky & 1);            // craft_pack_weights fragment index
    wreg[k] = reinterpret_cast<const uint4*>(wpk)[src * 64 + (i & 63)];
  }
  // flow patch rows ty0-3 .. ty0+10, columns tx0-3 .. tx0+20 (24 columns x 2 channels), zero outside the image (clamped loads)
  constexpr int NPP = (F1_PH * 24 + 255) / 256;
  float2 preg[NPP];
  bool pok[NPP];
#pragma unroll
  for (int k = 0; k < NPP; ++k) {
    const int i = min(tid + 256 * k, F1_PH * 24 - 1);
    const int py = i / 24, px = i - py * 24;
    const int y = ty0 + py - 3, x = tx0 + px - 3;
    pok[k] = y >= 0 && y < H8 && x >= 0 && x < W8;
    preg[k] = *reinterpret_cast<const float2*>(flow + (img + (long)min(max(y, 0), H8 - 1) * W8 + min(max(x, 0), W8 - 1)) * 2);
  }
#pragma unroll
  for (int k = 0; k < NWP; ++k) reinterpret_cast<uint4*>(wl)[tid + 256 * k] = wreg[k];
#pragma unroll
  for (int k = 0; k < NPP; ++k) {
    const int i = tid + 256 * k;
    if (i < F1_PH * 24) {
      const int py = i / 24, px = i - py * 24;
      *reinterpret_cast<float2*>(&patch[py * F1_LD + px * 2]) = pok[k] ? preg[k] : make_float2(0.f, 0.f);
    }
  }
  __syncthreads();
  const int m = lane & 31, g = lane >> 5;
  const int ty = 2 * wave + (m >> 4), tx = m & 15;
  f32x16 acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
#pragma unroll
  for (int ky = 0; ky < 7; ++ky) {
    const float* p = &patch[(ty + ky) * F1_LD + (tx + 4 * g) * 2];
    const float2 v0 = *reinterpret_cast<const float2*>(p), v1 = *reinterpret_cast<const float2*>(p + 2),
                 v2 = *reinterpret_cast<const float2*>(p + 4), v3 = *reinterpret_cast<const float2*>(p + 6);
    const float4 lo4 = {v0.x, v0.y, v1.x, v1.y}, hi4 = {v2.x, v2.y, v3.x, v3.y};
    frag_t ah, al;
    if constexpr (PREC == CRAFT_PREC_F16X3) {
      f16x4 h0, l0, h1, l1;
      split_f16x3(lo4, h0, l0);
      split_f16x3(hi4, h1, l1);
#pragma unroll
      for (int j = 0; j < 4; ++j) { ah[j] = h0[j]; ah[4 + j] = h1[j]; al[j] = l0[j]; al[4 + j] = l1[j]; }
    } else {
      ah[0] = (h_t)lo4.x; ah[1] = (h_t)lo4.y; ah[2] = (h_t)lo4.z; ah[3] = (h_t)lo4.w;
      ah[4] = (h_t)hi4.x; ah[5] = (h_t)hi4.y; ah[6] = (h_t)hi4.z; ah[7] = (h_t)hi4.w;
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const uint16_t* wb = &wl[(((ky * 4 + n) * PL) * 64 + lane) * 8];
      const frag_t bh = *reinterpret_cast<const frag_t*>(wb);
      if constexpr (PREC == CRAFT_PREC_F16X3) {
        const frag_t bl = *reinterpret_cast<const frag_t*>(wb + 512);           // plane 1 of the same (ky, nb)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[n], 0, 0, 0);
      } else if constexpr (PREC == CRAFT_PREC_BF16) {
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[n], 0, 0, 0);
      } else {
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[n], 0, 0, 0);
      }
    }
  }
  // C layout: column (channel within the 32-wide tile) = lane & 31, row (pixel) = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int co = n * 32 + (lane & 31);
    const float bv = bias[co];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int mm = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      const int y = ty0 + 2 * wave + (mm >> 4), x = tx0 + (mm & 15);
      if (y < H8 && x < W8) out[(img + (long)y * W8 + x) * ldo + co] = fmaxf(acc[n][e] + bv, 0.f);
    }
  }
}

int launch_convf1_mfma(const float* flow, const void* w_packed, const float* bias, int B, int H8, int W8, float* out, long ldo, int prec,
                       hipStream_t s) {
  const int tiles = ((W8 + F1_TW - 1) / F1_TW) * ((H8 + F1_TH - 1) / F1_TH) * B;
  if (tiles <= 0) return 0;
  const uint16_t* w = reinterpret_cast<const uint16_t*>(w_packed);
  if (prec == CRAFT_PREC_F16X3) hipLaunchKernelGGL((k_convf1_mfma<CRAFT_PREC_F16X3>), dim3(tiles), dim3(256), 0, s, flow, w, bias, H8, W8, out, ldo);
  else if (prec == CRAFT_PREC_F16) hipLaunchKernelGGL((k_convf1_mfma<CRAFT_PREC_F16>), dim3(tiles), dim3(256), 0, s, flow, w, bias, H8, W8, out, ldo);
  else if (prec == CRAFT_PREC_BF16) hipLaunchKernelGGL((k_convf1_mfma<CRAFT_PREC_BF16>), dim3(tiles), dim3(256), 0, s, flow, w, bias, H8, W8, out, ldo);
  else return CRAFT_ERR_ARG;
  return (int)hipGetLastError();
}

}  // namespace craft
