// Fused epilogues of the update-block convolutions, shared by k_gemm_conv (generic implicit GEMM) and
// k_conv_halo (halo-tile conv).  `row` is the token index of the output pixel, `col` the output channel.
#pragma once
#include "gemm_engine.hpp"
#include "launch.hpp"

namespace craft {

__device__ __forceinline__ void conv_epilogue(const ConvGemmParams& p, long row, int col, float v) {
  const int N = p.cout;
  const float bcol = (col < N) ? (p.bias_field ? p.bias_field[row * p.ld_bf + col] : p.bias[col]) : 0.f;
  switch (p.epi) {
    case CONV_EPI_BIAS_ACT:          // out = act(conv + bias) * scale
      if (col < N) {
        v += bcol;
        if (p.act == CRAFT_ACT_RELU) v = fmaxf(v, 0.f);
        p.out[row * p.ldo + col] = v * p.scale;
      }
      break;
    case CONV_EPI_GRU_ZR:            // cols [0,128): z = sigmoid -> out ; cols [128,256): r = sigmoid, r*h -> aux1
      if (col < N) {
        const float s = sigmoid_precise(v + bcol);
        if (col < 128) p.out[row * p.ldo + col] = s;
        else p.aux1[row * p.ld1 + (col - 128)] = s * p.aux0[row * p.ld0 + (col - 128)];
      }
      break;
    case CONV_EPI_GRU_Q:             // q = tanh; h' = (1-z) h + z q  (z = aux1, h = aux0; out may alias h)
      if (col < N) {
        const float q = tanhf(v + bcol);
        const float z = p.aux1[row * p.ld1 + col];
        const float h = p.aux0[row * p.ld0 + col];
        p.out[row * p.ldo + col] = (1.f - z) * h + z * q;
      }
      break;
    case CONV_EPI_MENC:              // cols [0,N): relu(conv) ; cols N, N+1: the 2 flow channels (update.py:86-87)
      if (col < N) p.out[row * p.ldo + col] = fmaxf(v + bcol, 0.f);
      else if (col < N + 2) p.out[row * p.ldo + col] = p.aux0[row * p.ld0 + (col - N)];
      break;
  }
}

// Per-(image, channel) moments of the biased conv output for a lazy InstanceNorm: column sums of this wave's
// accumulator tile over its valid rows -> double atomics into stats[(b*cout + col)*2 + {0,1}].
// rowmask bit (mt*16 + e) set = that accumulator row of this lane is a valid output pixel of image b.
template <int MT, int NT>
__device__ __forceinline__ void conv_col_stats(const ConvGemmParams& p, f32x16 (&acc)[MT][NT], int lane, int cb, long b,
                                               unsigned rowmask_lo, unsigned rowmask_hi) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = cb + nt * 32 + (lane & 31);
    const float bias = (col < p.cout) ? p.bias[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int bit = mt * 16 + e;
        const bool ok = bit < 32 ? (rowmask_lo >> bit) & 1u : (rowmask_hi >> (bit - 32)) & 1u;
        const float v = ok ? acc[mt][nt][e] + bias : 0.f;
        s1 += v;
        s2 += v * v;
      }
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (lane < 32 && col < p.cout) {
      atomicAdd(&p.stats[(b * p.cout + col) * 2], (double)s1);
      atomicAdd(&p.stats[(b * p.cout + col) * 2 + 1], (double)s2);
    }
  }
}

}  // namespace craft
