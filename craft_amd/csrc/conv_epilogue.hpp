// Fused epilogues of the update-block convolutions, shared by k_gemm_conv (generic implicit GEMM) and
// k_conv_halo (halo-tile conv).  `row` is the token index of the output pixel, `col` the output channel.
#pragma once
#include "gemm_engine.hpp"
#include "launch.hpp"

namespace craft {

__device__ __forceinline__ void conv_epilogue(const ConvGemmParams& p, long row, int col, float v) {
  const int N = p.cout;
  switch (p.epi) {
    case CONV_EPI_BIAS_ACT:          // out = act(conv + bias) * scale
      if (col < N) {
        v += p.bias[col];
        if (p.act == CRAFT_ACT_RELU) v = fmaxf(v, 0.f);
        p.out[row * p.ldo + col] = v * p.scale;
      }
      break;
    case CONV_EPI_GRU_ZR:            // cols [0,128): z = sigmoid -> out ; cols [128,256): r = sigmoid, r*h -> aux1
      if (col < N) {
        const float s = sigmoid_precise(v + p.bias[col]);
        if (col < 128) p.out[row * p.ldo + col] = s;
        else p.aux1[row * p.ld1 + (col - 128)] = s * p.aux0[row * p.ld0 + (col - 128)];
      }
      break;
    case CONV_EPI_GRU_Q:             // q = tanh; h' = (1-z) h + z q  (z = aux1, h = aux0; out may alias h)
      if (col < N) {
        const float q = tanhf(v + p.bias[col]);
        const float z = p.aux1[row * p.ld1 + col];
        const float h = p.aux0[row * p.ld0 + col];
        p.out[row * p.ldo + col] = (1.f - z) * h + z * q;
      }
      break;
    case CONV_EPI_MENC:              // cols [0,N): relu(conv) ; cols N, N+1: the 2 flow channels (update.py:86-87)
      if (col < N) p.out[row * p.ldo + col] = fmaxf(v + p.bias[col], 0.f);
      else if (col < N + 2) p.out[row * p.ldo + col] = p.aux0[row * p.ld0 + (col - N)];
      break;
  }
}

}  // namespace craft
