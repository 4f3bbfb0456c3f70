// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define CRAFT_PREC_F32 0
#define CRAFT_PREC_BF16 1
#define CRAFT_PREC_F16 2
#define CRAFT_PREC_F16X3 3   // split-fp16 emulation of fp32 products (gemm_engine.hpp)

#define CRAFT_ACT_NONE 0
#define CRAFT_ACT_TANH 1
#define CRAFT_ACT_RELU 2
#define CRAFT_ACT_SIGMOID 3

// developer experiment (tools/build_variant.py -DCRAFT_X3_TERMS=5|6): which terms of the split-fp16 product the CONVOLUTION kernels
// execute -- bit 0: lo(activation) x hi(weight), bit 1: hi(activation) x lo(weight), bit 2: hi x hi.  7 = the shipped f16x3.
#ifndef CRAFT_X3_TERMS
#define CRAFT_X3_TERMS 7
#endif

#define CRAFT_ATTN_CLIP 100.0f   // setrans.py:98
#define CRAFT_LN_EPS 1e-12f      // setrans.py:715, :362; corr.py:203
#define CRAFT_STATS_REPLICAS 64  // == include/craft_hip.h

// XCD-aware work mapping.  The hardware deals consecutive block ids round-robin over the 8 XCDs (block b runs on XCD b % 8, each
// with a private L2).  xcd_chunk(b, total) renumbers the blocks so that XCD x owns a CONTIGUOUS eighth of the work list:
// blocks that re-read the same operand (adjacent in the list) then share one L2 instead of pulling 8 copies through the fabric.
// Bijective for any block count.
__device__ __forceinline__ int xcd_chunk(int bid, int total) {
  const int xcd = bid & 7, qn = total >> 3, rn = total & 7;
  return (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case CRAFT_ACT_TANH: return tanhf(v);
    case CRAFT_ACT_RELU: return fmaxf(v, 0.f);
    case CRAFT_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}
__device__ __forceinline__ float sigmoid_precise(float v) { return 1.f / (1.f + expf(-v)); }

// order-preserving float <-> uint map, for atomicMax on floats
__device__ __forceinline__ unsigned f2ord(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// sliding positional bias: biases[(dh+R)*(2R+1) + (dw+R)] if |dh|<=R && |dw|<=R else 0
// (closed form of SlidingPosBiases2D.forward, setrans.py:690-708)
__device__ __forceinline__ float pos_bias_at(const float* __restrict__ tab, int R, int h1, int w1, int h2, int w2) {
  int dh = h2 - h1, dw = w2 - w1;
  if (dh < -R || dh > R || dw < -R || dw > R) return 0.f;
  return tab[(dh + R) * (2 * R + 1) + (dw + R)];
}

// fp32 -> two fp16 planes (hi = fp16(v), lo = fp16(v - hi)) of the f16x3 scheme, 4 values: 2 v_cvt_pk_f16_f32 (RNE) +
// 4 v_fma_mix_f32 (float(hi) * -1 + v with the fp16 operand read straight from a half of the packed register, one rounding:
// the same value as v - (float)hi) + 2 v_cvt_pk_f16_f32 = 8 VALU instead of 20 for cvt / cvt-back / sub per element.
__device__ __forceinline__ void split_f16x3(const float4& v, f16x4& h, f16x4& l) {
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  f16x2 h0, h1;
  h0[0] = (_Float16)v.x; h0[1] = (_Float16)v.y; h1[0] = (_Float16)v.z; h1[1] = (_Float16)v.w;
  const unsigned u0 = __builtin_bit_cast(unsigned, h0), u1 = __builtin_bit_cast(unsigned, h1);
  float lx, ly, lz, lw;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lx) : "v"(u0), "v"(v.x));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ly) : "v"(u0), "v"(v.y));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lz) : "v"(u1), "v"(v.z));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lw) : "v"(u1), "v"(v.w));
  h[0] = h0[0]; h[1] = h0[1]; h[2] = h1[0]; h[3] = h1[1];
  l[0] = (_Float16)lx; l[1] = (_Float16)ly; l[2] = (_Float16)lz; l[3] = (_Float16)lw;
}

#define HIP_CHECK_RET(expr)                 \
  do {                                      \
    hipError_t _e = (expr);                 \
    if (_e != hipSuccess) return (int)_e;   \
  } while (0)
