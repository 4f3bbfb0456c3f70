// Fused epilogues of the update-block convolutions, shared by k_gemm_conv (generic implicit GEMM) and
// k_conv_halo (halo-tile conv).  `row` is the token index of the output pixel, `col` the output channel.
#pragma once
#include "gemm_engine.hpp"
#include "launch.hpp"

namespace craft {

// 16-bit MFMA operand fragment (8 values per lane) and the 32x32x16 MFMA of a precision code
template <int PREC> struct FragT;
template <> struct FragT<CRAFT_PREC_BF16> { typedef bf16x8 t; };
template <> struct FragT<CRAFT_PREC_F16> { typedef f16x8 t; };
template <> struct FragT<CRAFT_PREC_F16X3> { typedef f16x8 t; };

template <int PREC>
__device__ __forceinline__ f32x16 mfma16(typename FragT<PREC>::t a, typename FragT<PREC>::t b, f32x16 c) {
  if constexpr (PREC == CRAFT_PREC_BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}


// sigmoid / tanh.  FAST: v_exp_f32 + v_rcp_f32 (1 ulp each, saturating correctly at +-inf) for the 16-bit / split
// MFMA modes; the exact fp32 mode keeps the libm-grade expf / tanhf so that it stays comparable to the oracle
// at fp32 round-off.
template <bool FAST> __device__ __forceinline__ float epi_sigmoid(float v) {
  if constexpr (FAST) return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
  else return sigmoid_precise(v);
}
template <bool FAST> __device__ __forceinline__ float epi_tanh(float v) {
  if constexpr (FAST) return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * v));
  else return tanhf(v);
}

// Epilogue of FOUR consecutive output pixels (tokens row0 .. row0+3, the first `nvalid` of them real) of output
// channel `col`: v[i] is the raw accumulator of pixel i.  The epilogue kind is a template parameter -- the switch
// on p.epi is taken ONCE per kernel (CONV_EPI_DISPATCH), not once per element -- and the full group (nvalid == 4)
// runs without exec-mask branches so that its loads are issued back to back.
template <int EPI, bool FAST>
__device__ __forceinline__ void conv_epilogue4(const ConvGemmParams& p, long row0, int nvalid, int col, const float (&v)[4]) {
  const int N = p.cout;
  if (nvalid <= 0 || col >= (EPI == CONV_EPI_MENC ? N + 2 : N)) return;
  if (EPI == CONV_EPI_MENC && col >= N) {     // the 2 pass-through flow channels (update.py:86-87)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < nvalid) p.out[(row0 + i) * p.ldo + col] = p.aux0[(row0 + i) * p.ld0 + (col - N)];
    return;
  }
  float b[4];
  if (p.bias_field) {
    const float* bf = p.bias_field + row0 * p.ld_bf + col;
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = col >= p.bf_col0 ? bf[(long)min(i, nvalid - 1) * p.ld_bf] : 0.f;
  } else {
    const float b0 = p.bias[col];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = b0;
  }
  float o[4];
  float* dst; long ldd;
  if constexpr (EPI == CONV_EPI_BIAS_ACT) {            // out = act(conv + bias) * scale
    const bool relu = p.act == CRAFT_ACT_RELU;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float t = v[i] + b[i]; o[i] = (relu ? fmaxf(t, 0.f) : t) * p.scale; }
    if (p.res) {                                       // fused residual tail: relu(x + y)
      const float* rp = p.res + row0 * p.ld_res + col;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = fmaxf(o[i] + rp[(long)min(i, nvalid - 1) * p.ld_res], 0.f);
    }
    if (p.mask) {                                      // fused ReLU backward of the layer below
      const float* mp = p.mask + row0 * p.ld_mask + col;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = mp[(long)min(i, nvalid - 1) * p.ld_mask] > 0.f ? o[i] : 0.f;
    }
    dst = p.out + row0 * p.ldo + col; ldd = p.ldo;
  } else if constexpr (EPI == CONV_EPI_GRU_ZR) {       // cols [0,128): z = sigmoid -> out ; cols [128,256): r*h -> aux1
    if (col < 128) {                                   // (wave-uniform: a wave's 32 columns never straddle 128)
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = epi_sigmoid<FAST>(v[i] + b[i]);
      dst = p.out + row0 * p.ldo + col; ldd = p.ldo;
    } else {
      const float* hp = p.aux0 + row0 * p.ld0 + (col - 128);
      float h[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) h[i] = hp[(long)min(i, nvalid - 1) * p.ld0];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = epi_sigmoid<FAST>(v[i] + b[i]) * h[i];
      dst = p.aux1 + row0 * p.ld1 + (col - 128); ldd = p.ld1;
    }
  } else if constexpr (EPI == CONV_EPI_GRU_Q) {        // q = tanh; h' = (1-z) h + z q  (z = aux1, h = aux0; out may alias h)
    const float* zp = p.aux1 + row0 * p.ld1 + col;
    const float* hp = p.aux0 + row0 * p.ld0 + col;
    float z[4], h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const long j = min(i, nvalid - 1); z[i] = zp[j * p.ld1]; h[i] = hp[j * p.ld0]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (1.f - z[i]) * h[i] + z[i] * epi_tanh<FAST>(v[i] + b[i]);
    dst = p.out + row0 * p.ldo + col; ldd = p.ldo;
  } else {                                             // CONV_EPI_MENC, conv columns: relu(conv + bias)
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = fmaxf(v[i] + b[i], 0.f);
    dst = p.out + row0 * p.ldo + col; ldd = p.ldo;
  }
  if (nvalid >= 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i * ldd] = o[i];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i < nvalid) dst[i * ldd] = o[i];
  }
}

// The same epilogue in two steps for a group whose 4 pixels and column are all real (the common case: a patch inside the image): the
// operands it reads -- per-pixel bias field, h, z -- are LOADED for several groups first (epi_load4) and only then used (epi_finish4).
// Inside conv_epilogue4 every group's loads sit behind that group's bounds branches, so each group waits for its own round trip
// (s_waitcnt vmcnt(0) per group: 16 dependent L2 round trips per lane at the end of every convolution launch).
struct EpiOperands { float b[4], x0[4], x1[4]; };
// rows[i]: the token index pixel i READS its operands from (a clamped, always valid index for a pixel outside the image)
template <int EPI>
__device__ __forceinline__ void epi_load4r(const ConvGemmParams& p, const long (&rows)[4], int col, EpiOperands& o) {
  if (p.bias_field) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o.b[i] = col >= p.bf_col0 ? p.bias_field[rows[i] * p.ld_bf + col] : 0.f;
  } else {
    const float b0 = p.bias[col];
#pragma unroll
    for (int i = 0; i < 4; ++i) o.b[i] = b0;
  }
  if constexpr (EPI == CONV_EPI_BIAS_ACT) {
    if (p.res) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o.x0[i] = p.res[rows[i] * p.ld_res + col];
    }
    if (p.mask) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o.x1[i] = p.mask[rows[i] * p.ld_mask + col];
    }
  } else if constexpr (EPI == CONV_EPI_GRU_ZR) {
    if (col >= 128) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o.x0[i] = p.aux0[rows[i] * p.ld0 + (col - 128)];
    }
  } else if constexpr (EPI == CONV_EPI_GRU_Q) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { o.x1[i] = p.aux1[rows[i] * p.ld1 + col]; o.x0[i] = p.aux0[rows[i] * p.ld0 + col]; }
  }
}
template <int EPI>
__device__ __forceinline__ void epi_load4(const ConvGemmParams& p, long row0, int col, EpiOperands& o) {
  const long rows[4] = {row0, row0 + 1, row0 + 2, row0 + 3};
  epi_load4r<EPI>(p, rows, col, o);
}
// okmask bit i: pixel i is a real output pixel (its result is stored at row0 + i); ALL: every bit set (no predicates at all)
template <int EPI, bool FAST, bool ALL = true>
__device__ __forceinline__ void epi_finish4(const ConvGemmParams& p, long row0, int col, const float (&v)[4], const EpiOperands& o, unsigned okmask = 15u) {
  float r[4];
  float* dst; long ldd;
  if constexpr (EPI == CONV_EPI_BIAS_ACT) {
    const bool relu = p.act == CRAFT_ACT_RELU;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float t = v[i] + o.b[i]; r[i] = (relu ? fmaxf(t, 0.f) : t) * p.scale; }
    if (p.res) {
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = fmaxf(r[i] + o.x0[i], 0.f);
    }
    if (p.mask) {
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = o.x1[i] > 0.f ? r[i] : 0.f;
    }
    dst = p.out + row0 * p.ldo + col; ldd = p.ldo;
  } else if constexpr (EPI == CONV_EPI_GRU_ZR) {
    if (col < 128) {
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = epi_sigmoid<FAST>(v[i] + o.b[i]);
      dst = p.out + row0 * p.ldo + col; ldd = p.ldo;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = epi_sigmoid<FAST>(v[i] + o.b[i]) * o.x0[i];
      dst = p.aux1 + row0 * p.ld1 + (col - 128); ldd = p.ld1;
    }
  } else if constexpr (EPI == CONV_EPI_GRU_Q) {
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (1.f - o.x1[i]) * o.x0[i] + o.x1[i] * epi_tanh<FAST>(v[i] + o.b[i]);
    dst = p.out + row0 * p.ldo + col; ldd = p.ldo;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = fmaxf(v[i] + o.b[i], 0.f);
    dst = p.out + row0 * p.ldo + col; ldd = p.ldo;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (ALL || ((okmask >> i) & 1u)) dst[i * ldd] = r[i];
}

// run BODY(EPI) with the compile-time epilogue kind that matches p.epi
#define CONV_EPI_DISPATCH(p, BODY)                                  \
  switch ((p).epi) {                                                \
    case CONV_EPI_BIAS_ACT: { BODY(CONV_EPI_BIAS_ACT) } break;      \
    case CONV_EPI_GRU_ZR: { BODY(CONV_EPI_GRU_ZR) } break;          \
    case CONV_EPI_GRU_Q: { BODY(CONV_EPI_GRU_Q) } break;            \
    default: { BODY(CONV_EPI_MENC) } break;                         \
  }

// Epilogue of a halo-conv wave tile: accumulator rows are patch pixels, GEMM row r -> token (y0 + r/16, x0 + r%16);
// the 4 rows of one accumulator quad (e = 4q..4q+3) are 4 consecutive pixels of one patch row.
// PERM: the 32 GEMM rows of a fragment are assigned to the two 16-pixel patch rows by patch_row_perm (below) instead
// of linearly (quads stay quads, so the 4 rows of an accumulator quad are still 4 consecutive pixels).
//
// patch_row_perm: ds_read_b128 is serviced in lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32 for the upper
// half-wave).  With the linear assignment a group reads 8 pixels of one patch row and 8 of the next, whose LDS
// addresses are HWd halo rows apart -- 2-way bank conflicts unless HWd is a multiple of 16 (PMC: 36 % of the LDS cycles
// of k_conv_halo_wf).  Assigning quad qi = l >> 2 to patch row parity(qi) and pixel quad qi >> 1 makes every lane group
// read 16 consecutive pixels of ONE patch row: conflict-free for every tap and halo width.
__device__ __forceinline__ int patch_row_perm(int l) {          // l in 0..31 -> 16 * row + x
  const int qi = l >> 2;
  return (((qi ^ (qi >> 1) ^ (qi >> 2)) & 1) << 4) | ((qi >> 1) << 2) | (l & 3);
}
template <int EPI, bool FAST, int MT, int NT, bool PERM = false>
__device__ __forceinline__ void conv_epilogue_patch(const ConvGemmParams& p, const f32x16 (&acc)[MT][NT], int wm0, int lane,
                                                    int cb, long img, int y0, int x0) {
  const int c_lane = lane & 31, rh4 = 4 * (lane >> 5);
  // whole patch inside the image and every column of this wave a real output channel (wave-uniform): the operands of the four groups
  // of a row fragment are requested together (epi_load4), see there
  if (y0 + 8 <= p.g.H && x0 + 16 <= p.g.W && cb + NT * 32 <= p.cout) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        EpiOperands ops[4];
        long rows[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = wm0 + mt * 32 + (PERM ? patch_row_perm(8 * q + rh4) : 8 * q + rh4);
          rows[q] = img + (long)(y0 + (r >> 4)) * p.g.W + x0 + (r & 15);
          epi_load4<EPI>(p, rows[q], cb + nt * 32 + c_lane, ops[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v[4] = {acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
          epi_finish4<EPI, FAST>(p, rows[q], cb + nt * 32 + c_lane, v, ops[q]);
        }
      }
    return;
  }
  // RAGGED patch (the image is not a multiple of 8 x 16: 46 x 62 at configs[3]) with full columns in the same batched form -- operands
  // read from CLAMPED pixels (always valid addresses, requested together), results stored under a per-pixel predicate -- instead of the
  // per-group path below.
#ifdef CRAFT_EPI_RAGGED_BATCHED  // (developer A/B, tools/build_variant.py; measured SLOWER at configs[3]: 53.45 against 53.15 ms per step -- the
                                 // per-group path skips the quads outside the image altogether -- and therefore off)
  if (cb + NT * 32 <= p.cout) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        EpiOperands ops[4];
        long row0[4];
        unsigned ok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = wm0 + mt * 32 + (PERM ? patch_row_perm(8 * q + rh4) : 8 * q + rh4);
          const int y = y0 + (r >> 4), x = x0 + (r & 15);
          const int yc = min(y, p.g.H - 1);
          const int nv = y < p.g.H ? max(0, min(4, p.g.W - x)) : 0;
          ok[q] = (1u << nv) - 1u;
          row0[q] = img + (long)y * p.g.W + x;
          long rr[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) rr[i] = img + (long)yc * p.g.W + min(x + i, p.g.W - 1);
          epi_load4r<EPI>(p, rr, cb + nt * 32 + c_lane, ops[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v[4] = {acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
          epi_finish4<EPI, FAST, false>(p, row0[q], cb + nt * 32 + c_lane, v, ops[q], ok[q]);
        }
      }
    return;
  }
#endif
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = wm0 + mt * 32 + (PERM ? patch_row_perm(8 * q + rh4) : 8 * q + rh4);
      const int y = y0 + (r >> 4), x = x0 + (r & 15);
      const int nvalid = y < p.g.H ? min(4, p.g.W - x) : 0;
      const long row0 = img + (long)y * p.g.W + x;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float v[4] = {acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
        conv_epilogue4<EPI, FAST>(p, row0, nvalid, cb + nt * 32 + c_lane, v);
      }
    }
}

// Epilogue of a generic implicit-GEMM wave tile: accumulator row r of the tile is token rb + r.
template <int EPI, bool FAST, int MT, int NT>
__device__ __forceinline__ void conv_epilogue_rows(const ConvGemmParams& p, const f32x16 (&acc)[MT][NT], int lane, long rb, int cb, long M) {
  const int c_lane = lane & 31, rh4 = 4 * (lane >> 5);
  if (rb + MT * 32 <= M && cb + NT * 32 <= p.cout) {          // whole wave tile real: operands of four groups loaded together (epi_load4)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        EpiOperands ops[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) epi_load4<EPI>(p, rb + mt * 32 + 8 * q + rh4, cb + nt * 32 + c_lane, ops[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v[4] = {acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
          epi_finish4<EPI, FAST>(p, rb + mt * 32 + 8 * q + rh4, cb + nt * 32 + c_lane, v, ops[q]);
        }
      }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long row0 = rb + mt * 32 + 8 * q + rh4;
      const int nvalid = (int)max(0L, min(4L, M - row0));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float v[4] = {acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
        conv_epilogue4<EPI, FAST>(p, row0, nvalid, cb + nt * 32 + c_lane, v);
      }
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
      // CRAFT_STATS_REPLICAS copies of the table, picked by block id: thousands of blocks add into the same
      // (image, channel) cell at about the same time and same-address atomics serialise in L2
      double* st = p.stats + (long)(blockIdx.x % CRAFT_STATS_REPLICAS) * (p.g.npix / (p.g.H * p.g.W)) * p.cout * 2;
      atomicAdd(&st[(b * p.cout + col) * 2], (double)s1);
      atomicAdd(&st[(b * p.cout + col) * 2 + 1], (double)s2);
    }
  }
}

}  // namespace craft
