// GEMM-engine instantiations with store-type epilogues:
//   * rows x rows GEMM (nn.Linear, V^T projection, attention apply O = P V)
//   * NHWC implicit-GEMM convolution with the update block's fused epilogues
#include "conv_epilogue.hpp"

namespace craft {

// ---------------------------------------------------------------------------------------------
// rows GEMM:  C[z][m, n] = act(scale * sum_k A[z][m,k] B[z][n,k] + bias[n])
// ---------------------------------------------------------------------------------------------
template <int PREC, int BN, bool A16>
__global__ __launch_bounds__(NTHREADS) void k_gemm_rows(RowsGemmParams p) {
  constexpr int BM = 128, WM = 2, WN = 2, MT = BM / WM / 32, NT = BN / WN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, z = blockIdx.z;
  const int z0 = z / p.zdiv, z1 = z - z0 * p.zdiv;
  f32x16 acc[MT][NT];
  acc_zero(acc);
  LoaderRowsF32<BN> lb;
  lb.init(reinterpret_cast<const float*>(p.B) + z0 * p.b_bs0 + z1 * p.b_bs1, p.ldb, n0, p.N, p.K, tid);
  const int nk = (p.K + BK - 1) / BK;
  if constexpr (A16) {
    LoaderRowsH16<BM> la;
    la.init(reinterpret_cast<const uint16_t*>(p.A) + z0 * p.a_bs0 + z1 * p.a_bs1, p.lda, m0, p.M, p.K, tid);
    gemm_mainloop<PREC, BM, BN, WM, WN>(la, lb, nk, acc, NoFold());
  } else {
    LoaderRowsF32<BM> la;
    la.init(reinterpret_cast<const float*>(p.A) + z0 * p.a_bs0 + z1 * p.a_bs1, p.lda, m0, p.M, p.K, tid);
    gemm_mainloop<PREC, BM, BN, WM, WN>(la, lb, nk, acc, NoFold());
  }
  const long cbase = z0 * p.c_bs0 + z1 * p.c_bs1;
  const int rb = m0 + (wave / WN) * (BM / WM), cb = n0 + (wave % WN) * (BN / WN);
  acc_foreach<MT, NT>(acc, lane, [&](int r, int c, float v, int, int, int) {
    const int row = rb + r, col = cb + c;
    if (row < p.M && col < p.N) {
      v *= p.scale;
      if (p.bias) v += p.bias[col];
      v = act_apply(v, p.act);
      const long o = cbase + (long)row * p.ldc + col;
      if (p.c_dtype == CRAFT_PREC_F32) reinterpret_cast<float*>(p.C)[o] = v;
      else if (p.c_dtype == CRAFT_PREC_BF16) reinterpret_cast<__bf16*>(p.C)[o] = (__bf16)v;
      else reinterpret_cast<_Float16*>(p.C)[o] = (_Float16)v;
    }
  });
}

// ---------------------------------------------------------------------------------------------
// O = P . V with BOTH operands already 16-bit in HBM (P from k_attn_probs, V^T from craft_linear_t):
// K-tile 64, pure 16-byte copies global -> registers -> LDS (no conversion), 128 x 128 tile, fp32 out.
// HBM-bound on P (streamed exactly once: the n-tile covers all of Dv = 128).
// ---------------------------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(NTHREADS) void k_pv16(RowsGemmParams p) {
  typedef typename PrecT<PREC>::lds_t lds_t;
  constexpr int BM = 128, BN = 128, KT = 64, LD = 72, WN = 2, MT = 2, NT = 2;
  constexpr int TILE = 128 * LD;
  __shared__ __attribute__((aligned(16))) lds_t S[4 * TILE];      // A0 | A1 | B0 | B1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, z = blockIdx.z;
  const int z0 = z / p.zdiv, z1 = z - z0 * p.zdiv;
  const uint16_t* A = reinterpret_cast<const uint16_t*>(p.A) + z0 * p.a_bs0 + z1 * p.a_bs1;
  const uint16_t* B = reinterpret_cast<const uint16_t*>(p.B) + z0 * p.b_bs0 + z1 * p.b_bs1;
  const int c8 = tid & 7, r0 = tid >> 3;
  const uint16_t* pa[4];
  const uint16_t* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = min(m0 + r0 + 32 * i, p.M - 1), rb_ = min(n0 + r0 + 32 * i, p.N - 1);   // clamped: unconditional loads
    pa[i] = A + (long)ra * p.lda;
    pb[i] = B + (long)rb_ * p.ldb;
  }
  const int nk = (p.K + KT - 1) / KT;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 va[4], vb[4];
  bool kzero = false;
  auto fetch = [&](int kt) __attribute__((always_inline)) {
    const int k = kt * KT + c8 * 8;
    const bool kok = k < p.K;
    const int kc = kok ? k : p.K - 8;
    kzero = !kok;                 // applied in store(): a select here would stall on the loads
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      va[i] = *reinterpret_cast<const u32x4*>(pa[i] + kc);
      vb[i] = *reinterpret_cast<const u32x4*>(pb[i] + kc);
    }
  };
  auto store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(&S[buf * TILE + (r0 + 32 * i) * LD + c8 * 8]) = kzero ? u32x4{0u, 0u, 0u, 0u} : va[i];
      *reinterpret_cast<u32x4*>(&S[(2 + buf) * TILE + (r0 + 32 * i) * LD + c8 * 8]) = kzero ? u32x4{0u, 0u, 0u, 0u} : vb[i];
    }
  };
  const int wm0 = (wave / WN) * 64, wn0 = (wave % WN) * 64;
  const int r = lane & 31, g = lane >> 5;
  f32x16 acc[MT][NT];
  acc_zero(acc);
  fetch(0);
  store(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    fetch(min(kt + 1, nk - 1));          // always (clamped duplicate at the end): no branch around the loads
    __builtin_amdgcn_sched_barrier(0);   // keep the loads ABOVE the MFMAs (hipcc otherwise sinks them and waits at once)
    const lds_t* As = &S[cur * TILE];
    const lds_t* Bs = &S[(2 + cur) * TILE];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if constexpr (PREC == CRAFT_PREC_BF16) {
        bf16x8 a[MT], b[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(&As[(wm0 + mt * 32 + r) * LD + kk * 16 + g * 8]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const bf16x8*>(&Bs[(wn0 + nt * 32 + r) * LD + kk * 16 + g * 8]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
      } else {
        f16x8 a[MT], b[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const f16x8*>(&As[(wm0 + mt * 32 + r) * LD + kk * 16 + g * 8]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const f16x8*>(&Bs[(wn0 + nt * 32 + r) * LD + kk * 16 + g * 8]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    store(cur ^ 1);
    __syncthreads();
  }
  float* C = reinterpret_cast<float*>(p.C) + z0 * p.c_bs0 + z1 * p.c_bs1;
  acc_foreach<MT, NT>(acc, lane, [&](int rr, int cc, float v, int, int, int) {
    const int row = m0 + wm0 + rr, col = n0 + wn0 + cc;
    if (row < p.M && col < p.N) C[(long)row * p.ldc + col] = v;
  });
}

int launch_pv16(const RowsGemmParams& p, int prec, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0 || p.batch <= 0) return 0;
  if ((p.K & 7) || (p.lda & 7) || (p.ldb & 7) || p.c_dtype != CRAFT_PREC_F32) return CRAFT_ERR_ALIGN;
  dim3 grid((p.M + 127) / 128, (p.N + 127) / 128, p.batch);
  if (prec == CRAFT_PREC_BF16) hipLaunchKernelGGL((k_pv16<CRAFT_PREC_BF16>), grid, dim3(NTHREADS), 0, s, p);
  else if (prec == CRAFT_PREC_F16) hipLaunchKernelGGL((k_pv16<CRAFT_PREC_F16>), grid, dim3(NTHREADS), 0, s, p);
  else return CRAFT_ERR_ARG;
  return (int)hipGetLastError();
}

template <int PREC, int BN, bool A16> static int launch_rows_t(const RowsGemmParams& p, hipStream_t s) {
  dim3 grid((p.M + 127) / 128, (p.N + BN - 1) / BN, p.batch);
  hipLaunchKernelGGL((k_gemm_rows<PREC, BN, A16>), grid, dim3(NTHREADS), 0, s, p);
  return (int)hipGetLastError();
}

static int pick_bn(int N) {
  if (N % 128 == 0) return 128;
  if (N % 64 == 0) return 64;
  return N > 64 ? 128 : 64;
}

int launch_gemm_rows(const RowsGemmParams& p, int prec, bool a16, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0 || p.batch <= 0) return 0;
  if ((p.K & 3) || (p.lda & 3) || (p.ldb & 3)) return CRAFT_ERR_ALIGN;
  if (a16 && ((p.K & 7) || (p.lda & 7) || (prec != CRAFT_PREC_BF16 && prec != CRAFT_PREC_F16))) return CRAFT_ERR_ALIGN;
  const int bn = pick_bn(p.N);
#define GO(PR, BNV, A) return launch_rows_t<PR, BNV, A>(p, s)
  if (prec == CRAFT_PREC_F32) { if (bn == 128) GO(CRAFT_PREC_F32, 128, false); else GO(CRAFT_PREC_F32, 64, false); }
  if (prec == CRAFT_PREC_BF16) {
    if (a16) { if (bn == 128) GO(CRAFT_PREC_BF16, 128, true); else GO(CRAFT_PREC_BF16, 64, true); }
    if (bn == 128) GO(CRAFT_PREC_BF16, 128, false); else GO(CRAFT_PREC_BF16, 64, false);
  }
  if (prec == CRAFT_PREC_F16) {
    if (a16) { if (bn == 128) GO(CRAFT_PREC_F16, 128, true); else GO(CRAFT_PREC_F16, 64, true); }
    if (bn == 128) GO(CRAFT_PREC_F16, 128, false); else GO(CRAFT_PREC_F16, 64, false);
  }
  if (prec == CRAFT_PREC_F16X3) { if (bn == 128) GO(CRAFT_PREC_F16X3, 128, false); else GO(CRAFT_PREC_F16X3, 64, false); }
#undef GO
  return CRAFT_ERR_ARG;
}

// ---------------------------------------------------------------------------------------------
// NHWC implicit-GEMM conv: out[pix, co] = epi( sum_{tap,c} in[pix+tap, c] * W[co, tap, c] + bias[co] )
// ---------------------------------------------------------------------------------------------
template <int PREC, int BN, bool ENC>
__global__ __launch_bounds__(NTHREADS) void k_gemm_conv(ConvGemmParams p) {
  constexpr int BM = 128, WM = 2, WN = 2, MT = BM / WM / 32, NT = BN / WN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  f32x16 acc[MT][NT];
  acc_zero(acc);
  const int K = p.g.KH * p.g.KW * (p.g.c0 + p.g.c1);
  LoaderConvF32<BM> la;
  la.init(p.g, m0, tid);
  LoaderRowsF32<BN> lb;
  lb.init(p.W, K, n0, p.cout, K, tid);
  gemm_mainloop<PREC, BM, BN, WM, WN>(la, lb, K / BK, acc, NoFold());
  const int rb = m0 + (wave / WN) * (BM / WM), cb = n0 + (wave % WN) * (BN / WN);
  const int M = p.g.npix;
#define BODY(E) conv_epilogue_rows<E, PREC != CRAFT_PREC_F32, MT, NT>(p, acc, lane, (long)rb, cb, (long)M);
  CONV_EPI_DISPATCH(p, BODY)
#undef BODY
  if (ENC && p.stats) {
    // all rows of a tile must belong to one image (checked by the launcher: H*W % 128 == 0)
    const int hw = p.g.H * p.g.W;
    unsigned mlo = 0u, mhi = 0u;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = rb + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int bit = mt * 16 + e;
        if (r < M) { if (bit < 32) mlo |= 1u << bit; else mhi |= 1u << (bit - 32); }
      }
    conv_col_stats<MT, NT>(p, acc, lane, cb, (long)(m0 / hw), mlo, mhi);
  }
}

template <int PREC, int BN> static int launch_conv_t(const ConvGemmParams& p, hipStream_t s) {
  const int ncols = p.epi == CONV_EPI_MENC ? p.cout + 2 : p.cout;
  dim3 grid((p.g.npix + 127) / 128, (ncols + BN - 1) / BN, 1);
  if (p.stats) hipLaunchKernelGGL((k_gemm_conv<PREC, BN, true>), grid, dim3(NTHREADS), 0, s, p);
  else hipLaunchKernelGGL((k_gemm_conv<PREC, BN, false>), grid, dim3(NTHREADS), 0, s, p);
  return (int)hipGetLastError();
}

int launch_gemm_conv(const ConvGemmParams& p, int prec, hipStream_t s) {
  if (p.g.npix <= 0) return 0;
  if ((p.g.c0 % 32) || (p.g.c1 % 32) || (p.g.ld0 & 3) || (p.g.c1 && (p.g.ld1 & 3))) return CRAFT_ERR_ALIGN;
  if (p.g.KH * p.g.KW > 1 && !p.force_generic && p.g.stride == 1) {
    const int rc = launch_conv_halo(p, prec, s);
    if (rc != CRAFT_ERR_UNSUPPORTED) return rc;
  }
  if (p.w_packed) return CRAFT_ERR_UNSUPPORTED;     // the generic implicit GEMM reads raw fp32 weights
  if (p.g.in_norm) return CRAFT_ERR_UNSUPPORTED;    // lazy input normalisation is a k_conv_halo feature
  if (p.stats && (p.g.H * p.g.W) % 128) return CRAFT_ERR_UNSUPPORTED;
  const int ncols = p.epi == CONV_EPI_MENC ? p.cout + 2 : p.cout;
  const int bn = pick_bn(ncols);
#define GO(PR) do { if (bn == 128) return launch_conv_t<PR, 128>(p, s); else return launch_conv_t<PR, 64>(p, s); } while (0)
  if (prec == CRAFT_PREC_F32) GO(CRAFT_PREC_F32);
  if (prec == CRAFT_PREC_BF16) GO(CRAFT_PREC_BF16);
  if (prec == CRAFT_PREC_F16) GO(CRAFT_PREC_F16);
  if (prec == CRAFT_PREC_F16X3) GO(CRAFT_PREC_F16X3);
#undef GO
  return CRAFT_ERR_ARG;
}

}  // namespace craft
