// GEMM-engine instantiations with store-type epilogues:
//   * rows x rows GEMM (nn.Linear, V^T projection, attention apply O = P V)
//   * NHWC implicit-GEMM convolution with the update block's fused epilogues
#include <cstdlib>
#include <type_traits>
#include "conv_epilogue.hpp"

namespace craft {

// Store epilogue of a rows-GEMM wave tile (acc[MT][NT], rows rb.., columns cb..) -- shared by k_gemm_rows and k_gemm_rows_wf.
template <int MT, int NT>
__device__ __forceinline__ void rows_epilogue(const RowsGemmParams& p, const f32x16 (&acc)[MT][NT], int rb, int cb, int z, int z0, int z1, int lane) {
  const long cbase = z0 * p.c_bs0 + z1 * p.c_bs1;
  // Epilogue in quads of 4 consecutive rows (the accumulator layout).  Everything that does not depend on the element is decided ONCE
  // outside the loops -- element type / layout, the deferred row divisor, and whether the wave's tile lies inside the matrix (no bounds
  // tests then): with the per-element forms (`p.row_div ? .. : ..`, `row0 + i < M`) hipcc branched around loads 64 times per lane and
  // the K = 128 .. 352 products of the refinement loop (V^T projection, convc1) spent a quarter of their 31 us there (craft_gemm's
  // kernel, same main loop, hoisted stores: 24 us -- tools/bench_1x1.py).  act is NONE or RELU here (checked by the launcher).
  const bool relu = p.act == CRAFT_ACT_RELU;
  const int c_lane = lane & 31, rh4 = 4 * (lane >> 5);
  const float scale = p.scale;
  const int M = p.M, N = p.N;
  const bool inside = rb + MT * 32 <= M && cb + NT * 32 <= N;
  const float* rdiv = p.row_div ? p.row_div + (long)z * p.rd_bs : nullptr;
  auto run = [&](auto dt_tag, auto frag_tag, auto rd_tag, auto in_tag) __attribute__((always_inline)) {
    constexpr int DT = decltype(dt_tag)::value;
    constexpr bool FRAG = decltype(frag_tag)::value, RD = decltype(rd_tag)::value, INB = decltype(in_tag)::value;
    typedef typename std::conditional<DT == CRAFT_PREC_F32, float, typename std::conditional<DT == CRAFT_PREC_BF16, __bf16, _Float16>::type>::type out_t;
    out_t* C = reinterpret_cast<out_t*>(p.C) + cbase;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = cb + nt * 32 + c_lane;
      if (!INB && col >= N) continue;
      const float bias = p.bias ? p.bias[col] : 0.f;
      long fbase = 0;
      if constexpr (FRAG) {      // MFMA B-fragment order per group of c_frag columns (k_pv16's V^T operand): row = key, col = V^T row
        const int grp = col / p.c_frag, n = col - grp * p.c_frag;
        fbase = (long)grp * p.c_frag * p.ldc + (long)(n >> 5) * 512 + (n & 31) * 8;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row0 = rb + mt * 32 + 8 * q + rh4;
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float t = acc[mt][nt][4 * q + i] * scale;
            if constexpr (RD) t /= rdiv[min(row0 + i, M - 1)];
            t += bias;
            v[i] = relu ? fmaxf(t, 0.f) : t;
          }
          if constexpr (FRAG) {
            // keys row0..row0+3 share (key >> 3): 4 consecutive 16-bit values, one 8-byte store (lanes l, l+32 pair up)
            // (accumulator key order: half = bit 2 of the key, position within the lane = 4 * bit 3 + low two bits)
            out_t* d = C + fbase + (long)(row0 >> 4) * (p.c_frag >> 5) * 512 +
                       (p.c_frag_acc ? ((row0 >> 2) & 1) * 256 + ((row0 >> 3) & 1) * 4 : ((row0 >> 3) & 1) * 256 + (row0 & 7));
            if (INB || row0 + 3 < M) {
              typedef out_t o4 __attribute__((ext_vector_type(4)));
              o4 h;
              h[0] = (out_t)v[0]; h[1] = (out_t)v[1]; h[2] = (out_t)v[2]; h[3] = (out_t)v[3];
              *reinterpret_cast<o4*>(d) = h;
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i) if (row0 + i < M) d[i] = (out_t)v[i];
            }
          } else {
            out_t* d = C + (long)row0 * p.ldc + col;
#pragma unroll
            for (int i = 0; i < 4; ++i) if (INB || row0 + i < M) d[(long)i * p.ldc] = (out_t)v[i];
          }
        }
    }
  };
  typedef std::integral_constant<int, CRAFT_PREC_F32> T32;
  typedef std::integral_constant<int, CRAFT_PREC_BF16> TBF;
  typedef std::integral_constant<int, CRAFT_PREC_F16> TF16;
  auto pick_in = [&](auto dt_tag, auto frag_tag, auto rd_tag) __attribute__((always_inline)) {
    if (inside) run(dt_tag, frag_tag, rd_tag, std::true_type()); else run(dt_tag, frag_tag, rd_tag, std::false_type());
  };
  auto pick_rd = [&](auto dt_tag, auto frag_tag) __attribute__((always_inline)) {
    if (rdiv) pick_in(dt_tag, frag_tag, std::true_type()); else pick_in(dt_tag, frag_tag, std::false_type());
  };
  if (p.c_dtype == CRAFT_PREC_F32) pick_rd(T32(), std::false_type());
  else if (p.c_dtype == CRAFT_PREC_BF16) { if (p.c_frag) pick_rd(TBF(), std::true_type()); else pick_rd(TBF(), std::false_type()); }
  else { if (p.c_frag) pick_rd(TF16(), std::true_type()); else pick_rd(TF16(), std::false_type()); }
}

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
  const int rb = m0 + (wave / WN) * (BM / WM), cb = n0 + (wave % WN) * (BN / WN);
  rows_epilogue<MT, NT>(p, acc, rb, cb, z, z0, z1, lane);
}

// ---------------------------------------------------------------------------------------------
// k_gemm_rows_wf (round 5): the rows GEMM with the WEIGHT operand pre-packed in MFMA fragment order (craft_pack_weights: [K/32][N/32]
// [plane][k-half][lane][8], K padded to a multiple of 32 with zeros) -- nn.Linear / 1x1-convolution products with short K (128 .. 352:
// the V^T projection of the motion aggregator, convc1 of the motion encoder, the q / k projections).  k_gemm_rows stages BOTH operands
// through LDS per 32-wide K-tile (fp32 -> planes conversion of the weights in every block, 24 MFMAs per wave between barriers: PMC
// round 5: matrix pipe 10 - 31 % busy, 12 VALU per MFMA).  Here:
//   * weights: one coalesced 1 KiB load per MFMA operand from L2 straight into registers (the k_conv_halo_wf idiom), requested a
//     K-chunk ahead; never converted, never in LDS;
//   * activations: fp32 rows -> hi / lo fp16 planes in LDS per 64-wide K chunk (double buffer, ONE barrier per chunk = 48 MFMAs per
//     wave), the next chunk's rows requested before this chunk's MFMAs;
//   * a wave owns all 128 rows x 32 columns (MT = 4): 12 MFMAs per 8 LDS reads + 2 weight loads per k-half.
// Epilogue: rows_epilogue (row-major fp32 / 16-bit, MFMA fragment order for V^T, bias, ReLU, deferred row divisor).
// ---------------------------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(NTHREADS) void k_gemm_rows_wf(RowsGemmParams p) {
  typedef typename PrecT<PREC>::lds_t lds_t;
  typedef typename FragT<PREC>::t frag_t;
  constexpr int PL = Planes<PREC>::N;
  constexpr int BM = 128, MT = 4, BN = 128, KC = 64, LD = KC + 8;
  constexpr int NP = BM / 16;                           // float4 per thread and chunk: thread -> (row r0 + 16 i, floats c4*4 .. +3)
  constexpr int A_ELEMS = PL * BM * LD;
  __shared__ __attribute__((aligned(16))) lds_t As[2 * A_ELEMS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, z = blockIdx.z;
  const int z0 = z / p.zdiv, z1 = z - z0 * p.zdiv;
  const int K = p.K, nkt = (K + 31) / 32, nch = (nkt + 1) / 2;
  const int c4 = tid & 15, r0 = tid >> 4;
  const float* arow[NP];
  {
    const float* A = reinterpret_cast<const float*>(p.A) + z0 * p.a_bs0 + z1 * p.a_bs1;
#pragma unroll
    for (int i = 0; i < NP; ++i) arow[i] = A + (long)min(m0 + r0 + 16 * i, p.M - 1) * p.lda;      // clamped: unconditional loads
  }
  float4 ra[NP];
  auto fetch_a = [&](int ch, bool& zero) __attribute__((always_inline)) {
    const int k = ch * KC + c4 * 4;
    zero = k >= K;                                     // (K % 4 == 0: a float4 is in or out as a whole)
    const int kc = zero ? K - 4 : k;
#pragma unroll
    for (int i = 0; i < NP; ++i) ra[i] = *reinterpret_cast<const float4*>(arow[i] + kc);
  };
  auto store_a = [&](int buf, bool zero) __attribute__((always_inline)) {
    lds_t* A0 = &As[buf * A_ELEMS];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int row = r0 + 16 * i;
      float4 v = ra[i];
      v.x = zero ? 0.f : v.x; v.y = zero ? 0.f : v.y; v.z = zero ? 0.f : v.z; v.w = zero ? 0.f : v.w;
      if constexpr (PREC == CRAFT_PREC_F16X3) {
        f16x4 h, l;
        split_f16x3(v, h, l);
        *reinterpret_cast<f16x4*>(&A0[row * LD + c4 * 4]) = h;
        *reinterpret_cast<f16x4*>(&A0[(BM + row) * LD + c4 * 4]) = l;
      } else if constexpr (PREC == CRAFT_PREC_BF16) {
        bf16x4 h;
        h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4*>(&A0[row * LD + c4 * 4]) = h;
      } else {
        f16x4 h;
        h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
        *reinterpret_cast<f16x4*>(&A0[row * LD + c4 * 4]) = h;
      }
    }
  };
  // weight fragments of this wave's 32 columns: [kt][nb][pl][kk][lane][8]; a column block beyond N re-reads the last one (discarded)
  const int NBtot = (p.N + 31) / 32;
  const int nb = min(n0 / 32 + wave, NBtot - 1);
  const uint16_t* wb = reinterpret_cast<const uint16_t*>(p.B) + (z0 * p.b_bs0 + z1 * p.b_bs1) + (long)nb * (PL * 1024) + lane * 8;
  const long kt_stride = (long)NBtot * (PL * 1024);
  frag_t bq[PL][4];                                    // the four k-halves of one chunk
  auto fetch_b = [&](int ch) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int kt = min(2 * ch + (h >> 1), nkt - 1);   // (a missing second K-tile of the last chunk meets zeroed activations)
      const uint16_t* q = wb + kt * kt_stride + (h & 1) * 512;
#pragma unroll
      for (int pl = 0; pl < PL; ++pl) bq[pl][h] = *reinterpret_cast<const frag_t*>(q + pl * 1024);
    }
  };
  f32x16 acc[MT][1];
  acc_zero(acc);
  const int r = lane & 31, g8 = (lane >> 5) * 8;
  bool zero;
  fetch_a(0, zero);
  fetch_b(0);
  store_a(0, zero);
  if (nch > 1) fetch_a(1, zero);
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    const lds_t* A0 = &As[(ch & 1) * A_ELEMS];
    frag_t bc[PL][4];
#pragma unroll
    for (int pl = 0; pl < PL; ++pl)
#pragma unroll
      for (int h = 0; h < 4; ++h) bc[pl][h] = bq[pl][h];
    if (ch + 1 < nch) fetch_b(ch + 1);                  // (lands behind this chunk's 48 MFMAs)
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      frag_t ah[MT], al[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        ah[mt] = *reinterpret_cast<const frag_t*>(&A0[(mt * 32 + r) * LD + h * 16 + g8]);
        if constexpr (PL == 2) al[mt] = *reinterpret_cast<const frag_t*>(&A0[(BM + mt * 32 + r) * LD + h * 16 + g8]);
      }
      if constexpr (PL == 2) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][0] = mfma16<PREC>(al[mt], bc[0][h], acc[mt][0]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][0] = mfma16<PREC>(ah[mt], bc[1][h], acc[mt][0]);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][0] = mfma16<PREC>(ah[mt], bc[0][h], acc[mt][0]);
      if (h == 1 && ch + 1 < nch) {                     // publish the next chunk into the idle buffer, request the one after it
        store_a((ch + 1) & 1, zero);
        if (ch + 2 < nch) fetch_a(ch + 2, zero);
      }
    }
    __syncthreads();
  }
  rows_epilogue<MT, 1>(p, acc, m0, n0 + wave * 32, z, z0, z1, lane);
}

static int launch_rows_wf(const RowsGemmParams& p, int prec, hipStream_t s) {
  dim3 grid((p.M + 127) / 128, (p.N + 127) / 128, p.batch);
  if (prec == CRAFT_PREC_F16X3) hipLaunchKernelGGL((k_gemm_rows_wf<CRAFT_PREC_F16X3>), grid, dim3(NTHREADS), 0, s, p);
  else if (prec == CRAFT_PREC_F16) hipLaunchKernelGGL((k_gemm_rows_wf<CRAFT_PREC_F16>), grid, dim3(NTHREADS), 0, s, p);
  else if (prec == CRAFT_PREC_BF16) hipLaunchKernelGGL((k_gemm_rows_wf<CRAFT_PREC_BF16>), grid, dim3(NTHREADS), 0, s, p);
  else return CRAFT_ERR_UNSUPPORTED;
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// O = P . V with BOTH operands already 16-bit in HBM: P [rows][K] from k_attn_probs, V^T in MFMA FRAGMENT order
// from craft_linear_t (frag_rows = Dv): vT[((g * NB + nb) * 64 + lane) * 8 + j] = V^T[nb*32 + (lane & 31)][g*16 +
// (lane >> 5)*8 + j], NB = Dv / 32, g = key / 16.
//
// The kernel is HBM-bound on P (streamed exactly once) and nothing but its own pipeline hides the HBM latency, so
// the two things that matter are bytes in flight and an even block count:
//   * a block owns BM = 32*MT rows of P and 128 columns of O; a wave owns all BM rows x 32 columns.  MT (4..7) is
//     picked per launch so that the grid is a whole number of resident rounds (448x1024: N = 7168 = 32 * 224 -> MT = 7
//     gives 512 blocks = exactly 2 per CU; with MT = 4 the 896 blocks run as 1.75 rounds and the tail costs 12 %);
//   * P: 128 B per row and K-tile (64 keys), global -> registers TWO tiles ahead (two register sets) -> LDS (double
//     buffer), pure 16-byte copies;
//   * V^T never touches LDS: each B operand is one coalesced 1 KiB load (L2 hit) straight into MFMA registers,
//     re-requested for the next tile right after its last use; one barrier per K-tile.
// ---------------------------------------------------------------------------------------------
#ifndef CRAFT_PV_ABL
#define CRAFT_PV_ABL 0      // developer ablation (tools/build_variant.py -DCRAFT_PV_ABL=n): 1 no MFMAs (| 8: s_sleep for their issue time instead), 2 no A-fragment LDS reads, 4 V^T fragments loaded once
#endif
// WR = 2 (round 6): an 8-wave block of 2 x 32 MT rows -- waves (wr, wc) = (wave >> 2, wave & 3) own row half wr x column group wc.  One
// block per CU (129 KB of LDS at MT = 7) holds the same 8 waves as two 4-wave blocks, but a V^T fragment is fetched once per 64 MT rows
// of P instead of once per 32 MT (VERDICT r5 "next" 7: the per-block V^T traffic), and 448x1024 x batch 4 x 4 modes is 256 blocks =
// exactly one per CU.  Per-thread work is unchanged: thread t moves chunk (t & 255) of the 4 KiB band tiles of its row half.
template <int PREC, int MT, int WR>
__global__ __launch_bounds__(NTHREADS * WR) void k_pv16(RowsGemmParams p) {
  typedef typename PrecT<PREC>::lds_t lds_t;
  constexpr int BMH = 32 * MT, BM = BMH * WR, KT = 64, LD = KT + 8;
  constexpr int TILE = BM * LD;
  __shared__ __attribute__((aligned(16))) lds_t S[2 * TILE];      // A0 | A1
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, wr = tid >> 8, t256 = tid & 255;
  // XCD-aware block map.  Blocks are dealt to the 8 XCDs round-robin (block b -> XCD b % 8), and every block of one batch entry z
  // re-reads that entry's V^T (N x Dv x 2 B: 1.8 MB at 448x1024) from ITS XCD's L2.  With z outermost in the grid all 8 L2s
  // fetched all batch x modes copies (8 x 29 MB of a 1.97 GB launch, 13 % over the algorithmic bytes: PMC FETCH_SIZE); here XCD x
  // owns a contiguous eighth of the (z, row block) list, i.e. two whole entries at batch 4 x 4 modes, whose V^T stay L2-resident.
  const int gx = (p.M + BM - 1) / BM, gy = p.N / 128;
#ifdef CRAFT_PV_NO_XCD_MAP                       // developer A/B (tools/build_variant.py): z outermost, round-robin over the XCDs
  const int lin = blockIdx.x;
#else
  const int lin = xcd_chunk(blockIdx.x, gridDim.x);
#endif
  const int bx = lin % gx, byz = lin / gx;
  const int m0 = bx * BM, n0 = (byz % gy) * 128, z = byz / gy;
  const int z0 = z / p.zdiv, z1 = z - z0 * p.zdiv;
  const uint16_t* A = reinterpret_cast<const uint16_t*>(p.A) + z0 * p.a_bs0 + z1 * p.a_bs1;
  const int NB = p.N / 32, ng = (int)(p.ldb / 16);
  const uint16_t* Bf = reinterpret_cast<const uint16_t*>(p.B) + z0 * p.b_bs0 + z1 * p.b_bs1 + (long)(n0 / 32 + wave) * 512 + lane * 8;
  const long g_stride = (long)NB * 512;
  const int c8 = t256 & 7, r0 = wr * BMH + (t256 >> 3);
  const uint16_t* pa[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)                                                              // clamped: unconditional loads
    pa[i] = p.a_tiled ? A + (long)min((bx * WR + wr) * MT + i, (p.M - 1) >> 5) * 32 * p.lda + t256 * 8 : A + (long)min(m0 + r0 + 32 * i, p.M - 1) * p.lda;
  const int nk = (p.K + KT - 1) / KT;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef typename std::conditional<PREC == CRAFT_PREC_BF16, bf16x8, f16x8>::type frag_t;
  u32x4 va0[MT], va1[MT];
  frag_t bq[4];
  auto fetch_a = [&](int kt, u32x4 (&va)[MT]) __attribute__((always_inline)) {
    // tiled P: the block's 32 rows x 64 keys of a band are one contiguous 4 KiB (DRAM pages stay open; row-major P is 32 x 128 B
    // segments 2 * ldp bytes apart, one DRAM row activation each)
    const int k = min(kt, nk - 1) * KT + c8 * 8;
    const long kc = p.a_tiled ? (long)min(kt, nk - 1) * (32 * KT) : (long)(k < p.K ? k : p.K - 8);
#pragma unroll
    // non-temporal: P (1.6 GB per launch at 448x1024 x 4) is streamed once per launch and must not displace V^T / O lines; on the tiled
    // layout the same read-only stream measures 6.9 TB/s with nt against 6.1 without, LDS-DMA or not (tools/ubench/hbm_rows_dma.hip,
    // profiles/r5/hbm_rows_dma.txt -- VERDICT r4 #6's experiment: the gain is the cache policy, not the DMA)
#ifdef CRAFT_PV_NO_NT
    for (int i = 0; i < MT; ++i) va[i] = *reinterpret_cast<const u32x4*>(pa[i] + kc);
#else
    for (int i = 0; i < MT; ++i) va[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(pa[i] + kc));
#endif
  };
  auto fetch_b = [&](int kt, int kk) __attribute__((always_inline)) {
    const int g = min(min(kt, nk - 1) * 4 + kk, ng - 1);     // groups beyond K meet zeroed P columns
    bq[kk] = *reinterpret_cast<const frag_t*>(Bf + g * g_stride);
  };
  // zeroing of the K tail happens at LDS-store time with a bitwise mask (no select next to the load, no exec branch)
  auto store = [&](int kt, const u32x4 (&va)[MT]) __attribute__((always_inline)) {
    const unsigned keep = (kt * KT + c8 * 8 >= p.K) ? 0u : ~0u;
    lds_t* D = &S[(kt & 1) * TILE];
#pragma unroll
    for (int i = 0; i < MT; ++i) *reinterpret_cast<u32x4*>(&D[(r0 + 32 * i) * LD + c8 * 8]) = va[i] & keep;
  };
  const int r = lane & 31, g8 = (lane >> 5) * 8;
  f32x16 acc[MT][1];
  acc_zero(acc);
  // one pipeline step for tile kt: request A of tile kt+2 into `vfar`; MFMAs of tile kt (each B operand re-requested
  // for tile kt+1 after its use); publish tile kt+1 (A already in `vnear`) to the other LDS buffer.
  auto step = [&](int kt, u32x4 (&vnear)[MT], u32x4 (&vfar)[MT]) __attribute__((always_inline)) {
    fetch_a(kt + 2, vfar);
    __builtin_amdgcn_sched_barrier(0);       // keep the P loads above the MFMAs they hide behind
    const lds_t* As = &S[(kt & 1) * TILE];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      frag_t a[MT];
#if !(CRAFT_PV_ABL & 2)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const frag_t*>(&As[(wr * BMH + mt * 32 + r) * LD + kk * 16 + g8]);
#else
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = bq[(kk + mt) & 3];
#endif
#if !(CRAFT_PV_ABL & 1)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if constexpr (PREC == CRAFT_PREC_BF16) acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], bq[kk], acc[mt][0], 0, 0, 0);
        else acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], bq[kk], acc[mt][0], 0, 0, 0);
      }
#else
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) asm volatile("" :: "v"(a[mt]), "v"(bq[kk]));
#if CRAFT_PV_ABL & 8                       // ... and the wave idles for about the MFMAs' issue time instead (7 x 32 cycles per k-step)
      __builtin_amdgcn_s_sleep(3);
      __builtin_amdgcn_s_sleep(1);
#endif
#endif
#if !(CRAFT_PV_ABL & 4)
      fetch_b(kt + 1, kk);
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    store(kt + 1, vnear);
    __syncthreads();
  };
  fetch_a(0, va0);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fetch_b(0, kk);
  store(0, va0);
  fetch_a(1, va1);
  __syncthreads();
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    step(kt, va1, va0);          // tile kt+1 is in va1; tile kt+2 goes to va0
    step(kt + 1, va0, va1);
  }
  if (kt < nk) step(kt, va1, va0);
  float* C = reinterpret_cast<float*>(p.C) + z0 * p.c_bs0 + z1 * p.c_bs1;
  const int col = n0 + wave * 32 + r;
  const int rh4 = 4 * (lane >> 5);
  const float* rdiv = p.row_div ? p.row_div + (long)z * p.rd_bs : nullptr;     // deferred softmax normalisation
  // (the 16 row sums of a fragment are requested together through clamped indices: one load + wait per element under the row
  // guard serialised 112 round trips at the end of every block)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float rd[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) rd[e] = rdiv ? rdiv[min(m0 + wr * BMH + mt * 32 + (e & 3) + 8 * (e >> 2) + rh4, p.M - 1)] : 1.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + wr * BMH + mt * 32 + (e & 3) + 8 * (e >> 2) + rh4;
      float v = acc[mt][0][e];
      if (rdiv) v *= __builtin_amdgcn_rcpf(rd[e]);      // (16-bit P: a 1-ulp reciprocal is far below its rounding)
      if (row < p.M) C[(long)row * p.ldc + col] = v;
    }
  }
}

template <int PREC, int MT, int WR = 1> static void launch_pv_t(const RowsGemmParams& p, hipStream_t s) {
  dim3 grid((unsigned)((long)((p.M + 32 * MT * WR - 1) / (32 * MT * WR)) * (p.N / 128) * p.batch), 1, 1);
  hipLaunchKernelGGL((k_pv16<PREC, MT, WR>), grid, dim3(NTHREADS * WR), 0, s, p);
}

int launch_pv16(const RowsGemmParams& p, int prec, int rows32, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0 || p.batch <= 0) return 0;
  if ((p.K & 15) || (p.lda & 7) || (p.ldb & 15) || (p.N & 127) || p.c_dtype != CRAFT_PREC_F32) return CRAFT_ERR_ALIGN;
  if (p.a_tiled && ((p.K & 63) || p.lda != p.K)) return CRAFT_ERR_ALIGN;
  // rows per block: minimise (resident rounds) x (rows a CU works through per round).  Blocks per CU from the register / LDS budget of
  // each instantiation (4 waves: MT = 4: 3, MT >= 5: 2; 8 waves (WR = 2): one).  rows32 forces an instantiation (tests / A-B runs):
  // 4..7 = MT of the 4-wave kernel, 8 / 10 / 12 / 14 = the 8-wave kernel with MT = rows32 / 2.
  if (rows32 && !((rows32 >= 4 && rows32 <= 7) || (rows32 >= 8 && rows32 <= 14 && rows32 % 2 == 0))) return CRAFT_ERR_ARG;
  int best = 4, best_wr = 1; long best_cost = -1;
  for (int mt = 4; mt <= 7; ++mt) {                       // the 4-wave kernel: as in rounds 3-5
    const long blocks = (long)((p.M + 32 * mt - 1) / (32 * mt)) * (p.N / 128) * p.batch;
    const long slots = 256L * (mt == 4 ? 3 : 2);
    const long cost = ((blocks + slots - 1) / slots) * mt;
    if (best_cost < 0 || cost < best_cost) { best = mt; best_cost = cost; }
  }
  if (tuning().pv_wr2) {
    // the 8-wave kernel takes over when it fills every CU and a CU works through no more rows than with the pick above (a CU holds
    // one 8-wave block or `per` 4-wave blocks side by side; ties go to the 8-wave kernel: same rounds, half the V^T fetches)
    const long per = best == 4 ? 3 : 2;
    const long rows_cu1 = best_cost * per;               // rounds x 32-row groups per block x resident blocks
    int best2 = 0; long rows_cu2 = -1;
    for (int mt = 4; mt <= 7; ++mt) {
      const long blocks = (long)((p.M + 64 * mt - 1) / (64 * mt)) * (p.N / 128) * p.batch;
      if (blocks < 256) continue;
      const long c = ((blocks + 255) / 256) * 2 * mt;
      if (rows_cu2 < 0 || c < rows_cu2) { best2 = mt; rows_cu2 = c; }
    }
    if (best2 && rows_cu2 <= rows_cu1) { best = best2; best_wr = 2; }
  }
  if (rows32 >= 8) { best = rows32 / 2; best_wr = 2; }
  else if (rows32) { best = rows32; best_wr = 1; }
#define GO(PR) do { if (best_wr == 2) { switch (best) { case 4: launch_pv_t<PR, 4, 2>(p, s); break; case 5: launch_pv_t<PR, 5, 2>(p, s); break; \
                                                        case 6: launch_pv_t<PR, 6, 2>(p, s); break; default: launch_pv_t<PR, 7, 2>(p, s); break; } } \
                    else { switch (best) { case 4: launch_pv_t<PR, 4>(p, s); break; case 5: launch_pv_t<PR, 5>(p, s); break; \
                                           case 6: launch_pv_t<PR, 6>(p, s); break; default: launch_pv_t<PR, 7>(p, s); break; } } } while (0)
  if (prec == CRAFT_PREC_BF16) GO(CRAFT_PREC_BF16);
  else if (prec == CRAFT_PREC_F16) GO(CRAFT_PREC_F16);
  else return CRAFT_ERR_ARG;
#undef GO
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
  if (p.act != CRAFT_ACT_NONE && p.act != CRAFT_ACT_RELU) return CRAFT_ERR_UNSUPPORTED;
  if (p.c_frag && (p.c_dtype == CRAFT_PREC_F32 || p.c_frag % 32 || p.N % p.c_frag || p.ldc % 16)) return CRAFT_ERR_ALIGN;
  if (p.b_packed) {          // weights from craft_pack_weights (K padded to 32): k_gemm_rows_wf
    if (a16 || prec == CRAFT_PREC_F32 || (p.K & 3) || (p.lda & 3) || p.zdiv != 1 || p.b_bs0 || p.b_bs1) return CRAFT_ERR_UNSUPPORTED;
    return launch_rows_wf(p, prec, s);
  }
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
    // a 128-row tile may straddle images (H*W not a multiple of 128): one masked pass per image it touches
    const int hw = p.g.H * p.g.W;
    const int b_first = m0 / hw, b_last = min(m0 + BM - 1, M - 1) / hw;
    for (int b = b_first; b <= b_last; ++b) {
      const long lo = (long)b * hw, hi = min(lo + hw, (long)M);
      unsigned mlo = 0u, mhi = 0u;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = rb + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          const int bit = mt * 16 + e;
          if (r >= lo && r < hi) { if (bit < 32) mlo |= 1u << bit; else mhi |= 1u << (bit - 32); }
        }
      conv_col_stats<MT, NT>(p, acc, lane, cb, (long)b, mlo, mhi);
    }
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
  // (1x1 with fragment-order weights takes the weight-fragment kernel too: one tap, weights from L2 straight into MFMA registers)
  if ((p.g.KH * p.g.KW > 1 || p.w_packed) && !p.force_generic && p.g.stride == 1) {
    const int rc = launch_conv_halo(p, prec, s);
    if (rc != CRAFT_ERR_UNSUPPORTED) return rc;
  }
  if (p.w_packed) return CRAFT_ERR_UNSUPPORTED;     // the generic implicit GEMM reads raw fp32 weights
  if (p.g.in_norm) return CRAFT_ERR_UNSUPPORTED;    // lazy input normalisation is a k_conv_halo feature
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
