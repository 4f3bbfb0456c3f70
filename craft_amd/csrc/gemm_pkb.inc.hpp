// Batched GEMM over PACKED operands (include/craft_hip.h: craft_gemm_pk) -- the attention products of the training pass.
//
//   C[z][m][n] = sum_k A_z[m, k] * B_z[n, k]        z = (outer, inner) batch, fp32 out, 16-bit / f16x3 operands
//
// Both operands are packs  P[plane][channel group][row][32]  (k_pack_operands, or written directly by a producer such as
// k_attn_softmax_fwd), and each declares along which of its two axes the contraction runs:
//
//   kind ROWS  K = the rows of the pack, M (N) = its channels.  A 32 x 32 K-tile chunk is 2 KiB contiguous; the MFMA operand (a column
//              of the chunk per lane) comes out of LDS through ds_read_b64_tr_b16 -- the layout of kernels_gemm_pk.hip;
//   kind CH    K = the channels, M (N) = the rows.  The same 2 KiB chunk is now [32 m][32 k]: a lane reads its 8 consecutive k with one
//              ds_read_b128.  The LDS-DMA writes a chunk lane-linearly but each lane fetches an arbitrary 16 bytes, so the XOR swizzle
//              that makes those reads conflict-free ((row >> 2) & 3 on the 16-byte slot of a 64-byte row: every 16-lane service group
//              of ds_read_b128 then covers all 16 slots of the 256-byte bank row) is applied on the GLOBAL side of the copy.
//
// With P packed as (rows i, channels j) this covers the three products of an attention layer without a transposed copy of anything:
//   O = P V        A = P   (CH: m = i, k = j)      B = V  (ROWS: k = j, n = c)
//   dV = P^T dO    A = P   (ROWS: k = i, m = j)    B = dO (ROWS: k = i, n = c)
//   dP = dO V^T    A = dO  (CH: m = i, k = c)      B = V  (CH: n = j, k = c)
// (and Q K^T / dS K / dS^T Q of the scores in the same three forms).
//
// Structure as k_gemm_pk: 8 waves, tiles up to 256 x 256, the K loop a pure LDS-DMA copy (inline asm, hand-placed waits) -- but with
// NST >= 2 stages: P V with 128 output columns is HBM-bound (1 GB of P per call) and one block per CU has nobody to hide a 2 us DMA
// round trip behind; NST - 1 K-tiles stay in flight.
#pragma once
#include "launch.hpp"

namespace craft {

#define CRAFT_LDS __attribute__((address_space(3)))
typedef __fp16 pkb_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __bf16 pkb_bf16x4 __attribute__((__vector_size__(4 * sizeof(__bf16))));

struct PkbOperand {
  const unsigned char* base;
  long plane; unsigned cg;            // byte strides: cg = rows_p * 64, plane = ncg * cg
  long row0, row_outer, row_inner;    // first row of batch (outer, inner): row0 + outer * row_outer + inner * row_inner
  int cg0, cg_outer, cg_inner;        // first channel group, likewise
};
struct PkbParams {
  PkbOperand A, B;
  float* C; long ldc, c_outer, c_inner;
  int c_blk_shift; long c_blk_stride;   // > 0: output columns in blocks of 2^shift, consecutive blocks c_blk_stride elements apart (CRAFT_PK_CBLK)
  float alpha;
  int inner, nbatch;
  int M, N, K;                        // K: padded to a multiple of 32 (zeros in BOTH packs)
  int ntile_m, ntile_n;
};

constexpr int pkb_stages(int planes, int bm, int bn) {
  const int stage = planes * (bm + bn) * 64;
  const int n = (152 * 1024) / stage;
  return n > 4 ? 4 : n;
}

template <int PLANES, int BM, int BN, int WM, int WN, bool BF16, int AK, int BK>
__global__ __launch_bounds__(512) void k_gemm_pkb(PkbParams p) {
  constexpr int MT = BM / WM / 32, NT = BN / WN / 32;
  constexpr int A_CH = PLANES * (BM / 32), B_CH = PLANES * (BN / 32);          // 2 KiB chunks per stage
  constexpr int STAGE = (A_CH + B_CH) * 2048;
  constexpr int NST = pkb_stages(PLANES, BM, BN), D = NST - 1;
  constexpr int NDMA = 2 * (A_CH + B_CH), DPW = (NDMA + 7) / 8;                 // 1 KiB DMA pieces per stage / per wave
  constexpr int DUMMY = 8 * DPW - NDMA;                                         // every wave issues exactly DPW loads per K-tile (the vmcnt
  static_assert(NST >= 2, "tile too large for the LDS");                        // arithmetic below): the surplus lands in a scratch KiB each
  __shared__ __attribute__((aligned(1024))) unsigned char S[NST * STAGE + DUMMY * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware map: XCD x owns batches x, x + 8, ... and walks the tiles of one batch back to back (its operands stay in that L2)
  const int ntile = p.ntile_m * p.ntile_n;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int bt = xcd + 8 * (jj / ntile), tile = jj - (jj / ntile) * ntile;
  if (bt >= p.nbatch) return;
  const int tm = tile % p.ntile_m, tn = tile / p.ntile_m;
  const int bo = bt / p.inner, bi = bt - bo * p.inner;
  const int nk = p.K / 32;

  // ---- DMA plan: one buffer descriptor per piece (its base carries operand, plane, chunk, half), a per-piece K step and lane offset
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  unsigned blo[DPW], bhi[DPW], kstep[DPW], voff[DPW];
  const unsigned v_lin = lane * 16;
  const unsigned v_swz = (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int id = min(wave * DPW + i, NDMA - 1), ch = id >> 1, half = id & 1;
    const bool isA = ch < A_CH;
    const PkbOperand& X = isA ? p.A : p.B;
    const int kind = isA ? AK : BK;
    const int per = isA ? BM / 32 : BN / 32;
    const int c2 = isA ? ch : ch - A_CH;
    const int pl = c2 / per;
    const int ext = ((isA ? p.M : p.N) + 31) / 32;                            // 32-wide groups of this operand's M (N) extent
    const int g = min((isA ? tm : tn) * per + c2 % per, ext - 1);             // (clamped: the surplus accumulators are never stored)
    const long row = X.row0 + (long)bo * X.row_outer + (long)bi * X.row_inner;
    const long cg = X.cg0 + (long)bo * X.cg_outer + (long)bi * X.cg_inner;
    const unsigned char* base = X.base + (long)pl * X.plane + half * 1024;
    if (kind == 0) base += (cg + g) * X.cg + row * 64;                          // K down the rows: step 2 KiB
    else base += cg * X.cg + (row + 32L * g) * 64;                              // K along the channel groups: step = one group
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    blo[i] = __builtin_amdgcn_readfirstlane((unsigned)a);
    bhi[i] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    kstep[i] = __builtin_amdgcn_readfirstlane(kind == 0 ? 2048u : X.cg);
    voff[i] = kind == 0 ? v_lin : v_swz;
  }
  const unsigned lds0 = (unsigned)(size_t)(CRAFT_LDS unsigned char*)(S);
  auto dma1 = [&](int stage, int kt, int i) __attribute__((always_inline)) {
    {
      const int id = wave * DPW + i;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(id < NDMA ? stage * STAGE + id * 1024 : NST * STAGE + (id - NDMA) * 1024));
      const unsigned soff = (unsigned)kt * kstep[i];
      u32x4 d;
      d[0] = blo[i]; d[1] = bhi[i]; d[2] = 0xffffffffu; d[3] = 0x00020000u;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff[i]), "s"(dst), "s"(d), "s"(soff) : "memory");
    }
  };
  auto dma = [&](int stage, int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < DPW; ++i) dma1(stage, kt, i);
  };

  // ---- fragments.  k-slot assignment of a 16-k MFMA step (the same for both operands): lane half h = lane >> 5 holds k = 8 h .. 8 h + 7.
  //   ROWS: two ds_read_b64_tr_b16 (rows 8 h + 0..3 and 8 h + 4..7 of the step; within a 16-lane group lane j supplies the address of row
  //         j >> 2, columns 4 (j & 3) .., and receives column j);   CH: one ds_read_b128 of the lane's row, slot (2 ks + h) ^ swizzle.
  const int wm = wave / WN, wn = wave - wm * WN;
  const int j16 = lane & 15, mh = (lane >> 4) & 1, h = lane >> 5, r32 = lane & 31;
  const unsigned off_t = (unsigned)((8 * h + (j16 >> 2)) * 64 + (16 * mh + 4 * (j16 & 3)) * 2);
  const unsigned off_c = (unsigned)(r32 * 64), swz = (unsigned)((r32 >> 2) & 3);
  typedef typename std::conditional<BF16, bf16x8, f16x8>::type frag_t;
  auto frag = [&](int kind, unsigned chunk, int ks) __attribute__((always_inline)) {
    frag_t v;
    if (kind == 0) {
      const unsigned addr = chunk + off_t + (unsigned)ks * 1024u;
      if constexpr (BF16) {
        const pkb_bf16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((CRAFT_LDS pkb_bf16x4*)((CRAFT_LDS unsigned char*)(S) + addr));
        const pkb_bf16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((CRAFT_LDS pkb_bf16x4*)((CRAFT_LDS unsigned char*)(S) + addr + 256));
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = r0[i]; v[4 + i] = r1[i]; }
      } else {
        const pkb_fp16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((CRAFT_LDS pkb_fp16x4*)((CRAFT_LDS unsigned char*)(S) + addr));
        const pkb_fp16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((CRAFT_LDS pkb_fp16x4*)((CRAFT_LDS unsigned char*)(S) + addr + 256));
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = (_Float16)r0[i]; v[4 + i] = (_Float16)r1[i]; }
      }
    } else {
      const unsigned addr = chunk + off_c + ((((unsigned)(2 * ks + h)) ^ swz) << 4);
      v = *reinterpret_cast<const frag_t*>(S + addr);
    }
    return v;
  };
  auto mma = [&](const frag_t& a, const frag_t& b, f32x16& c) __attribute__((always_inline)) {
    if constexpr (BF16) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // ---- K loop: tiles kt .. kt + D - 1 in flight; a tile index past the end re-reads the last tile into a stage nobody reads again
#pragma unroll
  for (int t = 0; t < D; ++t) dma(t, min(t, nk - 1));
  int st = 0;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW * (D - 1)) : "memory");      // this wave's pieces of tile kt have landed ...
    __syncthreads();                                                          // ... and everybody's; stage (kt - 1) % NST is free
    // the next tile's DPW pieces are issued BETWEEN the MFMA groups of this one: right after the barrier all 8 waves would sit in their
    // DMA issue (~100 cycles a piece beside LDS reads) with the matrix pipes idle
    const int nst = st == 0 ? NST - 1 : st - 1, nkt = min(kt + D, nk - 1);
    constexpr int NPAIR = 2 * MT * NT;
    const unsigned sb = (unsigned)st * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      frag_t ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        ah[mt] = frag(AK, sb + (unsigned)(wm * MT + mt) * 2048u, ks);
        if constexpr (PLANES == 2) al[mt] = frag(AK, sb + (unsigned)(BM / 32 + wm * MT + mt) * 2048u, ks);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        bh[nt] = frag(BK, sb + (unsigned)(A_CH + wn * NT + nt) * 2048u, ks);
        if constexpr (PLANES == 2) bl[nt] = frag(BK, sb + (unsigned)(A_CH + BN / 32 + wn * NT + nt) * 2048u, ks);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (PLANES == 2) {
            mma(al[mt], bh[nt], acc[mt][nt]);
            mma(ah[mt], bl[nt], acc[mt][nt]);
          }
          mma(ah[mt], bh[nt], acc[mt][nt]);
          const int q = (ks * MT + mt) * NT + nt;                             // pieces [ceil(q DPW / NPAIR), ceil((q + 1) DPW / NPAIR)) go here
#pragma unroll
          for (int c = (q * DPW + NPAIR - 1) / NPAIR; c < ((q + 1) * DPW + NPAIR - 1) / NPAIR; ++c) dma1(nst, nkt, c);
        }
    }
    st = st + 1 == NST ? 0 : st + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // (the surplus loads of the tail)

  // ---- epilogue (C/D layout: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 h): plain stores, 128 contiguous bytes per row
  float* cb = p.C + (long)bo * p.c_outer + (long)bi * p.c_inner;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = tn * BN + (wn * NT + nt) * 32 + r32;
    if (n >= p.N) continue;
    const long noff = p.c_blk_shift ? (long)(n >> p.c_blk_shift) * p.c_blk_stride + (n & ((1 << p.c_blk_shift) - 1)) : (long)n;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m0 = tm * BM + (wm * MT + mt) * 32 + 4 * h;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (e & 3) + 8 * (e >> 2);
        if (m < p.M) cb[(long)m * p.ldc + noff] = acc[mt][nt][e] * p.alpha;
      }
    }
  }
}

// one translation unit per operand-kind pair (kernels_gemm_pkb_*.hip): 3 tiles x 3 operand modes each
template <int AK, int BK>
int launch_gemm_pkb_kind(PkbParams& p, int prec, hipStream_t s) {
  // wide outputs: 256 x 256 (MFMA-bound).  Narrow outputs (N <= 128: O = P V, dQ = dS K) stream the A operand once and are HBM-bound:
  // 128-row tiles with 4 stages keep 96 KiB in flight per CU and quantise to ~3 rounds of blocks instead of 1.5
  const int bn = p.N > 128 ? 256 : (p.N > 64 ? 128 : 64);
  const int bm = bn == 256 ? 256 : 128;
  p.ntile_m = (p.M + bm - 1) / bm;
  p.ntile_n = (p.N + bn - 1) / bn;
  const long tiles = (long)p.ntile_m * p.ntile_n;
  dim3 grid((unsigned)(8 * tiles * ((p.nbatch + 7) / 8)));
#define GO2(PL, BM_, BN_, WM_, WN_, BF) hipLaunchKernelGGL((k_gemm_pkb<PL, BM_, BN_, WM_, WN_, BF, AK, BK>), grid, dim3(512), 0, s, p)
#define GO(PL, BF) do { \
    if (bn == 256) GO2(PL, 256, 256, 2, 4, BF); \
    else if (bn == 128) GO2(PL, 128, 128, 2, 4, BF); \
    else GO2(PL, 128, 64, 4, 2, BF); } while (0)
  if (prec == CRAFT_PREC_F16X3) GO(2, false);
  else if (prec == CRAFT_PREC_F16) GO(1, false);
  else GO(1, true);
#undef GO
#undef GO2
  return (int)hipGetLastError();
}

int launch_gemm_pkb_tt(PkbParams& p, int prec, hipStream_t s);
int launch_gemm_pkb_ct(PkbParams& p, int prec, hipStream_t s);
int launch_gemm_pkb_cc(PkbParams& p, int prec, hipStream_t s);

}  // namespace craft
