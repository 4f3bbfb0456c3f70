// Halo-tile NHWC convolution, weight-fragment variant (packed 16-bit weights: bf16 / fp16 / f16x3).
//
// Same tiling as k_conv_halo (kernels_conv.hip): a block owns an 8x16 patch of output pixels and BN output
// channels, the (8+KH-1) x (16+KW-1) halo of 32 input channels is staged into LDS once per channel chunk and all
// KH*KW taps run from it.  What differs is the weight operand: craft_pack_weights lays the weights out in MFMA
// FRAGMENT order ([k-tile][32-column block][plane][k-half][lane][8 halves]), so a wave fetches the B operand of
// one 32x32x16 MFMA with ONE fully coalesced 1 KiB global_load_dwordx4 straight into the registers the MFMA
// reads.  The weights never touch LDS:
//   * no per-K-tile weight staging (4 ds_write_b128 + 8 ds_read_b128 per wave and K-tile gone),
//   * no per-K-tile barrier: the only block-wide synchronisation left is ONE barrier per channel chunk (the halo
//     is double-buffered), i.e. every KH*KW K-tiles; in between the four waves run free, so one wave's LDS /
//     global latency overlaps the other waves' MFMAs instead of stalling the whole block at a barrier.
// The weight tile of a block (BN x K, <= 1 MB) is shared by every block of the same column block and lives in L2.
// Each wave owns 32 output channels (NT = 1) and MT = 128 / (32 * WM) row fragments:
//   BN = 128: WM = 1, WN = 4, MT = 4   |   BN = 64: WM = 2, WN = 2, MT = 2.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "conv_epilogue.hpp"

namespace craft {

constexpr int WF_PATCH_H = 8, WF_PATCH_W = 16;

// scheduling pipeline of one k-half: N x { 1 MFMA, 1 LDS op, up to 5 VALU, 1 VMEM read }
template <int N> __device__ __forceinline__ void wf_interleave() {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
  }
}

// TT: number of taps when known at compile time (5: 1x5 / 5x1, 9: 3x3; 0: run-time KH*KW).  With static taps the tap loop
// is unrolled and the fp32 -> fp16-plane conversion of the next chunk's halo is cut into one piece per k-half, each in
// the same scheduling region as that k-half's MFMAs (VALU work only hides behind MFMAs of the same wave).
// TERMS (f16x3 only): which terms of the split product run -- bit 0: lo(activation) x hi(weight), bit 1: hi(activation) x lo(weight),
// bit 2: hi x hi.  7 = the fp32-class product.  5 = the WEIGHT operand as one fp16 plane (its lo plane is neither fetched nor multiplied:
// two MFMAs per product): the input-gradient convolutions of the "mixed" training policy (CRAFT_CONV_W16; dY keeps both planes).
template <int PREC, int WM, int WN, bool ENC, int TT, int TERMS = CRAFT_X3_TERMS>
__global__ __launch_bounds__(NTHREADS) void k_conv_halo_wf(ConvGemmParams p, int xcd_map) {
  typedef typename PrecT<PREC>::lds_t lds_t;
  typedef typename FragT<PREC>::t frag_t;
  constexpr int LD = PrecT<PREC>::LD, PL = Planes<PREC>::N;
  constexpr int BM = 128, MT = BM / WM / 32, BN = WN * 32;
  static_assert(WM * WN == NTHREADS / 64, "4 waves");
  constexpr int HR_MAX = (WF_PATCH_H + 4) * WF_PATCH_W;    // 192 halo rows: enough for 5x1 / 1x5 / 3x3
  constexpr int NA = HR_MAX / 32;                          // float4 per thread for one halo chunk
  constexpr int A_ELEMS = PL * HR_MAX * LD;
  __shared__ __attribute__((aligned(16))) lds_t As[2 * A_ELEMS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const ConvGeom& g = p.g;
  const int KH = g.KH, KW = g.KW, T = TT ? TT : KH * KW;
  const int HWd = WF_PATCH_W + KW - 1, HH = WF_PATCH_H + KH - 1, HR = HH * HWd;
  const int tiles_x = (g.W + WF_PATCH_W - 1) / WF_PATCH_W, tiles_y = (g.H + WF_PATCH_H - 1) / WF_PATCH_H;
  // (xcd_map: block b runs on XCD b % 8; give each XCD a contiguous eighth of the patch list so that neighbouring patches share the
  // halo rows they both read through ONE L2 -- CRAFT_CONV_XCD, developer A/B)
  int bid = xcd_map == 1 ? xcd_chunk(blockIdx.x, gridDim.x) : blockIdx.x;
  int by = blockIdx.y;
  if (xcd_map == 2 && gridDim.y == 2) {
    // CRAFT_CONV_XCD=2 (developer A/B): 256 < blocks <= 512 = at most two per CU.  If the dispatcher gives CU slot j the blocks j and j + 256,
    // make that pair share ONE column block (and be neighbouring patches), so that their weight-fragment loads -- the same addresses at
    // about the same time -- can meet in the CU's vector L1 instead of both going to L2.
    const int total = 2 * gridDim.x, lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int r1 = total - 256, T = total / 4;
    if (r1 > 0 && T <= r1 && T <= 256) {
      const int j = lin & 255, r = lin >> 8;
      if (j < T) { by = 0; bid = j + r * T; } else { by = 1; bid = (j - T) + r * (256 - T); }
    }
  }
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = ty * WF_PATCH_H, x0 = tx * WF_PATCH_W;
  const int n0 = by * BN;
  const int ctot = g.c0 + g.c1, nchunk = ctot / BK;
  const long img = (long)b * g.H * g.W;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * 32;
  const int c4 = tid & 7, r0 = tid >> 3;

  // ---- halo gather: per-thread pixel offsets (-1: outside the image / beyond the halo)
  const int hwd_magic = 65536 / HWd + 1;
  int hpix[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int hr = r0 + 32 * i;
    const int hy = (hr * hwd_magic) >> 16, hx = hr - hy * HWd;     // hr / HWd for hr < 192, HWd in 16..20 (exact)
    const int y = y0 - g.padH + hy, x = x0 - g.padW + hx;
    hpix[i] = (hr < HR && y >= 0 && y < g.H && x >= 0 && x < g.W) ? y * g.W + x : -1;
  }
  int hoff[NA];                                  // clamped pixel index within the whole tensor
#pragma unroll
  for (int i = 0; i < NA; ++i) hoff[i] = (int)img + max(hpix[i], 0);
  auto fetch_halo = [&](int chunk, float4 (&r)[NA]) __attribute__((always_inline)) {
    const int cb = chunk * BK;
    const float* sp; int ld, c;
    if (cb < g.c0) { sp = g.seg0; ld = g.ld0; c = cb; } else { sp = g.seg1; ld = g.ld1; c = cb - g.c0; }
    // unconditional loads (clamped pixel); out-of-image taps are zeroed by a value select in store_halo.  Address =
    // wave-uniform base + 32-bit lane offset (the launcher routes tensors beyond 2^31 elements elsewhere): 2 VALU per load
    const float* spc = sp + c;
#pragma unroll
    for (int i = 0; i < NA; ++i) r[i] = *reinterpret_cast<const float4*>(spc + (unsigned)(hoff[i] * ld + c4 * 4));
  };
  // (mean, rstd) of this thread's 4 input channels for the lazy input normalisation of chunk `chunk`, image b
  auto load_norm = [&](int chunk, float4& mu, float4& rs) __attribute__((always_inline)) {
    mu = make_float4(0.f, 0.f, 0.f, 0.f); rs = make_float4(1.f, 1.f, 1.f, 1.f);
    if (ENC && g.in_norm) {
      const float* t = g.in_norm + ((long)b * g.c0 + chunk * BK + c4 * 4) * 2;
      const float4 t0 = *reinterpret_cast<const float4*>(t), t1 = *reinterpret_cast<const float4*>(t + 4);
      mu = make_float4(t0.x, t0.z, t1.x, t1.z);
      rs = make_float4(t0.y, t0.w, t1.y, t1.w);
    }
  };
  // one float4 of the halo (row r0 + 32 i): normalise / zero / split -> LDS buffer hb
  auto store_piece = [&](int hb, const float4 (&r)[NA], int i, const float4& mu, const float4& rs) __attribute__((always_inline)) {
    lds_t* A0 = &As[hb * A_ELEMS];
    const int row = r0 + 32 * i;                // rows >= HR are written too (zeros, never read): no exec branch
    const bool ok = hpix[i] >= 0;
    float4 v = r[i];
    if (ENC && g.in_norm) {
      v.x = fmaxf((v.x - mu.x) * rs.x, 0.f); v.y = fmaxf((v.y - mu.y) * rs.y, 0.f);
      v.z = fmaxf((v.z - mu.z) * rs.z, 0.f); v.w = fmaxf((v.w - mu.w) * rs.w, 0.f);
    }
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    if constexpr (PREC == CRAFT_PREC_BF16) {
      bf16x4 h;
      h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
      *reinterpret_cast<bf16x4*>(&A0[row * LD + c4 * 4]) = h;
    } else {
      if constexpr (PREC == CRAFT_PREC_F16X3) {
        f16x4 h, l;
        split_f16x3(v, h, l);
        *reinterpret_cast<f16x4*>(&A0[row * LD + c4 * 4]) = h;
        *reinterpret_cast<f16x4*>(&A0[(HR_MAX + row) * LD + c4 * 4]) = l;
      } else {
        f16x4 h;
        h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
        *reinterpret_cast<f16x4*>(&A0[row * LD + c4 * 4]) = h;
      }
    }
  };
  auto store_halo = [&](int hb, int chunk, const float4 (&r)[NA]) __attribute__((always_inline)) {
    float4 mu, rs;
    load_norm(chunk, mu, rs);
#pragma unroll
    for (int i = 0; i < NA; ++i) store_piece(hb, r, i, mu, rs);
  };

  // ---- weight fragments: [kt][nb][pl][kk][lane][8].  Column blocks beyond the packed width re-read the last one
  // (those output columns are discarded / overwritten by the epilogue).
  const int NBtot = (p.cout + 31) / 32;
  const int nb = min((n0 + wn0) / 32, NBtot - 1);
  const uint16_t* wb = reinterpret_cast<const uint16_t*>(p.W) + (long)nb * (PL * 1024) + lane * 8;
  const long kt_stride = (long)NBtot * (PL * 1024);
  // B ring: BD k-halves in flight (static taps: 4, so that an L2 round trip under load is covered; else 2).  Slot of
  // k-half number g (counted from the start of the K loop) = g % BD; 2*TT*2 is a multiple of 4, so the slots are
  // compile-time constants inside the 2x unrolled chunk loop.
  constexpr int BD = (TT > 0 && !(ENC && MT == 4 && TT == 9)) ? 4 : 2;
  frag_t bq[PL][BD];
  auto fetch_b = [&](int kt, int kk, int slot) __attribute__((always_inline)) {
    const uint16_t* q = wb + kt * kt_stride + kk * 512;
#pragma unroll
    for (int pl = 0; pl < PL; ++pl)
      if (pl == 0 || (TERMS & 2)) bq[pl][slot] = *reinterpret_cast<const frag_t*>(q + pl * 1024);
  };

  // lane's base halo element offsets for its MT output-row fragments
  int arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int r = wm0 + mt * 32 + patch_row_perm(lane & 31);       // (conflict-free LDS lane groups, conv_epilogue.hpp)
    arow[mt] = ((r >> 4) * HWd + (r & 15)) * LD + (lane >> 5) * 8;
  }
  // A fragments of one k-half: hi (and lo) plane rows of the halo buffer `hb`, shifted by the tap offset
  auto read_a = [&](int hb, int toff, int kk, frag_t (&h)[MT], frag_t (&l)[MT]) __attribute__((always_inline)) {
    const lds_t* A0 = &As[hb * A_ELEMS + toff + kk * 16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      h[mt] = *reinterpret_cast<const frag_t*>(&A0[arow[mt]]);
      if constexpr (PL == 2) l[mt] = *reinterpret_cast<const frag_t*>(&A0[HR_MAX * LD + arow[mt]]);
    }
  };

  f32x16 acc[MT][1];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[mt][0][e] = 0.f;
  // MFMAs of one k-half; term-major order so that consecutive MFMAs hit different accumulators
  auto mma_half = [&](const frag_t (&h)[MT], const frag_t (&l)[MT], int slot) __attribute__((always_inline)) {
    if constexpr (PL == 2) {
      if constexpr (TERMS & 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][0] = mfma16<PREC>(l[mt], bq[0][slot], acc[mt][0]);
      }
      if constexpr (TERMS & 2) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][0] = mfma16<PREC>(h[mt], bq[1][slot], acc[mt][0]);
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] = mfma16<PREC>(h[mt], bq[0][slot], acc[mt][0]);
  };

  float4 ra[NA], rb[NA];
  fetch_halo(0, ra);
#pragma unroll
  for (int g = 0; g < BD; ++g) {          // k-halves 0 .. BD-1 of chunk 0: tap g / 2, half g % 2
    const int tp = min(g / 2, T - 1);
    fetch_b(tp * nchunk, g & 1, g);
  }
  fetch_halo(min(1, nchunk - 1), rb);                     // issued before ra is consumed: both HBM round trips overlap
  store_halo(0, 0, ra);
  __syncthreads();

  // K loop: chunk-outer / tap-inner; K-tile index in the packed weights = tap * nchunk + chunk.  Software pipeline:
  //   * A fragments (LDS) are requested ONE k-half (MT*3 MFMAs) ahead (two register sets a0 / a1),
  //   * the B fragments (L2) of k-half kk of the NEXT K-tile are requested into the same registers right after the
  //     MFMAs that consumed them (two k-halves ahead of their use, no second register set, no copies),
  //   * halos (HBM/L2) are requested TWO chunks ahead (register sets ra / rb alternate, hence the 2x unrolled chunk
  //     loop): with few input channels a chunk is shorter than an HBM round trip.  Chunk c+1 is written to the idle
  //     LDS buffer after the second tap of chunk c and published by the single barrier of the chunk, placed right
  //     before the first read of that buffer (in the middle of the last tap).
  // Tap offsets are tracked incrementally (tx, trow): no division in the loop.
  frag_t a0h[MT], a0l[MT], a1h[MT], a1l[MT];
  const int smid = min(1, T - 1);
  int hb = 0;
  read_a(0, 0, 0, a0h, a0l);
  auto do_chunk = [&](int chunk, float4 (&rnear)[NA], float4 (&rfar)[NA], auto phase_c) __attribute__((always_inline)) {
    constexpr int PHASE = decltype(phase_c)::value;      // ring slot of this chunk's first k-half
    const int cn = min(chunk + 1, nchunk - 1);
    if constexpr (TT == 0) fetch_halo(min(chunk + 2, nchunk - 1), rfar);     // rfar held this chunk's halo, already in LDS
    int tx = 0, trow = 0;                                 // tap = (trow / HWd) * KW + tx; toff = (trow + tx) * LD
    if constexpr (TT > 0) {
      // static taps: k-half h = 2 * tap + kk converts halo piece (h - 1) / PSTEP of the next chunk, if any
      constexpr int PSTEP = (2 * TT - 2) / NA > 1 ? 2 : 1;
      float4 mu, rs;
      load_norm(cn, mu, rs);
#pragma unroll
      for (int tap = 0; tap < TT; ++tap) {
        const bool last_tap = tap + 1 == TT;
        const int toff = (trow + tx) * LD;
        if (++tx == KW) { tx = 0; trow += HWd; }
        const int toffn = last_tap ? 0 : (trow + tx) * LD;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int h = 2 * tap + kk;
          if (kk == 0) read_a(hb, toff, 1, a1h, a1l);
          else {
            if (last_tap) { __syncthreads(); hb ^= 1; }
            read_a(hb, toffn, 0, a0h, a0l);               // next tile (after the last chunk: a harmless re-read)
          }
          const int slot = (PHASE + h) % BD;
          if (kk == 0) mma_half(a0h, a0l, slot); else mma_half(a1h, a1l, slot);
          {   // B fragments of k-half h + BD (this chunk, or the head of the next one)
            const int hf = h + BD;
            const int ktf = hf < 2 * TT ? (hf / 2) * nchunk + chunk : ((hf - 2 * TT) / 2) * nchunk + cn;
            fetch_b(ktf, hf & 1, slot);
          }
          if (h == 0) fetch_halo(min(chunk + 2, nchunk - 1), rfar);   // (address arithmetic behind the first MFMAs)
          if (h >= 1 && (h - 1) % PSTEP == 0 && (h - 1) / PSTEP < NA && !(last_tap && kk == 1))
            store_piece(hb ^ 1, rnear, (h - 1) / PSTEP, mu, rs);
          wf_interleave<MT * (PL == 2 ? 3 : 1)>();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
    for (int tap = 0; tap < T; ++tap) {
      const bool last_tap = tap + 1 == T;
      const int toff = (trow + tx) * LD;
      if (++tx == KW) { tx = 0; trow += HWd; }
      const int toffn = last_tap ? 0 : (trow + tx) * LD;
      const int ktn = last_tap ? cn : (tap + 1) * nchunk + chunk;
      read_a(hb, toff, 1, a1h, a1l);
      mma_half(a0h, a0l, 0);
      fetch_b(ktn, 0, 0);
      if (tap == smid) store_halo(hb ^ 1, cn, rnear);
      wf_interleave<MT * (PL == 2 ? 3 : 1)>();
      __builtin_amdgcn_sched_barrier(0);
      if (last_tap) { __syncthreads(); hb ^= 1; }
      read_a(hb, toffn, 0, a0h, a0l);                    // next tile (after the last chunk: a harmless re-read)
      mma_half(a1h, a1l, 1);
      fetch_b(ktn, 1, 1);
      wf_interleave<MT * (PL == 2 ? 3 : 1)>();
      __builtin_amdgcn_sched_barrier(0);
    }
    }
  };
  int chunk = 0;
  typedef std::integral_constant<int, 0> P0;
  typedef std::integral_constant<int, (2 * TT) % BD> P1;
  for (; chunk + 1 < nchunk; chunk += 2) {
    do_chunk(chunk, rb, ra, P0());
    do_chunk(chunk + 1, ra, rb, P1());
  }
  if (chunk < nchunk) do_chunk(chunk, rb, ra, P0());

  // ---- epilogue: GEMM row r of the patch -> token (y0 + r/16, x0 + r%16)
  const int cb = n0 + wn0;
  const int rh4 = 4 * (lane >> 5);
#define BODY(E) conv_epilogue_patch<E, true, MT, 1, true>(p, acc, wm0, lane, cb, img, y0, x0);
  CONV_EPI_DISPATCH(p, BODY)
#undef BODY
  if (ENC && p.stats) {
    unsigned mlo = ~0u, mhi = ~0u;
    if (y0 + WF_PATCH_H > g.H || x0 + WF_PATCH_W > g.W) {      // ragged patch only: per-row validity
      mlo = 0u; mhi = 0u;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = wm0 + mt * 32 + patch_row_perm((e & 3) + 8 * (e >> 2) + rh4);
          const bool ok = (y0 + (r >> 4)) < g.H && (x0 + (r & 15)) < g.W;
          const int bit = mt * 16 + e;
          if (ok) { if (bit < 32) mlo |= 1u << bit; else mhi |= 1u << (bit - 32); }
        }
    }
    conv_col_stats<MT, 1>(p, acc, lane, cb, (long)b, mlo, mhi);
  }
}

template <int PREC, int WM, int WN, int TT> static int launch_wf_tt(const ConvGemmParams& p, hipStream_t s) {
  constexpr int BN = WN * 32;
  const bool enc = p.g.in_norm != nullptr || p.stats != nullptr;
  const int ncols = p.epi == CONV_EPI_MENC ? p.cout + 2 : p.cout;
  const int tiles = ((p.g.W + WF_PATCH_W - 1) / WF_PATCH_W) * ((p.g.H + WF_PATCH_H - 1) / WF_PATCH_H) * (p.g.npix / (p.g.H * p.g.W));
  dim3 grid(tiles, (ncols + BN - 1) / BN, 1);
  const int xm = tuning().conv_xcd;
  if (enc) hipLaunchKernelGGL((k_conv_halo_wf<PREC, WM, WN, true, TT>), grid, dim3(NTHREADS), 0, s, p, xm);
  else if (PREC == CRAFT_PREC_F16X3 && TT > 0 && p.w16) hipLaunchKernelGGL((k_conv_halo_wf<PREC, WM, WN, false, TT, 5>), grid, dim3(NTHREADS), 0, s, p, xm);
  else hipLaunchKernelGGL((k_conv_halo_wf<PREC, WM, WN, false, TT>), grid, dim3(NTHREADS), 0, s, p, xm);
  return (int)hipGetLastError();
}
template <int PREC, int WM, int WN> static int launch_wf_t(const ConvGemmParams& p, hipStream_t s) {
  if ((long)p.g.npix * (long)max(p.g.ld0, p.g.c1 ? p.g.ld1 : 0) >= (1L << 31)) return CRAFT_ERR_UNSUPPORTED;   // 32-bit lane offsets
  const bool dyn = tuning().wf_dynamic_taps;       // A/B: always the run-time tap loop
  const int T = p.g.KH * p.g.KW;
  if (T == 5 && !dyn) return launch_wf_tt<PREC, WM, WN, 5>(p, s);
  if (T == 9 && !dyn) return launch_wf_tt<PREC, WM, WN, 9>(p, s);
  return launch_wf_tt<PREC, WM, WN, 0>(p, s);
}

// packed (fragment-order) weights only; called by launch_conv_halo
int launch_conv_halo_wf(const ConvGemmParams& p, int prec, hipStream_t s) {
  if (conv3x3_c64_applies(p) && !tuning().no_c64) return launch_conv3x3_c64(p, prec, s);     // encoder layer1 shape
  const int ncols = p.epi == CONV_EPI_MENC ? p.cout + 2 : p.cout;
  // BN = 128 only when that still leaves >= 2 blocks per CU's worth of tiles for wide outputs
  int bn = (ncols % 128 == 0 && ncols >= 256) ? 128 : 64;
  if (tuning().halo_bn) bn = tuning().halo_bn == 128 ? 128 : 64;   // developer A/B override
#define GO(PR) do { if (bn == 128) return launch_wf_t<PR, 1, 4>(p, s); else return launch_wf_t<PR, 2, 2>(p, s); } while (0)
  if (prec == CRAFT_PREC_BF16) GO(CRAFT_PREC_BF16);
  if (prec == CRAFT_PREC_F16) GO(CRAFT_PREC_F16);
  if (prec == CRAFT_PREC_F16X3) GO(CRAFT_PREC_F16X3);
#undef GO
  return CRAFT_ERR_ARG;
}

// ---------------------------------------------------------------------------------------------
// craft_pack_weights: fp32 [rows][K] -> MFMA fragment order of `prec` (see the header of this file):
//   out[((((kt * NB + nb) * PL + pl) * 2 + kk) * 64 + lane) * 8 + j] = plane_pl( w[nb*32 + (lane & 31)][kt*32 + kk*16 + (lane >> 5)*8 + j] )
// NB = ceil(rows / 32) (rows beyond `rows` are zero), PL = 2 for F16X3 (hi = fp16(w), lo = fp16(w - hi)), else 1.
// ---------------------------------------------------------------------------------------------
template <int PREC>
__global__ void k_pack_weights_wf(const float* __restrict__ w, int rows, int K, void* __restrict__ out) {
  typedef typename PrecT<PREC>::lds_t h_t;
  constexpr int PL = Planes<PREC>::N;
  const int NB = (rows + 31) / 32;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (kt, nb, kk, lane)
  const long total = (long)(K / 32) * NB * 128;
  if (i >= total) return;
  const int lane = (int)(i & 63), kk = (int)((i >> 6) & 1);
  const long t = i >> 7;
  const int nbi = (int)(t % NB);
  const long kt = t / NB;
  const int row = nbi * 32 + (lane & 31);
  const long k = kt * 32 + kk * 16 + (lane >> 5) * 8;
  h_t* o = reinterpret_cast<h_t*>(out) + (((kt * NB + nbi) * PL) * 2 + kk) * 512 + lane * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = row < rows ? w[(long)row * K + k + j] : 0.f;
    const h_t h = (h_t)v;
    o[j] = h;
    if constexpr (PL == 2) o[1024 + j] = (h_t)(v - (float)h);
  }
}

// ---------------------------------------------------------------------------------------------
// craft_pack_conv_weights: the same fragment order straight from PyTorch's conv weight layout [Cout][Cin][KH][KW] -- one launch
// instead of permute / contiguous / cat / flip / zero-pad tensor ops in front of craft_pack_weights (a training step re-packs every
// weight after every optimizer update: ~300 host-latency-bound tiny kernels per step).
//   source: up to two weights concatenated along Cout (w0: cout0 rows, w1: cout1 rows; the z | r gates of SepConvGRU), and a
//   selection of input channels [a0, a1) u [b0, b1) (all of them, or the hoisted / varying split of the GRU input).
//   forward form   : row = co,  k = (ky * KW + kx) * Cp + c,  c-th selected input channel          (Cp = selected channels, % 32 == 0)
//   transposed form: row = c-th selected input channel, k = (ky * KW + kx) * Cp + co, value W[co][ci][KH-1-ky][KW-1-kx]   (Cp = round_up(Cout, 32))
// the operand of the INPUT-gradient convolution.  Rows / channels beyond the source are zero.
// ---------------------------------------------------------------------------------------------
struct PackConvParams {
  const float* w0; const float* w1; int cout0, cout1, Cin, KH, KW, a0, a1, b0, b1, transposed, rows, Cp;
};
template <int PREC>
__global__ void k_pack_conv_weights(PackConvParams p, void* __restrict__ out) {
  typedef typename PrecT<PREC>::lds_t h_t;
  constexpr int PL = Planes<PREC>::N;
  const int NB = (p.rows + 31) / 32, taps = p.KH * p.KW;
  const long K = (long)taps * p.Cp;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (kt, nb, kk, lane)
  const long total = (K / 32) * NB * 128;
  if (i >= total) return;
  const int lane = (int)(i & 63), kk = (int)((i >> 6) & 1);
  const long t = i >> 7;
  const int nbi = (int)(t % NB);
  const long kt = t / NB;
  const int row = nbi * 32 + (lane & 31);
  const long k = kt * 32 + kk * 16 + (lane >> 5) * 8;
  const int tap = (int)(k / p.Cp), c0 = (int)(k - (long)tap * p.Cp);
  const int ky = tap / p.KW, kx = tap - ky * p.KW;
  const int nsel = (p.a1 - p.a0) + (p.b1 - p.b0), cout = p.cout0 + p.cout1;
  h_t* o = reinterpret_cast<h_t*>(out) + (((kt * NB + nbi) * PL) * 2 + kk) * 512 + lane * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c0 + j;
    int co, s, sy, sx;                       // source (co, s-th selected channel, tap)
    if (p.transposed) { co = c; s = row; sy = p.KH - 1 - ky; sx = p.KW - 1 - kx; }
    else { co = row; s = c; sy = ky; sx = kx; }
    float v = 0.f;
    if (co < cout && s < nsel) {
      const int ci = s < p.a1 - p.a0 ? p.a0 + s : p.b0 + (s - (p.a1 - p.a0));
      const float* w = co < p.cout0 ? p.w0 + (long)co * p.Cin * taps : p.w1 + (long)(co - p.cout0) * p.Cin * taps;
      v = w[((long)ci * p.KH + sy) * p.KW + sx];
    }
    const h_t h = (h_t)v;
    o[j] = h;
    if constexpr (PL == 2) o[1024 + j] = (h_t)(v - (float)h);
  }
}

int launch_pack_conv_weights(const float* w0, int cout0, const float* w1, int cout1, int Cin, int KH, int KW, int a0, int a1, int b0, int b1,
                             int transposed, int prec, void* out, hipStream_t s) {
  if (cout0 <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || cout1 < 0 || (cout1 > 0 && !w1)) return CRAFT_ERR_ARG;
  if (a0 < 0 || a1 < a0 || a1 > Cin || b0 < 0 || b1 < b0 || b1 > Cin) return CRAFT_ERR_ARG;
  PackConvParams p = {};
  p.w0 = w0; p.w1 = w1; p.cout0 = cout0; p.cout1 = cout1; p.Cin = Cin; p.KH = KH; p.KW = KW; p.a0 = a0; p.a1 = a1; p.b0 = b0; p.b1 = b1;
  p.transposed = transposed;
  const int nsel = (a1 - a0) + (b1 - b0), cout = cout0 + cout1;
  p.rows = transposed ? nsel : cout;
  p.Cp = ((transposed ? cout : nsel) + 31) / 32 * 32;
  const long total = ((long)KH * KW * p.Cp / 32) * ((p.rows + 31) / 32) * 128;
  dim3 grid((unsigned)((total + 255) / 256));
  if (prec == CRAFT_PREC_BF16) hipLaunchKernelGGL((k_pack_conv_weights<CRAFT_PREC_BF16>), grid, dim3(256), 0, s, p, out);
  else if (prec == CRAFT_PREC_F16) hipLaunchKernelGGL((k_pack_conv_weights<CRAFT_PREC_F16>), grid, dim3(256), 0, s, p, out);
  else if (prec == CRAFT_PREC_F16X3) hipLaunchKernelGGL((k_pack_conv_weights<CRAFT_PREC_F16X3>), grid, dim3(256), 0, s, p, out);
  else return CRAFT_ERR_UNSUPPORTED;
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// craft_pack_conv_weights_batch: n craft_pack_conv_weights jobs in ONE launch (a training step re-packs every convolution weight --
// forward and transposed forms, 71 operands at configs[3] -- after every optimizer update; the operands' addresses are stable across
// steps, so the descriptor table lives in DEVICE memory and is built once).  Block -> (job, local block) through a prefix table.
// ---------------------------------------------------------------------------------------------
struct PackConvJob { PackConvParams p; void* out; int prec; int pad_; };
template <int PREC>
__device__ __forceinline__ void pack_conv_block(const PackConvParams& p, void* __restrict__ out, long i) {
  typedef typename PrecT<PREC>::lds_t h_t;
  constexpr int PL = Planes<PREC>::N;
  const int NB = (p.rows + 31) / 32, taps = p.KH * p.KW;
  const long K = (long)taps * p.Cp;
  const long total = (K / 32) * NB * 128;
  if (i >= total) return;
  const int lane = (int)(i & 63), kk = (int)((i >> 6) & 1);
  const long t = i >> 7;
  const int nbi = (int)(t % NB);
  const long kt = t / NB;
  const int row = nbi * 32 + (lane & 31);
  const long k = kt * 32 + kk * 16 + (lane >> 5) * 8;
  const int tap = (int)(k / p.Cp), c0 = (int)(k - (long)tap * p.Cp);
  const int ky = tap / p.KW, kx = tap - ky * p.KW;
  const int nsel = (p.a1 - p.a0) + (p.b1 - p.b0), cout = p.cout0 + p.cout1;
  h_t* o = reinterpret_cast<h_t*>(out) + (((kt * NB + nbi) * PL) * 2 + kk) * 512 + lane * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c0 + j;
    int co, sdx, sy, sx;
    if (p.transposed) { co = c; sdx = row; sy = p.KH - 1 - ky; sx = p.KW - 1 - kx; }
    else { co = row; sdx = c; sy = ky; sx = kx; }
    float v = 0.f;
    if (co < cout && sdx < nsel) {
      const int ci = sdx < p.a1 - p.a0 ? p.a0 + sdx : p.b0 + (sdx - (p.a1 - p.a0));
      const float* w = co < p.cout0 ? p.w0 + (long)co * p.Cin * taps : p.w1 + (long)(co - p.cout0) * p.Cin * taps;
      v = w[((long)ci * p.KH + sy) * p.KW + sx];
    }
    const h_t h = (h_t)v;
    o[j] = h;
    if constexpr (PL == 2) o[1024 + j] = (h_t)(v - (float)h);
  }
}
__global__ __launch_bounds__(256) void k_pack_conv_weights_batch(const PackConvJob* __restrict__ jobs, const int* __restrict__ first, int n) {
  int lo = 0, hi = n;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)blockIdx.x >= first[mid]) lo = mid; else hi = mid; }
  const PackConvJob& j = jobs[lo];
  const long i = (long)(blockIdx.x - first[lo]) * 256 + threadIdx.x;
  if (j.prec == CRAFT_PREC_F16X3) pack_conv_block<CRAFT_PREC_F16X3>(j.p, j.out, i);
  else if (j.prec == CRAFT_PREC_F16) pack_conv_block<CRAFT_PREC_F16>(j.p, j.out, i);
  else pack_conv_block<CRAFT_PREC_BF16>(j.p, j.out, i);
}
size_t pack_conv_job_bytes() { return sizeof(PackConvJob); }
// fills one job record (HOST memory, caller copies the table to the device) and returns its block count, or a negative error code
long fill_pack_conv_job(void* job, const float* w0, int cout0, const float* w1, int cout1, int Cin, int KH, int KW, int a0, int a1, int b0, int b1,
                        int transposed, int prec, void* out) {
  if (cout0 <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || cout1 < 0 || (cout1 > 0 && !w1) || !w0 || !out) return -CRAFT_ERR_ARG;
  if (a0 < 0 || a1 < a0 || a1 > Cin || b0 < 0 || b1 < b0 || b1 > Cin) return -CRAFT_ERR_ARG;
  if (prec != CRAFT_PREC_BF16 && prec != CRAFT_PREC_F16 && prec != CRAFT_PREC_F16X3) return -CRAFT_ERR_UNSUPPORTED;
  PackConvJob j = {};
  PackConvParams& p = j.p;
  p.w0 = w0; p.w1 = w1; p.cout0 = cout0; p.cout1 = cout1; p.Cin = Cin; p.KH = KH; p.KW = KW; p.a0 = a0; p.a1 = a1; p.b0 = b0; p.b1 = b1;
  p.transposed = transposed;
  const int nsel = (a1 - a0) + (b1 - b0), cout = cout0 + cout1;
  p.rows = transposed ? nsel : cout;
  p.Cp = ((transposed ? cout : nsel) + 31) / 32 * 32;
  j.out = out; j.prec = prec;
  memcpy(job, &j, sizeof(j));
  const long total = ((long)KH * KW * p.Cp / 32) * ((p.rows + 31) / 32) * 128;
  return (total + 255) / 256;
}
int launch_pack_conv_weights_batch(const void* jobs_dev, const int* first_dev, int n, int total_blocks, hipStream_t s) {
  if (n <= 0 || total_blocks <= 0) return 0;
  hipLaunchKernelGGL(k_pack_conv_weights_batch, dim3((unsigned)total_blocks), dim3(256), 0, s, static_cast<const PackConvJob*>(jobs_dev), first_dev, n);
  return (int)hipGetLastError();
}

int launch_pack_weights(const float* w, int rows, int K, int prec, void* out, hipStream_t s) {
  if (rows <= 0 || K <= 0) return 0;
  if (prec == CRAFT_PREC_F32) return (int)hipMemcpyAsync(out, w, (size_t)rows * K * sizeof(float), hipMemcpyDeviceToDevice, s);
  if (K % 32) return CRAFT_ERR_ALIGN;
  const long total = (long)(K / 32) * ((rows + 31) / 32) * 128;
  dim3 grid((unsigned)((total + 255) / 256));
  if (prec == CRAFT_PREC_BF16) hipLaunchKernelGGL((k_pack_weights_wf<CRAFT_PREC_BF16>), grid, dim3(256), 0, s, w, rows, K, out);
  else if (prec == CRAFT_PREC_F16) hipLaunchKernelGGL((k_pack_weights_wf<CRAFT_PREC_F16>), grid, dim3(256), 0, s, w, rows, K, out);
  else if (prec == CRAFT_PREC_F16X3) hipLaunchKernelGGL((k_pack_weights_wf<CRAFT_PREC_F16X3>), grid, dim3(256), 0, s, w, rows, K, out);
  else return CRAFT_ERR_ARG;
  return (int)hipGetLastError();
}

}  // namespace craft
