// Halo-tile NHWC convolution on the MFMA engine (stride 1, "same" padding, KH*KW > 1).
//
// A block owns an 8x16 patch of output pixels (128 GEMM rows) and BN output channels.  For each chunk of
// 32 input channels the (8+KH-1) x (16+KW-1) halo patch is staged into LDS ONCE (zero-filled outside the
// image, converted / split as the precision requires) and all KH*KW taps run from it: the MFMA row
// fragment of output pixel (py, px) for tap (dy, dx) is halo row (py+dy)*(16+KW-1) + (px+dx), i.e. the
// lane's base row plus a wave-uniform offset.  Compared with the generic implicit GEMM (k_gemm_conv), which
// re-gathers and re-converts the activation tile for every tap, activation staging drops by 5x (1x5, 5x1)
// to 6.4x (3x3); only the weight tile [BN x 32] is staged per (chunk, tap), double-buffered with one
// barrier per K-tile.  Epilogues are shared with k_gemm_conv (conv_epilogue.hpp).  This is the RAW-weight
// (fp32 [Cout][KH][KW][Cin]) kernel; pre-packed 16-bit weights take k_conv_halo_wf (kernels_conv_wf.hip).
#include <cstdlib>
#include "conv_epilogue.hpp"

namespace craft {

constexpr int PATCH_H = 8, PATCH_W = 16;

// ENC: the encoder features (lazy InstanceNorm on the input, per-channel statistics of the output) are a
// compile-time variant so that the update-block instantiation keeps its register budget (2 waves per SIMD).
template <int PREC, int BN, bool ENC>
__global__ __launch_bounds__(NTHREADS) void k_conv_halo(ConvGemmParams p) {
  typedef typename PrecT<PREC>::lds_t lds_t;
  constexpr int LD = PrecT<PREC>::LD, PL = Planes<PREC>::N;
  constexpr int BM = 128, WM = 2, WN = 2, MT = BM / WM / 32, NT = BN / WN / 32;
  constexpr int HR_MAX = (PATCH_H + 4) * PATCH_W;       // 192 rows: enough for 5x1 / 1x5 / 3x3
  constexpr int NA = HR_MAX / 32;                       // float4 per thread for one halo chunk
  constexpr int B_ELEMS = PL * BN * LD;
  __shared__ __attribute__((aligned(16))) lds_t As[PL * HR_MAX * LD];
  __shared__ __attribute__((aligned(16))) lds_t Bs[2 * B_ELEMS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const ConvGeom& g = p.g;
  const int KH = g.KH, KW = g.KW, T = KH * KW;
  const int HWd = PATCH_W + KW - 1, HH = PATCH_H + KH - 1, HR = HH * HWd;
  const int tiles_x = (g.W + PATCH_W - 1) / PATCH_W, tiles_y = (g.H + PATCH_H - 1) / PATCH_H;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = ty * PATCH_H, x0 = tx * PATCH_W;
  const int n0 = blockIdx.y * BN;
  const int ctot = g.c0 + g.c1, nchunk = ctot / BK, K = T * ctot;
  const long img = (long)b * g.H * g.W;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
  const int c4 = tid & 7, r0 = tid >> 3;

  // ---- halo gather: per-thread pixel offsets (-1: outside the image / beyond the halo)
  int hpix[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int hr = r0 + 32 * i;
    const int hy = hr / HWd, hx = hr - hy * HWd;
    const int y = y0 - g.padH + hy, x = x0 - g.padW + hx;
    hpix[i] = (hr < HR && y >= 0 && y < g.H && x >= 0 && x < g.W) ? y * g.W + x : -1;
  }
  auto fetch_halo = [&](int chunk, float4 (&r)[NA]) __attribute__((always_inline)) {
    const int cb = chunk * BK;
    const float* sp; int ld, c;
    if (cb < g.c0) { sp = g.seg0; ld = g.ld0; c = cb; } else { sp = g.seg1; ld = g.ld1; c = cb - g.c0; }
    // unconditional loads (clamped pixel); out-of-image taps are zeroed by a value select in store_halo
#pragma unroll
    for (int i = 0; i < NA; ++i) r[i] = *reinterpret_cast<const float4*>(sp + (img + max(hpix[i], 0)) * ld + c + c4 * 4);
  };
  int cur_chunk_stored = 0;     // channel chunk whose halo is being written (for the lazy-norm table)
  auto store_halo = [&](const float4 (&r)[NA]) __attribute__((always_inline)) {
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), rs = make_float4(1.f, 1.f, 1.f, 1.f);
    if (ENC && g.in_norm) {     // (mean, rstd) of this thread's 4 input channels, image b
      const float* t = g.in_norm + ((long)b * g.c0 + cur_chunk_stored * BK + c4 * 4) * 2;
      const float4 t0 = *reinterpret_cast<const float4*>(t), t1 = *reinterpret_cast<const float4*>(t + 4);
      mu = make_float4(t0.x, t0.z, t1.x, t1.z);
      rs = make_float4(t0.y, t0.w, t1.y, t1.w);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int row = r0 + 32 * i;
      if (row >= HR) continue;
      const bool ok = hpix[i] >= 0;
      float4 v = r[i];
      if (ENC && g.in_norm) {
        v.x = fmaxf((v.x - mu.x) * rs.x, 0.f); v.y = fmaxf((v.y - mu.y) * rs.y, 0.f);
        v.z = fmaxf((v.z - mu.z) * rs.z, 0.f); v.w = fmaxf((v.w - mu.w) * rs.w, 0.f);
      }
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      if constexpr (PREC == CRAFT_PREC_F32) {
        *reinterpret_cast<float4*>(&As[row * LD + c4 * 4]) = v;
      } else if constexpr (PREC == CRAFT_PREC_BF16) {
        bf16x4 h;
        h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4*>(&As[row * LD + c4 * 4]) = h;
      } else {
        f16x4 h;
        h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
        *reinterpret_cast<f16x4*>(&As[row * LD + c4 * 4]) = h;
        if constexpr (PREC == CRAFT_PREC_F16X3) {
          f16x4 l;
          l[0] = (_Float16)(v.x - (float)h[0]); l[1] = (_Float16)(v.y - (float)h[1]);
          l[2] = (_Float16)(v.z - (float)h[2]); l[3] = (_Float16)(v.w - (float)h[3]);
          *reinterpret_cast<f16x4*>(&As[(HR_MAX + row) * LD + c4 * 4]) = l;
        }
      }
    }
  };

  // ---- weight tile loader: rows = output channels n0.., k = tap*ctot + chunk*32 .. +32.  Rows beyond cout
  // re-read the last real row: those output columns are discarded / overwritten by the epilogue, and an
  // unconditional load keeps exec-mask branches out of the K loop.
  const float* wp[BN / 32];
#pragma unroll
  for (int i = 0; i < BN / 32; ++i) {
    const int row = n0 + r0 + 32 * i;
    wp[i] = p.W + (long)min(row, p.cout - 1) * K + c4 * 4;
  }
  RegsF32<BN> rbf;
  rbf.zmask = 0u;
  auto fetch_w = [&](int chunk, int tap) __attribute__((always_inline)) {
    const int koff = tap * ctot + chunk * BK;
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) rbf.v[i] = *reinterpret_cast<const float4*>(wp[i] + koff);
  };
  auto store_w = [&](int buf) __attribute__((always_inline)) { stage_store<PREC>(&Bs[buf * B_ELEMS], rbf, tid); };

  // lane's base halo rows for its MT output-row fragments
  int arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int r = wm0 + mt * 32 + (lane & 31);
    arow[mt] = (r >> 4) * HWd + (r & 15);
  }
  const int g8 = lane >> 5;

  f32x16 acc[MT][NT];
  acc_zero(acc);
  float4 ra[NA];
  fetch_halo(0, ra);
  fetch_w(0, 0);
  store_halo(ra);
  store_w(0);
  __syncthreads();

  int chunk = 0, tap = 0;
  // MFMAs of one K-tile: weight buffer `buf`, activation rows shifted by the tap offset `toff`
  auto mma_step = [&](int buf, int toff) __attribute__((always_inline)) {
    const lds_t* Bc = &Bs[buf * B_ELEMS];
    if constexpr (PREC == CRAFT_PREC_F32) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        float4 a[MT], bq[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const float4*>(&As[(arow[mt] + toff) * LD + kk * 8 + g8 * 4]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bq[nt] = *reinterpret_cast<const float4*>(&Bc[(wn0 + nt * 32 + (lane & 31)) * LD + kk * 8 + g8 * 4]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].x, bq[nt].x, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].y, bq[nt].y, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].z, bq[nt].z, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].w, bq[nt].w, acc[mt][nt], 0, 0, 0);
          }
      }
    } else if constexpr (PREC == CRAFT_PREC_BF16) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 a[MT], bq[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(&As[(arow[mt] + toff) * LD + kk * 16 + g8 * 8]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bq[nt] = *reinterpret_cast<const bf16x8*>(&Bc[(wn0 + nt * 32 + (lane & 31)) * LD + kk * 16 + g8 * 8]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], bq[nt], acc[mt][nt], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        f16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          ah[mt] = *reinterpret_cast<const f16x8*>(&As[(arow[mt] + toff) * LD + kk * 16 + g8 * 8]);
          if constexpr (PREC == CRAFT_PREC_F16X3) al[mt] = *reinterpret_cast<const f16x8*>(&As[(HR_MAX + arow[mt] + toff) * LD + kk * 16 + g8 * 8]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          bh[nt] = *reinterpret_cast<const f16x8*>(&Bc[(wn0 + nt * 32 + (lane & 31)) * LD + kk * 16 + g8 * 8]);
          if constexpr (PREC == CRAFT_PREC_F16X3) bl[nt] = *reinterpret_cast<const f16x8*>(&Bc[(BN + wn0 + nt * 32 + (lane & 31)) * LD + kk * 16 + g8 * 8]);
        }
        // term-major order: consecutive MFMAs hit different accumulators (no back-to-back dependent chain)
        if constexpr (PREC == CRAFT_PREC_F16X3) {
          if constexpr (CRAFT_X3_TERMS & 1) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
          }
          if constexpr (CRAFT_X3_TERMS & 2) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
          }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
      }
    }

  };
  // Straight-line loop nest (no data-dependent branch around a load: at a control-flow join hipcc equalises the
  // outstanding-load counters of both paths by WAITING, which exposed a full memory latency per chunk):
  // the next chunk's halo and the next weight tile are always fetched (indices clamped at the end; the
  // duplicates are harmless) and always stored.
  int buf = 0;
  for (chunk = 0; chunk < nchunk; ++chunk) {
    const int cn = min(chunk + 1, nchunk - 1);
    fetch_halo(cn, ra);                                   // lands during this chunk's taps
    for (tap = 0; tap < T; ++tap) {
      const bool last_tap = tap + 1 == T;
      fetch_w(last_tap ? cn : chunk, last_tap ? 0 : tap + 1);
      // pin the issue order: without the fences hipcc sinks the weight loads below the MFMAs (to save VGPRs)
      // and then waits on them immediately, exposing an L2 round trip in every K-tile
      __builtin_amdgcn_sched_barrier(0);
      { const int ty_ = tap / KW; mma_step(buf, ty_ * HWd + (tap - ty_ * KW)); }
      __builtin_amdgcn_sched_barrier(0);
      store_w(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
    cur_chunk_stored = cn;
    store_halo(ra);                                       // every wave is past the last tap of this chunk
    __syncthreads();
  }

  // ---- epilogue: GEMM row r of the patch -> token (y0 + r/16, x0 + r%16)
  const int cb = n0 + wn0;
  const int rh4 = 4 * (lane >> 5);
#define BODY(E) conv_epilogue_patch<E, PREC != CRAFT_PREC_F32, MT, NT>(p, acc, wm0, lane, cb, img, y0, x0);
  CONV_EPI_DISPATCH(p, BODY)
#undef BODY
  if (ENC && p.stats) {
    unsigned mlo = 0u, mhi = 0u;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = wm0 + mt * 32 + (e & 3) + 8 * (e >> 2) + rh4;
        const bool ok = (y0 + (r >> 4)) < g.H && (x0 + (r & 15)) < g.W;
        const int bit = mt * 16 + e;
        if (ok) { if (bit < 32) mlo |= 1u << bit; else mhi |= 1u << (bit - 32); }
      }
    conv_col_stats<MT, NT>(p, acc, lane, cb, (long)b, mlo, mhi);
  }
}

template <int PREC, int BN> static int launch_halo_t(const ConvGemmParams& p, hipStream_t s) {
  const bool enc = p.g.in_norm != nullptr || p.stats != nullptr;
  const int ncols = p.epi == CONV_EPI_MENC ? p.cout + 2 : p.cout;
  const int tiles = ((p.g.W + PATCH_W - 1) / PATCH_W) * ((p.g.H + PATCH_H - 1) / PATCH_H) * (p.g.npix / (p.g.H * p.g.W));
  dim3 grid(tiles, (ncols + BN - 1) / BN, 1);
  if (enc) hipLaunchKernelGGL((k_conv_halo<PREC, BN, true>), grid, dim3(NTHREADS), 0, s, p);
  else hipLaunchKernelGGL((k_conv_halo<PREC, BN, false>), grid, dim3(NTHREADS), 0, s, p);
  return (int)hipGetLastError();
}

int launch_conv_halo(const ConvGemmParams& p, int prec, hipStream_t s) {
  if (p.g.KH > 5 || p.g.KW > 5 || (p.g.KH + 7) * (p.g.KW + 15) > (PATCH_H + 4) * PATCH_W) return CRAFT_ERR_UNSUPPORTED;
  // packed weights (craft_pack_weights, fragment order): the weight-fragment kernel (kernels_conv_wf.hip)
  if (p.w_packed) return prec == CRAFT_PREC_F32 ? CRAFT_ERR_ARG : launch_conv_halo_wf(p, prec, s);
  const int ncols = p.epi == CONV_EPI_MENC ? p.cout + 2 : p.cout;
  // BN = 128 only when that still leaves >= 2 blocks per CU's worth of tiles for wide outputs
  int bn = (ncols % 128 == 0 && ncols >= 256) ? 128 : 64;
  if (tuning().halo_bn) bn = tuning().halo_bn == 128 ? (ncols % 128 == 0 ? 128 : 64) : 64;   // developer A/B override
#define GO(PR) do { if (bn == 128) return launch_halo_t<PR, 128>(p, s); else return launch_halo_t<PR, 64>(p, s); } while (0)
  if (prec == CRAFT_PREC_F32) GO(CRAFT_PREC_F32);
  if (prec == CRAFT_PREC_BF16) GO(CRAFT_PREC_BF16);
  if (prec == CRAFT_PREC_F16) GO(CRAFT_PREC_F16);
  if (prec == CRAFT_PREC_F16X3) GO(CRAFT_PREC_F16X3);
#undef GO
  return CRAFT_ERR_ARG;
}

}  // namespace craft
