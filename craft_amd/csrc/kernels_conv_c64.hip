// 3x3 / stride-1 NHWC convolution for 64 input and 64 output channels (BasicEncoder layer1, extractor.py:151-153),
// packed 16-bit weights (bf16 / fp16 / f16x3, fragment order) -- the shape where k_conv_halo_wf is weakest: K = 576
// is 18 K-tiles, so a block that owns one 128-pixel patch spends most of its life in prologue / epilogue / halo
// round trips (PMC on the box: 2000 VALU + 1700 SALU instructions per wave for 257 MFMAs, two exposed HBM latencies).
//
//   * PERSISTENT blocks: grid = 2 per CU, each walks patches p = blockIdx.x, += gridDim.x.  The halo of the NEXT patch
//     is requested (12 float4 per thread) before the K loop of the current one and lands behind its 216 MFMAs.
//   * the WHOLE 64-channel halo (10 x 18 pixels) is staged at once: no channel chunks, no barrier inside the K loop
//     (two per patch, around the re-staging of the single 54 KB buffer; the second block of the CU fills the gap).
//   * the K loop is fully unrolled (9 taps x 4 k-steps of 16 channels): every LDS / weight offset is an immediate;
//     A fragments are read one k-step ahead, weight fragments three k-steps ahead and the ring wraps around to the
//     next patch (same weights), so the weight pipeline is primed once per block, not once per patch.
// Epilogue: bias (+ReLU) and the per-(image, channel) statistics of a lazy InstanceNorm, as k_conv_halo_wf<ENC>.
// Measured (8 x 224x512 images): 240 us vs 330 us for k_conv_halo_wf; by ablation 153 us K loop + halo loads, 53 us output
// stores, 20 us statistics, 18 us staging -- the phases ADD: on gfx9-family ISAs stores count in vmcnt, so the first
// weight-fragment wait of the next patch drains the previous patch's stores, and two waves per SIMD do not cover it.
#include <cstdlib>
#include "conv_epilogue.hpp"

namespace craft {

// TERMS: as in k_conv_halo_wf (7 = three-term f16x3 product, 5 = the weights' hi plane only: CRAFT_CONV_W16, input-gradient convolutions)
#ifdef CRAFT_C64_LINEAR            // developer A/B (tools/build_variant.py): the linear GEMM-row -> pixel assignment of rounds 2-4
#define C64_ROW(l) (l)
constexpr bool C64_PERM = false;
#else
#define C64_ROW(l) patch_row_perm(l)
constexpr bool C64_PERM = true;
#endif
template <int PREC, int TERMS = CRAFT_X3_TERMS>
__global__ __launch_bounds__(NTHREADS) void k_conv3x3_c64(ConvGemmParams p) {
  typedef typename PrecT<PREC>::lds_t lds_t;
  typedef typename FragT<PREC>::t frag_t;
  constexpr int PL = Planes<PREC>::N;
  constexpr int CIN = 64, LD = CIN + 8, PH = 8, PW = 16, HWd = PW + 2, HR = (PH + 2) * HWd, HRP = 192;   // 180 rows used
  constexpr int WM = 2, WN = 2, MT = 2;
  constexpr int NSTEP = 36;                    // 9 taps x 4 k-steps of 16 channels
  constexpr int NA = HRP / 32;                 // halo rows per thread and 32-channel half
  constexpr int BDEPTH = 3;                    // weight fragments requested this many k-steps ahead
  __shared__ __attribute__((aligned(16))) lds_t As[PL * HRP * LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const ConvGeom& g = p.g;
  const int tiles_x = (g.W + PW - 1) / PW, tiles_y = (g.H + PH - 1) / PH;
  const int per_img = tiles_x * tiles_y, npatch = per_img * (g.npix / (g.H * g.W));
  const int wm0 = (wave / WN) * 64, wn0 = (wave % WN) * 32;
  const int c4 = tid & 7, r0 = tid >> 3;

  // weight fragments of this wave's 32 output channels: [kt = tap*2 + half][nb][pl][kk][lane][8]
  const int NBtot = (p.cout + 31) / 32;
  const uint16_t* wb = reinterpret_cast<const uint16_t*>(p.W) + (long)min(wn0 / 32, NBtot - 1) * (PL * 1024) + lane * 8;
  const long kt_stride = (long)NBtot * (PL * 1024);
  auto fetch_b = [&](int step, frag_t (&dst)[PL]) __attribute__((always_inline)) {
    const int tap = step >> 2, c16 = step & 3;
    const uint16_t* q = wb + (long)(tap * 2 + (c16 >> 1)) * kt_stride + (c16 & 1) * 512;
#pragma unroll
    for (int pl = 0; pl < PL; ++pl)
      if (pl == 0 || (TERMS & 2)) dst[pl] = *reinterpret_cast<const frag_t*>(q + pl * 1024);
  };
  // lane's base halo element offsets for its MT output-row fragments
  int arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    // (patch_row_perm: every 16-lane service group of ds_read_b128 reads 16 consecutive pixels of ONE patch row -- 144-byte rows, 16 of
    // them cover the 64 banks exactly once.  Round 5 PMC of the linear assignment: SQ_LDS_BANK_CONFLICT = 0.50 of SQ_LDS_IDX_ACTIVE.)
    const int r = wm0 + mt * 32 + C64_ROW(lane & 31);
    arow[mt] = ((r >> 4) * HWd + (r & 15)) * LD + (lane >> 5) * 8;
  }
  auto read_a = [&](int step, frag_t (&h)[MT], frag_t (&l)[MT]) __attribute__((always_inline)) {
    const int tap = step >> 2, c16 = step & 3;
    const int off = ((tap / 3) * HWd + (tap % 3)) * LD + c16 * 16;        // compile-time after unrolling
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      h[mt] = *reinterpret_cast<const frag_t*>(&As[arow[mt] + off]);
      if constexpr (PL == 2) l[mt] = *reinterpret_cast<const frag_t*>(&As[HRP * LD + arow[mt] + off]);
    }
  };

  // ---- halo of patch `pt`: thread (row r0 + 32 i, channels half*32 + c4*4 .. +4); -1: outside the image / halo
  int hpix[NA];
  long img = 0;
  int y0 = 0, x0 = 0, bimg = 0;
  auto locate = [&](int pt, int (&hp)[NA], long& im, int& yy0, int& xx0, int& bb) __attribute__((always_inline)) {
    bb = pt / per_img;
    const int rem = pt - bb * per_img;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    yy0 = ty * PH; xx0 = tx * PW;
    im = (long)bb * g.H * g.W;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int hr = r0 + 32 * i;
      const int hy = (hr * 3641) >> 16, hx = hr - hy * HWd;          // hr / 18 for hr < 192
      const int y = yy0 - 1 + hy, x = xx0 - 1 + hx;
      hp[i] = (hr < HR && y >= 0 && y < g.H && x >= 0 && x < g.W) ? y * g.W + x : -1;
    }
  };
  float4 ra[2][NA];
  auto fetch_halo = [&](const int (&hp)[NA], long im) __attribute__((always_inline)) {
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int i = 0; i < NA; ++i)
        ra[half][i] = *reinterpret_cast<const float4*>(g.seg0 + (im + max(hp[i], 0)) * g.ld0 + half * 32 + c4 * 4);
  };
  auto store_halo = [&](const int (&hp)[NA], int bb) __attribute__((always_inline)) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), rs = make_float4(1.f, 1.f, 1.f, 1.f);
      if (g.in_norm) {            // (mean, rstd) of this thread's 4 input channels, image bb: relu((x - mean) * rstd)
        const float* t = g.in_norm + ((long)bb * CIN + half * 32 + c4 * 4) * 2;
        const float4 t0 = *reinterpret_cast<const float4*>(t), t1 = *reinterpret_cast<const float4*>(t + 4);
        mu = make_float4(t0.x, t0.z, t1.x, t1.z);
        rs = make_float4(t0.y, t0.w, t1.y, t1.w);
      }
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int row = r0 + 32 * i;
        const bool ok = hp[i] >= 0;
        float4 v = ra[half][i];
        if (g.in_norm) {
          v.x = fmaxf((v.x - mu.x) * rs.x, 0.f); v.y = fmaxf((v.y - mu.y) * rs.y, 0.f);
          v.z = fmaxf((v.z - mu.z) * rs.z, 0.f); v.w = fmaxf((v.w - mu.w) * rs.w, 0.f);
        }
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        lds_t* d = &As[row * LD + half * 32 + c4 * 4];
        if constexpr (PREC == CRAFT_PREC_BF16) {
          bf16x4 h;
          h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
          *reinterpret_cast<bf16x4*>(d) = h;
        } else {
          if constexpr (PREC == CRAFT_PREC_F16X3) {
            f16x4 h, l;
            split_f16x3(v, h, l);
            *reinterpret_cast<f16x4*>(d) = h;
            *reinterpret_cast<f16x4*>(d + HRP * LD) = l;
          } else {
            f16x4 h;
            h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
            *reinterpret_cast<f16x4*>(d) = h;
          }
        }
      }
    }
  };

  int pt = blockIdx.x;
  if (pt >= npatch) return;
  frag_t bfr[BDEPTH][PL];
#pragma unroll
  for (int s = 0; s < BDEPTH; ++s) fetch_b(s, bfr[s]);
  locate(pt, hpix, img, y0, x0, bimg);
  fetch_halo(hpix, img);

  for (; pt < npatch; pt += gridDim.x) {
    __syncthreads();                                   // every wave is done reading the previous patch's halo
    store_halo(hpix, bimg);
    __syncthreads();
    // next patch: locate + request its halo now, it arrives behind this patch's MFMAs
    const int ptn = pt + gridDim.x < npatch ? pt + gridDim.x : pt;
    int hpix_n[NA]; long img_n; int y0n, x0n, bn;
    locate(ptn, hpix_n, img_n, y0n, x0n, bn);
    fetch_halo(hpix_n, img_n);

    f32x16 acc[MT][1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][0][e] = 0.f;
    frag_t ah[2][MT], al[2][MT];
    read_a(0, ah[0], al[0]);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s + 1 < NSTEP) read_a(s + 1, ah[(s + 1) & 1], al[(s + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      frag_t (&h)[MT] = ah[s & 1];
      frag_t (&l)[MT] = al[s & 1];
      frag_t (&bw)[PL] = bfr[s % BDEPTH];
      if constexpr (PL == 2) {
        if constexpr (TERMS & 1) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt][0] = mfma16<PREC>(l[mt], bw[0], acc[mt][0]);
        }
        if constexpr (TERMS & 2) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt][0] = mfma16<PREC>(h[mt], bw[1], acc[mt][0]);
        }
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][0] = mfma16<PREC>(h[mt], bw[0], acc[mt][0]);
      fetch_b((s + BDEPTH) % NSTEP, bfr[s % BDEPTH]);      // wraps into the next patch: same weights
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue of patch pt
    const int cb = blockIdx.y * 64 + wn0;
    const int rh4 = 4 * (lane >> 5);
    conv_epilogue_patch<CONV_EPI_BIAS_ACT, true, MT, 1, C64_PERM>(p, acc, wm0, lane, cb, img, y0, x0);
    if (p.stats) {
      unsigned mlo = ~0u, mhi = ~0u;
      if (y0 + PH > g.H || x0 + PW > g.W) {
        mlo = 0u; mhi = 0u;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int r = wm0 + mt * 32 + C64_ROW((e & 3) + 8 * (e >> 2) + rh4);
            const bool ok = (y0 + (r >> 4)) < g.H && (x0 + (r & 15)) < g.W;
            const int bit = mt * 16 + e;
            if (ok) { if (bit < 32) mlo |= 1u << bit; else mhi |= 1u << (bit - 32); }
          }
      }
      conv_col_stats<MT, 1>(p, acc, lane, cb, (long)bimg, mlo, mhi);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) hpix[i] = hpix_n[i];
    img = img_n; y0 = y0n; x0 = x0n; bimg = bn;
  }
}

// packed weights, 3x3, stride 1, cin == 64 in one segment, cout <= 64, BIAS_ACT epilogue: called by launch_conv_halo_wf
bool conv3x3_c64_applies(const ConvGemmParams& p) {
  return p.g.KH == 3 && p.g.KW == 3 && p.g.c0 == 64 && p.g.c1 == 0 && p.cout <= 64 && p.epi == CONV_EPI_BIAS_ACT &&
         p.bias_field == nullptr && p.g.stride == 1;
}

int launch_conv3x3_c64(const ConvGemmParams& p, int prec, hipStream_t s) {
  const int tiles = ((p.g.W + 15) / 16) * ((p.g.H + 7) / 8) * (p.g.npix / (p.g.H * p.g.W));
  dim3 grid(tiles < 512 ? tiles : 512, 1, 1);            // persistent: 2 blocks per CU
  if (prec == CRAFT_PREC_BF16) hipLaunchKernelGGL((k_conv3x3_c64<CRAFT_PREC_BF16>), grid, dim3(NTHREADS), 0, s, p);
  else if (prec == CRAFT_PREC_F16) hipLaunchKernelGGL((k_conv3x3_c64<CRAFT_PREC_F16>), grid, dim3(NTHREADS), 0, s, p);
  else if (prec == CRAFT_PREC_F16X3 && p.w16) hipLaunchKernelGGL((k_conv3x3_c64<CRAFT_PREC_F16X3, 5>), grid, dim3(NTHREADS), 0, s, p);
  else if (prec == CRAFT_PREC_F16X3) hipLaunchKernelGGL((k_conv3x3_c64<CRAFT_PREC_F16X3>), grid, dim3(NTHREADS), 0, s, p);
  else return CRAFT_ERR_ARG;
  return (int)hipGetLastError();
}

}  // namespace craft
