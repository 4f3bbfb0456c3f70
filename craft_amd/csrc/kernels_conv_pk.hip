// Halo-tile convolution over PLANE-PACKED activations (round 5; include/craft_hip.h: craft_conv2d_pk).
//
// Same tiling and the same weight path as k_conv_halo_wf (kernels_conv_wf.hip): a block owns an 8 x 16 patch of output pixels and
// BN output channels, per 32-channel chunk the (8 + KH - 1) x (16 + KW - 1) halo is staged in LDS once and all KH * KW taps run from
// it, the weights arrive in MFMA fragment order straight from L2.  What differs is the ACTIVATION operand.  k_conv_halo_wf reads fp32
// tokens, so every block converts its halo (fp32 -> hi / lo fp16 planes: 14 VALU + 2 LDS stores per k-half, 48 staging VGPRs, the
// out-of-image selects) inside the K loop, once per column block of the same patch.  Here the input already IS the two fp16 planes
// -- the packed operand format of the backward pass (kernels_gemm_pk.hip: [plane][C/32][rows_p][32] over a zero-padded pixel grid,
// written by craft_pack_operand or directly by a producer's epilogue) -- and the halo is a pure copy:
//   * global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction): no VGPRs, no VALU, no LDS store
//     instructions, and the zero padding of the convolution is the pack's zero border (no masks);
//   * LDS layout [plane][16-byte chunk c of the 32 channels][halo pixel][16 B]: a DMA instruction gathers chunk c of 64 consecutive
//     halo pixels (lane = pixel: the per-lane global offset does the gather), and an A fragment of one 32x32x16 MFMA -- lane
//     (pixel m, k-group h) wants 16 bytes = chunk 2 kk + h of its pixel -- is ONE ds_read_b128 whose 16-lane service groups read 256
//     contiguous bytes (patch_row_perm: a group = 16 consecutive pixels of one patch row): conflict-free for every tap, and the tap /
//     k-half / plane / buffer offsets are all immediates (no address arithmetic in the loop);
//   * every vector-memory instruction of the K loop (weight fragments and DMA pieces) is issued from inline asm with hand-placed
//     s_waitcnt vmcnt(N): hipcc orders every ds_read behind a pending LDS-DMA it knows about, and cannot count one it does not know
//     about.  MEASURED on gfx950: LDS-DMA loads and register loads do NOT retire in order with respect to each other (a count that
//     allowed "the 6 younger DMA pieces" let a weight fragment through unfinished: tools/debug_conv_pk.py, first version), only
//     within their own kind.  So a weight wait never counts DMA pieces -- vmcnt(younger weight loads) is safe whatever the pieces
//     do -- and the DMA of the next chunk is retired by ONE vmcnt(0) per chunk, in front of the chunk's barrier.
// One barrier per 32-channel chunk, as before; the DMA of chunk c + 2 is issued right behind the barrier that retires chunk c.
#include <type_traits>
#include "conv_epilogue.hpp"

namespace craft {

#define CRAFT_LDS __attribute__((address_space(3)))

constexpr int PK_PATCH_H = 8, PK_PATCH_W = 16;

template <int N> __device__ __forceinline__ void pk_interleave() {     // N x { 1 MFMA, 1 LDS read }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  }
}

// PL: planes of the activation pack (2: f16x3 hi / lo, 1: one fp16 plane).  TERMS as k_conv_halo_wf (7: three-term product, 5: weights'
// hi plane only).  BF: bf16 single plane.
template <int PL, int KH, int KW, int WM, int WN, int TERMS, bool BF>
__global__ __launch_bounds__(NTHREADS) void k_conv_pk(ConvPkParams pp) {
  const ConvGemmParams& p = pp.c;
  constexpr int PREC = BF ? CRAFT_PREC_BF16 : (PL == 2 ? CRAFT_PREC_F16X3 : CRAFT_PREC_F16);
  typedef typename FragT<PREC>::t frag_t;
  constexpr int TT = KH * KW, BM = 128, MT = BM / WM / 32, BN = WN * 32;
  static_assert(WM * WN == NTHREADS / 64, "4 waves");
  constexpr int HWd = PK_PATCH_W + KW - 1, HH = PK_PATCH_H + KH - 1, HR = HH * HWd;
  constexpr int HR_MAX = 192, NPC = (HR + 63) / 64;        // halo pixels (rows), 64-row DMA pieces per slab
  static_assert(HR <= HR_MAX, "halo");
  constexpr int SLAB = HR_MAX * 16;                         // bytes of one (plane, chunk) slab
  constexpr int BUF = PL * 4 * SLAB;                        // one halo buffer (32 channels)
  constexpr int NDW = PL * NPC;                             // DMA pieces per wave and chunk
  constexpr int PLB = (PL == 2 && (TERMS & 2)) ? 2 : 1;     // weight planes fetched
  constexpr int BD = 4;                                     // weight ring: k-halves in flight
  static_assert(2 * TT > BD, "ring");
  __shared__ __attribute__((aligned(1024))) unsigned char S[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const ConvGeom& g = p.g;
  const int tiles_x = (g.W + PK_PATCH_W - 1) / PK_PATCH_W, tiles_y = (g.H + PK_PATCH_H - 1) / PK_PATCH_H;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = ty * PK_PATCH_H, x0 = tx * PK_PATCH_W;
  const int n0 = blockIdx.y * BN;
  const int nchunk = (g.c0 + g.c1) / BK;
  const long img = (long)b * g.H * g.W;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * 32;

  // ---- DMA plan.  Wave w copies the (plane, chunk) slabs w * PL .. w * PL + PL - 1, NPC pieces of 64 halo pixels each.
  const ConvPkIn& in = pp.in;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  unsigned rowoff[NPC];                                     // byte offset of this lane's halo pixel within a (plane, group) slab of the pack
#pragma unroll
  for (int j = 0; j < NPC; ++j) {
    const int hr = 64 * j + lane, hc = hr < HR ? hr : 0;
    const int hy = hc / HWd, hx = hc - hy * HWd;
    rowoff[j] = (unsigned)((hy * in.Wp + hx) * 64);
  }
  const unsigned lds0 = (unsigned)(size_t)(CRAFT_LDS unsigned char*)(S);
  const long rowbase = in.row0 + ((long)b * in.Hp + y0) * in.Wp + x0;      // pack row of halo pixel (0, 0)
  auto dma = [&](int chunk, int buf) __attribute__((always_inline)) {
    const int sg = chunk < in.ncg0 ? 0 : 1;
    const int cg = in.cg_off[sg] + chunk - (sg ? in.ncg0 : 0);
    const unsigned long long a = reinterpret_cast<unsigned long long>(in.seg[sg]);
    u32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    d[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    d[2] = __builtin_amdgcn_readfirstlane(in.bytes[sg]);                  // raw buffer: an offset beyond the pack reads zeros
    d[3] = 0x00020000u;
    const unsigned sb = (unsigned)(rowbase * 64) + (unsigned)cg * in.cgs[sg];
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      const int combo = wave * PL + i, pl = combo >> 2, c = combo & 3;
      const unsigned so = __builtin_amdgcn_readfirstlane(sb + (unsigned)pl * in.plane[sg] + (unsigned)c * 16u);
#pragma unroll
      for (int j = 0; j < NPC; ++j) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * BUF + combo * SLAB + j * 1024));
        const unsigned voff = rowoff[j] + so;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(dst), "s"(d) : "memory");
      }
    }
  };

  // ---- weight fragments [kt][nb][pl][kk][lane][8] (craft_pack_weights), L2 -> MFMA registers: scalar base + lane offset
  const int NBtot = (p.cout + 31) / 32;
  const int nb = min((n0 + wn0) / 32, NBtot - 1);
  const unsigned char* const wbase = reinterpret_cast<const unsigned char*>(p.W) + (long)nb * (PL * 2048);
  const long kt_stride = (long)NBtot * (PL * 2048);
  const unsigned wlane = lane * 16;
  frag_t bq[PLB][BD];
  auto fetch_b = [&](int kt, int kk, int slot) __attribute__((always_inline)) {
    const unsigned char* q = wbase + kt * kt_stride + kk * 1024;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bq[0][slot]) : "v"(wlane), "s"(q) : "memory");
    if constexpr (PLB == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(bq[1][slot]) : "v"(wlane), "s"(q) : "memory");
  };
  auto wait_b = [&bq](int slot, auto n_c) __attribute__((always_inline)) {
    constexpr int NW = decltype(n_c)::value;
    if constexpr (PLB == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(bq[0][slot]), "+v"(bq[1][slot]) : "n"(NW) : "memory");
    else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(bq[0][slot]) : "n"(NW) : "memory");
  };
  auto wait_all = [&bq]() __attribute__((always_inline)) {      // every request of this wave has landed (weights of the whole ring + DMA pieces)
    static_assert(BD == 4, "ring");
    if constexpr (PLB == 2)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[0][2]), "+v"(bq[0][3]), "+v"(bq[1][0]), "+v"(bq[1][1]), "+v"(bq[1][2]),
                   "+v"(bq[1][3]) : : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[0][2]), "+v"(bq[0][3]) : : "memory");
  };

  // ---- A fragments: lane (pixel m = lane & 31, k-group h = lane >> 5) reads 16 bytes of slab (plane, 2 kk + h) at its halo pixel
  unsigned abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int r = wm0 + mt * 32 + patch_row_perm(lane & 31);
    abase[mt] = (unsigned)((lane >> 5) * SLAB + ((r >> 4) * HWd + (r & 15)) * 16);
  }
  auto read_a = [&](int buf, int tap, int kk, frag_t (&h)[MT], frag_t (&l)[MT]) __attribute__((always_inline)) {
    const int shift = (tap / KW) * HWd + (tap % KW);
    const int off = buf * BUF + kk * 2 * SLAB + shift * 16;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      h[mt] = *reinterpret_cast<const frag_t*>(&S[abase[mt] + off]);
      if constexpr (PL == 2) l[mt] = *reinterpret_cast<const frag_t*>(&S[abase[mt] + off + 4 * SLAB]);
    }
  };

  f32x16 acc[MT][1];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[mt][0][e] = 0.f;
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
  constexpr int NMMA = MT * (PL == 2 ? ((TERMS & 1) + ((TERMS >> 1) & 1) + 1) : 1);

  // ---- prologue: halo chunks 0 and 1, weight k-halves 0 .. BD-1
  dma(0, 0);
  dma(min(1, nchunk - 1), 1);
#pragma unroll
  for (int q = 0; q < BD; ++q) fetch_b(min(q / 2, TT - 1) * nchunk, q & 1, q);
  frag_t a0h[MT], a0l[MT], a1h[MT], a1l[MT];
  // Per k-half h = 2 tap + kk of a chunk: wait for its weight fragments (vmcnt = the BD-1 younger weight k-halves; the first BD-1
  // k-halves of a chunk need no wait: the previous chunk's last k-half drained the queue), A fragments one k-half ahead from the other
  // register set, MFMAs, then the weight request BD k-halves ahead into the slot just consumed.
  auto khalf = [&](int chunk, auto hb_c, auto ph_c, auto h_c, auto first_c) __attribute__((always_inline)) {
    constexpr int hb = decltype(hb_c)::value, PHASE = decltype(ph_c)::value, h = decltype(h_c)::value;
    constexpr bool FIRST = decltype(first_c)::value;        // very first k-half of the kernel: its A fragments are read here
    constexpr int tap = h / 2, kk = h & 1;
    constexpr bool last = h == 2 * TT - 1;
    constexpr int slot = (PHASE + h) % BD;
    constexpr int NW = (BD - 1) * PLB;
    const int cn = min(chunk + 1, nchunk - 1);
    if constexpr (FIRST) {
      wait_all();                                            // the halo of chunk 0 (and 1) and the first BD weight k-halves have landed
      asm volatile("s_barrier" ::: "memory");
      read_a(0, 0, 0, a0h, a0l);
    }
    if constexpr (!last) {
      if constexpr (h >= BD - 1) wait_b(slot, std::integral_constant<int, NW>());
      if constexpr (kk == 0) read_a(hb, tap, 1, a1h, a1l); else read_a(hb, tap + 1, 0, a0h, a0l);
    } else {
      // last k-half of the chunk: drain this wave's queue (its weights, the ring, and the halo of chunk + 1 issued a whole chunk ago);
      // every wave's reads of this chunk's buffer have returned (lgkmcnt(0)) before the barrier, behind which the buffer is handed to
      // the DMA of chunk + 2
      wait_all();
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      read_a(hb ^ 1, 0, 0, a0h, a0l);                        // (after the last chunk: a harmless re-read)
      dma(min(chunk + 2, nchunk - 1), hb);
    }
    if constexpr (kk == 0) mma_half(a0h, a0l, slot); else mma_half(a1h, a1l, slot);
    {
      constexpr int hf = h + BD;
      if constexpr (hf < 2 * TT) fetch_b((hf / 2) * nchunk + chunk, hf & 1, slot);
      else fetch_b(((hf - 2 * TT) / 2) * nchunk + cn, hf & 1, slot);
    }
    pk_interleave<NMMA>();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto chunk_body = [&](int chunk, auto hb_c, auto ph_c, auto first_c) __attribute__((always_inline)) {
    auto run = [&](auto self, auto h_c, auto f_c) __attribute__((always_inline)) -> void {
      constexpr int h = decltype(h_c)::value;
      if constexpr (h < 2 * TT) {
        khalf(chunk, hb_c, ph_c, h_c, f_c);
        self(self, std::integral_constant<int, h + 1>(), std::false_type());
      }
    };
    run(run, std::integral_constant<int, 0>(), first_c);
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  typedef std::integral_constant<int, (2 * TT) % BD> P1;
  int chunk = 0;
  // peeled first chunk (its first k-half opens with the prologue's wait + barrier), then pairs of chunks (buffer / ring phase parity)
  chunk_body(0, I0(), I0(), std::true_type());
  chunk = 1;
  for (; chunk + 1 < nchunk; chunk += 2) {
    chunk_body(chunk, I1(), P1(), std::false_type());
    chunk_body(chunk + 1, I0(), I0(), std::false_type());
  }
  if (chunk < nchunk) chunk_body(chunk, I1(), P1(), std::false_type());
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the surplus requests of the tail (an LDS-DMA must not outlive the block)

  // ---- epilogue: GEMM row r of the patch -> token (y0 + r/16, x0 + r%16)
  const int cb = n0 + wn0;
#define BODY(E) conv_epilogue_patch<E, true, MT, 1, true>(p, acc, wm0, lane, cb, img, y0, x0);
  CONV_EPI_DISPATCH(p, BODY)
#undef BODY
}

template <int PL, int KH, int KW, int TERMS, bool BF> static int launch_pk_t(const ConvPkParams& pp, hipStream_t s) {
  const ConvGemmParams& p = pp.c;
  const int ncols = p.epi == CONV_EPI_MENC ? p.cout + 2 : p.cout;
  const int bn = (ncols % 128 == 0 && ncols >= 256) ? 128 : 64;
  const int tiles = ((p.g.W + PK_PATCH_W - 1) / PK_PATCH_W) * ((p.g.H + PK_PATCH_H - 1) / PK_PATCH_H) * (p.g.npix / (p.g.H * p.g.W));
  dim3 grid(tiles, (ncols + bn - 1) / bn, 1);
  if (bn == 128) hipLaunchKernelGGL((k_conv_pk<PL, KH, KW, 1, 4, TERMS, BF>), grid, dim3(NTHREADS), 0, s, pp);
  else hipLaunchKernelGGL((k_conv_pk<PL, KH, KW, 2, 2, TERMS, BF>), grid, dim3(NTHREADS), 0, s, pp);
  return (int)hipGetLastError();
}

// stride-1 KH x KW in {1x5, 5x1, 3x3} over packed activations; weights from craft_pack_weights / craft_pack_conv_weights
int launch_conv_pk(const ConvPkParams& pp, int prec, hipStream_t s) {
  const ConvGemmParams& p = pp.c;
  const int KH = p.g.KH, KW = p.g.KW;
  if ((p.g.c0 + p.g.c1) % BK || p.g.c0 % BK || !p.w_packed) return CRAFT_ERR_ARG;
  if (p.g.in_norm || p.stats) return CRAFT_ERR_UNSUPPORTED;
#define GO(PL, TERMS, BF) do { \
    if (KH == 1 && KW == 5) return launch_pk_t<PL, 1, 5, TERMS, BF>(pp, s); \
    if (KH == 5 && KW == 1) return launch_pk_t<PL, 5, 1, TERMS, BF>(pp, s); \
    if (KH == 3 && KW == 3) return launch_pk_t<PL, 3, 3, TERMS, BF>(pp, s); \
    return CRAFT_ERR_UNSUPPORTED; } while (0)
  if (prec == CRAFT_PREC_F16X3) { if (p.w16) GO(2, 5, false); else GO(2, 7, false); }
  if (prec == CRAFT_PREC_F16) GO(1, 7, false);
  if (prec == CRAFT_PREC_BF16) GO(1, 7, true);
#undef GO
  return CRAFT_ERR_UNSUPPORTED;
}

}  // namespace craft
