// Flash-style fused attention of a feature transformer (CrossAttFeatTrans + ExpandedFeatTrans up to the mode pooling,
// setrans.py:507-557 and :364-410):
//     O[b][m][i][:] = sum_j softmax_j( clamp?(Q_m(i).K_m(j) * scale) + pw*pb(i,j) + mask ) * V_m[j][:]
// The N x N probabilities of such a layer are used exactly once, so they never exist in memory: per 32-key tile a wave
// computes the scores of its 32 queries (f16x3 or fp16 MFMAs), updates the running row maximum / sum, and feeds the
// un-normalised probabilities straight back into the MFMAs of P.V as a 16-bit operand (online softmax).
//
// Layouts (all produced on the device, see below):
//   * Q and K are pre-split into fp16 planes and stored in MFMA FRAGMENT order by k_pack_qk:
//       [z = b*M + m][32-row block][k-step][plane][lane][8]   (lane -> row = lane & 31, k = 16*kstep + 8*(lane >> 5) + j)
//     Q is pre-multiplied by scale*log2(e): the softmax runs in the base-2 domain like k_attn_probs.
//   * "Swapped" score product S^T = K.Q^T: a lane owns ONE query (column lane & 31) and 16 of the 32 keys of the tile,
//     accumulator register i <-> key 8*(i >> 2) + 4*(lane >> 5) + (i & 3).  Registers 8*h .. 8*h+7 converted to fp16 are
//     directly the B operand of MFMA h of O^T += V^T.P^T, PROVIDED the A operand (V^T) enumerates the keys of a
//     16-key group in the same order: craft_linear_t(frag_rows = Dv | CRAFT_FRAG_ACC_ORDER) stores V^T that way.
//     No shuffle, no LDS round trip for P.
//   * K / V^T tiles (8 + 16 KB at d = 64, Dv = 256) are shared by the 8 waves of a block through LDS (double buffer,
//     fragment order = conflict-free ds_read_b128); global -> registers one tile ahead, registers -> LDS after the
//     MFMAs, one barrier per tile.
// grid ceil(N / 256) * B*M (XCD-aware mapping, see the kernel), 512 threads: wave w owns queries [256*bx + 32*w, +32).
#include <type_traits>
#include "launch.hpp"

namespace craft {

__device__ __forceinline__ float xhalf_max(float v) {       // max over the two half-waves (lanes l and l ^ 32)
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

constexpr int FLASH_TABW = 33;          // bias / mask table width: 2 * 15 + 3
constexpr float FLASH_RESCALE_THR = 8.f;

// ---------------------------------------------------------------------------------------------------------------------
// k_flash_attn2 (round 5): the same arithmetic as k_flash_attn, software-pipelined INSIDE the wave.
//
// k_flash_attn runs, per 32-key tile and wave, scores (12 MFMAs on one accumulator: a dependent chain, 764 cycles) -> softmax (VALU only,
// 1 065 cycles) -> P.V (16 MFMAs, 615) -> staging + barrier (695), strictly one after the other; the block barrier keeps the two waves of
// a SIMD in the same phase, so the matrix pipe idles through every softmax (PMC: 43 % busy).  VALU work overlaps MFMAs only when both come
// from the SAME wave (tools/ubench/mfma_valu_prio.hip), hence:
//   phase A of iteration t:  the score MFMAs of tile t+1   ||  the softmax VALU of tile t   (different accumulators: s_next / s_cur)
//   phase B of iteration t:  the P.V MFMAs of tile t        ||  bias / mask VALU of tile t+1 (only for tiles that touch the window)
// The dependent score chain of one wave (one MFMA per 64 cycles) leaves every other slot to its SIMD partner, which is in the same phase.
//   * K / V^T tiles travel global -> LDS by LDS-DMA (buffer_load ... lds: no staging registers -- they pay for the second score
//     accumulator -- no LDS store instructions), three stages: tile t+1's K and tile t's V^T are read while tile t+2 lands; one
//     s_waitcnt vmcnt(0) + barrier per tile (the DMA pieces are the only vector-memory traffic of the loop).
//   * registers: o 128 + Q 32 + s_cur 16 + s_next 16 + P' 8 + K fragments 16 + V^T fragments 16 = 232 of 256.
// ---------------------------------------------------------------------------------------------------------------------
#define CRAFT_LDS __attribute__((address_space(3)))
template <int D, int DV, int PL>
__global__ __launch_bounds__(512) void k_flash_attn2(FlashParams p) {
  constexpr int KS = D / 16, NB = DV / 32;
  constexpr int KT_H = KS * PL * 512, VT_H = NB * 2 * 512, TILE_H = KT_H + VT_H;
  constexpr int NPIECE = TILE_H / 512;            // 1 KiB DMA pieces per tile
  constexpr int PPW = (NPIECE + 7) / 8;
  constexpr int NS = 3;
  __shared__ __attribute__((aligned(1024))) uint16_t St[NS * TILE_H];
  __shared__ float s_tab[FLASH_TABW * FLASH_TABW];
  __shared__ unsigned s_need[2048 / 32 + 16];       // bit t: tile t takes the bias / mask / ragged path (N < 65536: at most 2048 tiles; the fill loop below
                                                    // writes whole 512-tile rounds up to tile nkt: 16 words past word nkt / 32 at nkt = 2048)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqx = (p.N + 255) / 256, total = nqx * p.B * p.M;
  const int v = xcd_chunk(blockIdx.x, total);
  const int z = v / nqx, bx = v - z * nqx, b = z / p.M, m = z - b * p.M;
  const int N = p.N, W8 = p.W8, R = p.R;
  const int q0 = bx * 256;
  const int qb = bx * 8 + wave;
  const int qidx = qb * 32 + (lane & 31);
  const int hh = lane >> 5;

  constexpr float LOG2E = 1.4426950408889634f;
  const bool clamp = p.clamp_ord != nullptr && ord2f(*p.clamp_ord) > CRAFT_ATTN_CLIP;
  const int mr = p.mask_radius > 0 ? p.mask_radius : 0;
  const int Re = max(p.pos_tab ? R : 0, mr), TW = 2 * Re + 3;
  for (int i = tid; i < TW * TW; i += 512) {
    const int dh = i / TW - Re - 1, dw = i - (i / TW) * TW - Re - 1;
    float tv = 0.f;
    if (p.pos_tab && abs(dh) <= R && abs(dw) <= R) tv = p.pos_tab[(dh + R) * (2 * R + 1) + dw + R] * (p.pos_w * LOG2E);
    if (mr > 0 && max(abs(dh), abs(dw)) > mr) tv += -1e9f;
    s_tab[i] = tv;
  }
  const float clipv = clamp ? CRAFT_ATTN_CLIP * LOG2E : 3.0e38f;
  const int qc = min(qidx, N - 1);
  const int h1 = qc / W8, w1 = qc - h1 * W8;
  const int ch = Re + 1 - h1, cw = Re + 1 - w1;
  const unsigned umax = 2 * Re + 2;
  const int q_hmin = q0 / W8, q_hmax = min(q0 + 255, N - 1) / W8;
  const bool always_tab = clamp || mr > 0;

  // ---- Q fragments of this wave (resident)
  f16x8 qf[KS][PL];
  {
    const uint16_t* qs = p.Qf + ((long)z * p.nqb + qb) * (KS * PL * 512) + lane * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int pl = 0; pl < PL; ++pl) qf[ks][pl] = *reinterpret_cast<const f16x8*>(qs + (ks * PL + pl) * 512);
  }

  // ---- LDS-DMA of tile t into stage st: piece id = wave * PPW + i; pieces [0, KT_H / 512) are the K tile, the rest V^T
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const int nkt = p.nkt;
  const unsigned lds0 = (unsigned)(size_t)(CRAFT_LDS uint16_t*)(St);
  const unsigned voff = lane * 16;
  unsigned klo, khi, vlo, vhi;
  {
    const unsigned long long ka = reinterpret_cast<unsigned long long>(p.Kf + (long)z * p.nkb * KT_H);
    const unsigned long long va_ = reinterpret_cast<unsigned long long>(p.Vf + (long)b * p.v_bs + (long)m * p.v_ms);
    klo = __builtin_amdgcn_readfirstlane((unsigned)ka); khi = __builtin_amdgcn_readfirstlane((unsigned)(ka >> 32) & 0xffffu);
    vlo = __builtin_amdgcn_readfirstlane((unsigned)va_); vhi = __builtin_amdgcn_readfirstlane((unsigned)(va_ >> 32) & 0xffffu);
  }
  auto dma = [&](int t, int st) __attribute__((always_inline)) {
    const int tc = min(t, nkt - 1);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int id = wave * PPW + i;
      if (NPIECE % 8 != 0 && id >= NPIECE) break;       // (wave-uniform)
      const bool isk = id < KT_H / 512;
      u32x4 d;
      d[0] = isk ? klo : vlo; d[1] = isk ? khi : vhi; d[2] = 0xffffffffu; d[3] = 0x00020000u;
      const unsigned soff = __builtin_amdgcn_readfirstlane(isk ? (unsigned)tc * (KT_H * 2) + (unsigned)id * 1024u
                                                                : (unsigned)tc * (VT_H * 2) + (unsigned)(id - KT_H / 512) * 1024u);
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(st * TILE_H * 2 + id * 1024));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff), "s"(dst), "s"(d), "s"(soff) : "memory");
    }
  };

  // scores of the tile in stage st: S^T = K . Q^T into s (a dependent MFMA chain; K fragments one k-step ahead)
  auto scores = [&](int st, f32x16& s) __attribute__((always_inline)) {
    const uint16_t* Kt = &St[st * TILE_H + lane * 8];
    f16x8 kf[2][PL];
#pragma unroll
    for (int pl = 0; pl < PL; ++pl) kf[0][pl] = *reinterpret_cast<const f16x8*>(Kt + pl * 512);
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) {
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) kf[(ks + 1) & 1][pl] = *reinterpret_cast<const f16x8*>(Kt + ((ks + 1) * PL + pl) * 512);
      }
      if constexpr (PL == 2) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks & 1][1], qf[ks][0], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks & 1][0], qf[ks][1], s, 0, 0, 0);
      }
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks & 1][0], qf[ks][0], s, 0, 0, 0);
    }
  };
  // bias window / mask / clamp / ragged tail of tile t applied to its raw scores
  auto bias = [&](int t, f32x16& s) __attribute__((always_inline)) {
    const int j0 = t * 32;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = j0 + 8 * (e >> 2) + 4 * hh + (e & 3);
      const int kh = (int)__umulhi((unsigned)key, p.w8_magic), kw = key - kh * W8;
      const unsigned u = min((unsigned)(kh + ch), umax), w_ = min((unsigned)(kw + cw), umax);
      float sv = __builtin_amdgcn_fmed3f(s[e], -clipv, clipv) + s_tab[u * TW + w_];
      if (key >= N) sv = -INFINITY;
      s[e] = sv;
    }
  };
  auto needs_bias = [&](int t) __attribute__((always_inline)) {
    const int j0 = t * 32;
    const int k_hmin = j0 / W8, k_hmax = min(j0 + 31, N - 1) / W8;
    return always_tab || (p.pos_tab != nullptr && k_hmax >= q_hmin - R && k_hmin <= q_hmax + R) || (j0 + 32 > N);
  };

  // which tiles need the table: decided once per block (thread i <-> tile i; the per-tile form cost 45 scalar instructions -- two
  // integer divisions -- in every iteration of a loop that is bound by instruction issue)
  for (int i0 = 0; i0 < nkt + 1; i0 += 512) {
    const int ti = i0 + tid;
    const bool f = ti < nkt && needs_bias(ti);
    const unsigned long long bal = __ballot(f);
    if (lane == 0) { s_need[(i0 >> 5) + 2 * wave] = (unsigned)bal; s_need[(i0 >> 5) + 2 * wave + 1] = (unsigned)(bal >> 32); }
  }

  f32x16 o[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[nb][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // ---- prologue: tiles 0, 1 (and 2) requested; S_0 computed without overlap
  dma(0, 0);
  dma(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                   // (also publishes s_tab)
  dma(2, 2);
  f32x16 s_a, s_b;
  scores(0, s_a);
  if (needs_bias(0)) bias(0, s_a);

  int st_cur = 0, st_nxt = 1, st_far = 2;            // stages of tiles t, t+1, t+2
  // one iteration: tile t's scores are in s_cur, tile t+1's are formed in s_next (the two trade places every iteration: the loop is
  // unrolled by two instead of copying 16 registers per tile)
  auto iter = [&](int t, f32x16& s_cur, f32x16& s_next) __attribute__((always_inline)) {
    // (scalar; decided here so that no branch separates phase A from phase B)
    const bool nb_ = (__builtin_amdgcn_readfirstlane(s_need[(t + 1) >> 5]) >> ((t + 1) & 31)) & 1u;
    // ---- phase A: S_{t+1} (MFMA chain)  ||  softmax of S_t (VALU).  (Tried: the score product on two accumulators, even / odd k-steps
    // alternating -- no dependent chain, 250 VGPRs: 0.87 ms against 0.85 for this form, profiles/r5/flash_experiments.txt.)
    const uint16_t* Kt = &St[st_nxt * TILE_H + lane * 8];
    f16x8 kf[2][PL];
#pragma unroll
    for (int pl = 0; pl < PL; ++pl) kf[0][pl] = *reinterpret_cast<const f16x8*>(Kt + pl * 512);
#pragma unroll
    for (int pl = 0; pl < PL; ++pl) kf[1][pl] = *reinterpret_cast<const f16x8*>(Kt + (PL + pl) * 512);
#pragma unroll
    for (int e = 0; e < 16; ++e) s_next[e] = 0.f;
    // A1: first k-step  ||  row maximum of S_t
    if constexpr (PL == 2) {
      s_next = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0][1], qf[0][0], s_next, 0, 0, 0);
      s_next = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0][0], qf[0][1], s_next, 0, 0, 0);
    }
    s_next = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0][0], qf[0][0], s_next, 0, 0, 0);
    // (v_max3_f32 from inline asm: fmaxf() canonicalises each MFMA result first -- 16 extra v_max x, x per tile)
    auto max3 = [](float a, float b_, float c) __attribute__((always_inline)) {
      float r;
      asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b_), "v"(c));
      return r;
    };
    float tm = max3(max3(max3(s_cur[0], s_cur[1], s_cur[2]), max3(s_cur[3], s_cur[4], s_cur[5]), max3(s_cur[6], s_cur[7], s_cur[8])),
                    max3(max3(s_cur[9], s_cur[10], s_cur[11]), max3(s_cur[12], s_cur[13], s_cur[14]), s_cur[15]), -INFINITY);
    tm = xhalf_max(tm);
    if (__any(tm > m_run + FLASH_RESCALE_THR)) {
      const float m_new = fmaxf(m_run, tm);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);       // m_run = -inf (first tile): 0
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[nb][e] *= alpha;
      l_run *= alpha;
      m_run = m_new;
    }
    // A2: k-steps 1 .. KS-1  ||  exponentials, row sums, fp16 P' -- a slice of the 16 elements per k-step, each slice in the
    // scheduling region of that k-step's MFMAs (sched_barrier: bounded live ranges -- a free scheduler hoists every LDS read of the
    // iteration to its top and spills 100 registers)
    __builtin_amdgcn_sched_barrier(0);
    f16x8 pb[2];
    float ls0 = 0.f, ls1 = 0.f;
#pragma unroll
    for (int ks = 1; ks < KS; ++ks) {
      if (ks + 1 < KS) {
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) kf[(ks + 1) & 1][pl] = *reinterpret_cast<const f16x8*>(Kt + ((ks + 1) * PL + pl) * 512);
      }
      if constexpr (PL == 2) {
        s_next = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks & 1][1], qf[ks][0], s_next, 0, 0, 0);
        s_next = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks & 1][0], qf[ks][1], s_next, 0, 0, 0);
      }
      s_next = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks & 1][0], qf[ks][0], s_next, 0, 0, 0);
      if (ks <= 2) {          // k-step 1: P' of keys 0..7 of this lane, k-step 2: keys 8..15
#pragma unroll
        for (int e = (ks - 1) * 8; e < ks * 8; ++e) {
          const float ex = __builtin_amdgcn_exp2f(s_cur[e] - m_run);
          if (e & 1) ls1 += ex; else ls0 += ex;
          pb[ks - 1][e & 7] = (_Float16)ex;
        }
        // (pinned: the values are only consumed by the P.V MFMAs of phase B, and hipcc's code sinking otherwise moves the exponentials
        // down to them, out from under the score MFMAs they are meant to hide behind)
        asm volatile("" : "+v"(pb[ks - 1]), "+v"(ls0), "+v"(ls1));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    l_run += ls0 + ls1;

    // ---- phase B: O^T += V_t^T . P_t^T  ||  bias / mask of S_{t+1} where the tile needs it (two elements per 32-row block of O^T)
    const uint16_t* Vt = &St[st_cur * TILE_H + KT_H + lane * 8];
    // (the P.V MFMAs are unconditional and only the bias slices sit under the block-uniform branch: two copies of the MFMA loop, one
    // per branch, made hipcc keep two sets of the 128 accumulator registers -- 500 bytes of scratch per lane)
    {
      // MFMA order: the first 16-key group of all NB blocks, then the second: the two MFMAs of one accumulator are NB MFMAs apart (a
      // dependent MFMA right behind its predecessor, with LDS reads / VALU in between, costs ~43 extra cycles each: MI355X_MICROARCH)
      f16x8 va[3];                                     // V^T fragments two MFMAs ahead
      va[0] = *reinterpret_cast<const f16x8*>(Vt);
      va[1] = *reinterpret_cast<const f16x8*>(Vt + 512);
      const int j0 = (t + 1) * 32;
      static_assert(16 % NB == 0 && NB >= 2, "bias slices");
#pragma unroll
      for (int i = 0; i < 2 * NB; ++i) {
        const int nb = i % NB, g = i / NB;
        if (i + 2 < 2 * NB) va[(i + 2) % 3] = *reinterpret_cast<const f16x8*>(Vt + (((i + 2) / NB) * NB + (i + 2) % NB) * 512);
        o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va[i % 3], pb[g], o[nb], 0, 0, 0);
        if (nb_ && (i & 1)) {
          const int e0 = (i >> 1) * (16 / NB);
#pragma unroll
          for (int e = e0; e < e0 + 16 / NB; ++e) {
            const int key = j0 + 8 * (e >> 2) + 4 * hh + (e & 3);
            const int kh = (int)__umulhi((unsigned)key, p.w8_magic), kw = key - kh * W8;
            const unsigned u = min((unsigned)(kh + ch), umax), w_ = min((unsigned)(kw + cw), umax);
            float sv = __builtin_amdgcn_fmed3f(s_next[e], -clipv, clipv) + s_tab[u * TW + w_];
            if (key >= N) sv = -INFINITY;
            s_next[e] = sv;
          }
        }
        if (i & 1) __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- tile t+2 has landed (this wave's pieces; requested one iteration ago), everybody is done with stage st_cur: refill it
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    dma(t + 3, st_cur);
    const int tmp = st_cur; st_cur = st_nxt; st_nxt = st_far; st_far = tmp;
  };
  int t = 0;
  for (; t + 1 < nkt; t += 2) {
    iter(t, s_a, s_b);
    iter(t + 1, s_b, s_a);
  }
  if (t < nkt) iter(t, s_a, s_b);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (an LDS-DMA must not outlive the block)

  const float inv = 1.f / xhalf_sum(l_run);
  if (qidx < N) {
    float* orow = p.O + ((long)z * N + qidx) * DV + 4 * hh;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(orow + nb * 32 + 8 * g) =
            make_float4(o[nb][4 * g] * inv, o[nb][4 * g + 1] * inv, o[nb][4 * g + 2] * inv, o[nb][4 * g + 3] * inv);
  }
}

// Q / K -> pre-split fragment order (see the header).  One thread per (z, block, k-step, lane): 8 consecutive floats in,
// 16 (+16) bytes out.  Rows >= N are zero.
template <int PL>
__global__ void k_pack_qk(const float* __restrict__ X, long ld, long bs, int N, int M, int D, int nblk, float mul,
                          uint16_t* __restrict__ out, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int KS = D / 16;
  const int lane = (int)(i & 63);
  long t = i >> 6;
  const int ks = (int)(t % KS); t /= KS;
  const int blk = (int)(t % nblk);
  const int z = (int)(t / nblk);
  const int b = z / M, m = z - b * M;
  const int row = blk * 32 + (lane & 31);
  float v[8];
  if (row < N) {
    const float* src = X + (long)b * bs + (long)row * ld + m * D + ks * 16 + (lane >> 5) * 8;
    const float4 a = *reinterpret_cast<const float4*>(src), c = *reinterpret_cast<const float4*>(src + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
  }
  f16x8 h, l;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = v[j] * mul;
    h[j] = (_Float16)x;
    l[j] = (_Float16)(x - (float)h[j]);
  }
  uint16_t* o = out + ((((long)z * nblk + blk) * KS + ks) * PL) * 512 + lane * 8;
  *reinterpret_cast<f16x8*>(o) = h;
  if constexpr (PL == 2) *reinterpret_cast<f16x8*>(o + 512) = l;
}

size_t flash_ws_bytes(int B, int M, int N, int d, int score_prec) {
  const int PL = score_prec == CRAFT_PREC_F16X3 ? 2 : 1;
  const long nqb = (long)((N + 255) / 256) * 8, nkb = (N + 31) / 32;
  return (size_t)B * M * (nqb + nkb) * (d / 16) * PL * 512 * 2;
}

int launch_flash_attn(const ScoreParams& sp, const void* vT, long ldt, int Dv, float* O, void* ws, int score_prec, int pv_prec,
                      hipStream_t s) {
  if (sp.d != 64 || Dv != 256 || pv_prec != CRAFT_PREC_F16 || sp.rb_h != nullptr) return CRAFT_ERR_UNSUPPORTED;
  if (score_prec != CRAFT_PREC_F16X3 && score_prec != CRAFT_PREC_F16) return CRAFT_ERR_UNSUPPORTED;
  if (sp.N >= 65536 || sp.W8 < 2 || sp.N < 1) return CRAFT_ERR_UNSUPPORTED;
  if ((sp.ldq & 3) || (sp.ldk & 3) || ldt % 32 || ldt < sp.N) return CRAFT_ERR_ALIGN;
  if (sp.pos_tab && sp.R > 15) return CRAFT_ERR_UNSUPPORTED;
  if (sp.mask_radius > 15) return CRAFT_ERR_UNSUPPORTED;
  const int PL = score_prec == CRAFT_PREC_F16X3 ? 2 : 1;
  const int Z = sp.B * sp.M, KS = sp.d / 16;
  const int nqb = ((sp.N + 255) / 256) * 8, nkb = (sp.N + 31) / 32;
  uint16_t* Qf = reinterpret_cast<uint16_t*>(ws);
  uint16_t* Kf = Qf + (long)Z * nqb * KS * PL * 512;
  constexpr float LOG2E = 1.4426950408889634f;
  {
    const long tq = (long)Z * nqb * KS * 64, tk = (long)Z * nkb * KS * 64;
    if (PL == 2) {
      hipLaunchKernelGGL((k_pack_qk<2>), dim3((unsigned)((tq + 255) / 256)), dim3(256), 0, s, sp.Q, sp.ldq, sp.q_bs, sp.N, sp.M,
                         sp.d, nqb, sp.scale * LOG2E, Qf, tq);
      hipLaunchKernelGGL((k_pack_qk<2>), dim3((unsigned)((tk + 255) / 256)), dim3(256), 0, s, sp.Kf, sp.ldk, sp.k_bs, sp.N, sp.M,
                         sp.d, nkb, 1.f, Kf, tk);
    } else {
      hipLaunchKernelGGL((k_pack_qk<1>), dim3((unsigned)((tq + 255) / 256)), dim3(256), 0, s, sp.Q, sp.ldq, sp.q_bs, sp.N, sp.M,
                         sp.d, nqb, sp.scale * LOG2E, Qf, tq);
      hipLaunchKernelGGL((k_pack_qk<1>), dim3((unsigned)((tk + 255) / 256)), dim3(256), 0, s, sp.Kf, sp.ldk, sp.k_bs, sp.N, sp.M,
                         sp.d, nkb, 1.f, Kf, tk);
    }
  }
  FlashParams p = {};
  p.Qf = Qf; p.Kf = Kf; p.Vf = reinterpret_cast<const uint16_t*>(vT);
  p.v_bs = (long)sp.M * Dv * ldt; p.v_ms = (long)Dv * ldt;
  p.O = O;
  p.B = sp.B; p.M = sp.M; p.N = sp.N; p.H8 = sp.H8; p.W8 = sp.W8;
  p.nkt = nkb; p.nqb = nqb; p.nkb = nkb;
  p.w8_magic = (unsigned)((0x100000000ULL + (unsigned)sp.W8 - 1) / (unsigned)sp.W8);
  p.pos_tab = sp.pos_tab; p.R = sp.R; p.pos_w = sp.pos_w; p.mask_radius = sp.mask_radius; p.clamp_ord = sp.clamp_ord;
  dim3 grid(((sp.N + 255) / 256) * Z);
  // k_flash_attn2 reads whole K / V^T tiles by LDS-DMA with no range limit: tile nkt - 1 must exist in full (nkb K blocks are allocated by
  // flash_ws_bytes; V^T needs ldt >= 32 * nkb, which ldt % 32 == 0 && ldt >= N guarantees)
  if (PL == 2) hipLaunchKernelGGL((k_flash_attn2<64, 256, 2>), grid, dim3(512), 0, s, p);
  else hipLaunchKernelGGL((k_flash_attn2<64, 256, 1>), grid, dim3(512), 0, s, p);
  return (int)hipGetLastError();
}

}  // namespace craft
