// Deferred-normalisation attention probabilities in ONE launch of independent waves (round 4; replaces the two chunked launches of
// attn_probs.inc.hpp -- row maxima, then P' -- for the canonical intra-frame attention: setrans.py:507-557, d = 32, 16-bit P).
//
//   P'[b][m][i][j] = 2^(t_ij - max_j t_ij)  in (0, 1],   rowsum[b][m][i] = sum_j P'_ij      (t = base-2 logits incl. bias / mask / clamp)
//
// Why: the old write kernel ran at 0.9 waves per SIMD with two block barriers per 128-key tile (rocprof: 33 % VALU issue, 35 % waiting,
// MFMA 7 % busy) and the maxima were a separate launch over the same products.  Here a wave is the unit of work: it owns 32 queries
// for the whole key range, keeps their Q fragments in registers, and streams the KEYS straight from L2 into MFMA registers -- K is
// pre-split ONCE into fp16 hi / lo planes in MFMA fragment order (one A operand = one contiguous 1 KiB: the "weights never touch LDS"
// idiom of k_conv_halo_wf), so there is no operand staging, no block barrier, and occupancy is bounded by registers only.  Phase A walks
// the keys for the exact row maxima (same three-term products as phase B, so P' <= 1 exactly), phase B walks them again, exponentiates and
// writes P' through a wave-private LDS transpose as whole 256-byte row segments.  Both phases are software-pipelined over 32-key halves:
// the epilogue of one half is issued between the MFMAs of the next (VALU only hides behind MFMAs of the same wave).  All waves of one (b, m) run on one XCD pair
// (z = block % (B*M)): K of a (b, m) is 0.9 MB and stays in that XCD's L2.
#include "gemm_engine.hpp"
#include "launch.hpp"
#include <type_traits>

namespace craft {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// K [B][N][ldk] fp32, mode m = columns m*32 .. m*32+31  ->  Kp[z = b*M + m][t32][plane][kk][lane][8] fp16:
// lane (r = lane & 31, g = lane >> 5) holds key 32*t32 + r, columns kk*16 + g*8 .. +7  (A operand of v_mfma_f32_32x32x16_f16)
__global__ __launch_bounds__(256) void k_pack_keys32(const float* __restrict__ K, long ldk, long k_bs, int N, int M, int nt32,
                                                     _Float16* __restrict__ Kp, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;         // one thread per (z, t32, kk, lane)
  if (i >= total) return;
  const int lane = (int)(i & 63), kk = (int)((i >> 6) & 1);
  const long zt = i >> 7;
  const int t32 = (int)(zt % nt32), z = (int)(zt / nt32);
  const int b = z / M, m = z - b * M;
  const int key = 32 * t32 + (lane & 31);
  f16x4 h0, l0, h1, l1;
  if (key < N) {
    const float* src = K + (long)b * k_bs + (long)key * ldk + m * 32 + kk * 16 + (lane >> 5) * 8;
    split_f16x3(*reinterpret_cast<const float4*>(src), h0, l0);
    split_f16x3(*reinterpret_cast<const float4*>(src + 4), h1, l1);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) { h0[e] = l0[e] = h1[e] = l1[e] = (_Float16)0.f; }
  }
  _Float16* dst = Kp + ((zt * 2 + 0) * 2 + kk) * 512 + lane * 8;        // plane 0 (hi); plane 1 (lo) is 2 * 512 halves further
  f16x8 hh, ll;
#pragma unroll
  for (int e = 0; e < 4; ++e) { hh[e] = h0[e]; hh[4 + e] = h1[e]; ll[e] = l0[e]; ll[4 + e] = l1[e]; }
  *reinterpret_cast<f16x8*>(dst) = hh;
  *reinterpret_cast<f16x8*>(dst + 2 * 512) = ll;
}

#ifndef CRAFT_AW_WAVES
#define CRAFT_AW_WAVES 3      // waves per SIMD the register allocation is held to (tools/build_variant.py -DCRAFT_AW_WAVES=3 for A/B)
#endif
#ifndef CRAFT_AW_APPROX_MAX
#define CRAFT_AW_APPROX_MAX 0  // developer A/B: 1 = phase A from the hi x hi products only (row maxima to ~1e-3 of |q||k|: P' <= 2^delta)
#endif
#ifndef CRAFT_AW_DBG
#define CRAFT_AW_DBG 0        // developer ablation: 1 no global stores of P', 2 no phase A, 4 no LDS transpose, 8 plain instead of non-temporal stores
#endif
constexpr int AW_TABW = 33;
constexpr int AW_PLD = 128 + 8;       // halves per staged row: 272 B (16-byte aligned rows)

// X3: f16x3 scores (hi*hi + hi*lo + lo*hi), else plain fp16 (hi planes only)
// TILED: P' in 32-query x 64-key tiles (CRAFT_P_TILED: element (i, j) at (i >> 5) * 32 * ldp + (j >> 6) * 2048 + (i & 31) * 64 + (j & 63)) --
// the wave's 128-key step is then 8 KiB of contiguous lines, and what k_pv16 reads per band and K-tile one contiguous 4 KiB
template <bool X3, typename prob_t, bool TILED>
__global__ __launch_bounds__(256, CRAFT_AW_WAVES) void k_attn_probs_w(ScoreParams p, const _Float16* __restrict__ Kp, prob_t* __restrict__ P, long ldp,
                                                      int nt32, unsigned w8_magic) {
  __shared__ float s_tab[AW_TABW * AW_TABW];
  __shared__ __attribute__((aligned(16))) uint16_t Pst[4][32 * AW_PLD];
  constexpr float LOG2E = 1.4426950408889634f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int BM = p.B * p.M;
  const int z = blockIdx.x % BM, qg = (blockIdx.x / BM) * 4 + wave;      // z is block-uniform; consecutive blocks cycle through z
  const int b = z / p.M, m = z - b * p.M;
  const int N = p.N, W8 = p.W8, R = p.R;
  const int mr = p.mask_radius > 0 ? p.mask_radius : 0;
  const int Re = max(p.pos_tab ? R : 0, mr), TW = 2 * Re + 3;
  for (int i = tid; i < TW * TW; i += 256) {
    const int dh = i / TW - Re - 1, dw = i - (i / TW) * TW - Re - 1;
    float v = 0.f;
    if (p.pos_tab && abs(dh) <= R && abs(dw) <= R) v = p.pos_tab[(dh + R) * (2 * R + 1) + dw + R] * (p.pos_w * LOG2E);
    if (mr > 0 && max(abs(dh), abs(dw)) > mr) v += -1e9f;
    s_tab[i] = v;
  }
  __syncthreads();                                   // the only block-level synchronisation of the kernel
  const int q0 = qg * 32;
  if (q0 >= N) return;
  const int r = lane & 31, g = lane >> 5;
  const int qcol = q0 + r, qc = min(qcol, N - 1);
  const int h1 = qc / W8, w1 = qc - h1 * W8;
  const bool clamp = p.clamp_ord != nullptr && ord2f(*p.clamp_ord) > CRAFT_ATTN_CLIP;
  const float clipv = clamp ? CRAFT_ATTN_CLIP * LOG2E : 3.0e38f;
  const int ch = Re + 1 - h1, cw = Re + 1 - w1 + 4 * g;          // (+ 4 g: the lane's key offset inside a quad row)
  const unsigned umax = 2 * Re + 2;
  const int q_hmin = q0 / W8, q_hmax = min(q0 + 31, N - 1) / W8;

  // Q fragments of this lane (B operand: query r, columns kk*16 + g*8 .. +7), pre-multiplied by scale * log2(e)
  f16x8 bh[2], bl[2];
  {
    const float* qp = p.Q + (long)b * p.q_bs + (long)qc * p.ldq + m * 32 + g * 8;
    const float qs = p.scale * LOG2E;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      float4 a = *reinterpret_cast<const float4*>(qp + kk * 16), c = *reinterpret_cast<const float4*>(qp + kk * 16 + 4);
      a.x *= qs; a.y *= qs; a.z *= qs; a.w *= qs; c.x *= qs; c.y *= qs; c.z *= qs; c.w *= qs;
      f16x4 h0, l0, h1_, l1_;
      if constexpr (X3) { split_f16x3(a, h0, l0); split_f16x3(c, h1_, l1_); }
      else {
        h0[0] = (_Float16)a.x; h0[1] = (_Float16)a.y; h0[2] = (_Float16)a.z; h0[3] = (_Float16)a.w;
        h1_[0] = (_Float16)c.x; h1_[1] = (_Float16)c.y; h1_[2] = (_Float16)c.z; h1_[3] = (_Float16)c.w;
        l0 = h0; l1_ = h1_;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { bh[kk][e] = h0[e]; bh[kk][4 + e] = h1_[e]; bl[kk][e] = l0[e]; bl[kk][4 + e] = l1_[e]; }
    }
  }

  const _Float16* kz = Kp + (long)z * nt32 * 2048 + lane * 8;        // a 32-key half-tile = 2 planes x 2 k-steps x 512 halves
  constexpr int NH = X3 ? 4 : 2;                                     // A fragments per 32-key half: [plane][kk]
  constexpr int NM = X3 ? 6 : 2;                                     // MFMAs per half
  // The unit of the software pipeline is a 32-key HALF tile (one 32x32 accumulator): while the matrix pipe works on the products of
  // half h + 1, the same wave issues the epilogue of half h between those MFMAs (VALU work only hides behind MFMAs of its own wave,
  // DESIGN 5).  Two accumulators and two operand sets alternate; the operands of half h + 2 are requested as soon as the MFMAs of half
  // h have been issued.
  auto fetch_half = [&](u32x4 (&f)[NH], int t32) __attribute__((always_inline)) {
    const _Float16* base = kz + (long)min(t32, nt32 - 1) * 2048;     // (halves beyond the packed range re-read the last one: never used)
#pragma unroll
    for (int pl = 0; pl < (X3 ? 2 : 1); ++pl)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) f[pl * 2 + kk] = *reinterpret_cast<const u32x4*>(base + (pl * 2 + kk) * 512);
  };
  auto mma_half = [&](const u32x4 (&f)[NH], f32x16& acc, auto full_c) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const f16x8 ah = __builtin_bit_cast(f16x8, f[kk]);
      if constexpr (X3 && FULL) {
        const f16x8 al = __builtin_bit_cast(f16x8, f[2 + kk]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[kk], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[kk], acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[kk], acc, 0, 0, 0);
    }
  };
  // one pipeline stage: products of the NEXT half into accN, base-2 logits of half t32 from accC handed element by element to
  // sink(e, logit) -- key 32 t32 + (e & 3) + 8 (e >> 2) + 4 g of this lane's query.  ONE wave-uniform branch per half picks the path
  // (bare / one-row table / generic); the MFMAs are issued inside each path so that they share a basic block with the epilogue they
  // hide, and the accumulator tuples are only read (in-place edits under control flow made hipcc copy them: 256 VGPRs).
  auto stage = [&](const u32x4 (&fN)[NH], f32x16& accN, const f32x16& accC, int t32, auto vper_c, auto full_c, auto&& sink) __attribute__((always_inline)) {
    constexpr int vper = decltype(vper_c)::value;
    const int j0 = t32 * 32;
    const int kh0 = (int)__umulhi((unsigned)j0, w8_magic), kw0 = j0 - kh0 * W8;         // wave-uniform
    const int k_hmax = (int)__umulhi((unsigned)min(j0 + 31, N - 1), w8_magic);
    const bool need_tab = clamp || mr > 0 || (p.pos_tab != nullptr && k_hmax >= q_hmin - R && kh0 <= q_hmax + R);
    const bool ragged = j0 + 32 > N;
    auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, vper, 0);
      }
    };
    if (!need_tab && !ragged) {
      mma_half(fN, accN, full_c);
#pragma unroll
      for (int e = 0; e < 16; ++e) sink(e, accC[e]);
      interleave();
    } else if (kw0 + 32 <= W8 && !ragged) {        // the half lies in one image row: kh uniform, kw = kw0 + offset
      mma_half(fN, accN, full_c);
      const unsigned u = min((unsigned)(kh0 + ch), umax);
      const float* trow = s_tab + u * TW;
      int vb = kw0 + cw;
      asm volatile("" : "+v"(vb));                 // (opaque: otherwise LICM keeps 16 lane-dependent (cw + offset) values live across the loop)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const unsigned v = min((unsigned)(vb + (e & 3) + 8 * (e >> 2)), umax);
        sink(e, __builtin_amdgcn_fmed3f(accC[e], -clipv, clipv) + trow[v]);
      }
      interleave();
    } else {
      mma_half(fN, accN, full_c);
      int jb = j0 + 4 * g, cwb = cw - 4 * g;
      asm volatile("" : "+v"(jb), "+v"(cwb));
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int j = jb + ((e & 3) + 8 * (e >> 2));
        const int kh = (int)__umulhi((unsigned)j, w8_magic), kw = j - kh * W8;
        const unsigned u = min((unsigned)(kh + ch), umax), v = min((unsigned)(kw + cwb), umax);
        const float sv = __builtin_amdgcn_fmed3f(accC[e], -clipv, clipv) + s_tab[u * TW + v];
        sink(e, j < N ? sv : -INFINITY);
        if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);        // (rare path: live ranges of one quad, no interleaving)
      }
    }
  };
  u32x4 fa0[NH], fa1[NH];
  f32x16 acc0, acc1;

  // ---- phase A: exact row maxima (nt32 is a multiple of 4)
  float m_run = -INFINITY;
  constexpr std::integral_constant<bool, !CRAFT_AW_APPROX_MAX> FULL_A{};
  auto sink_max = [&](int, float sv) __attribute__((always_inline)) { m_run = fmaxf(m_run, sv); };
  fetch_half(fa0, 0);
  fetch_half(fa1, 1);
  mma_half(fa0, acc0, FULL_A);
  __builtin_amdgcn_sched_barrier(0);
  fetch_half(fa0, 2);
  for (int t32 = 0; t32 < ((CRAFT_AW_DBG & 2) ? 2 : nt32); t32 += 2) {
    stage(fa1, acc1, acc0, t32, std::integral_constant<int, 3>{}, FULL_A, sink_max);
    __builtin_amdgcn_sched_barrier(0);
    fetch_half(fa1, t32 + 3);
    __builtin_amdgcn_sched_barrier(0);
    stage(fa0, acc0, acc1, t32 + 1, std::integral_constant<int, 3>{}, FULL_A, sink_max);
    __builtin_amdgcn_sched_barrier(0);
    fetch_half(fa0, t32 + 4);
    __builtin_amdgcn_sched_barrier(0);
  }
  m_run = fmaxf(m_run, __shfl_xor(m_run, 32));       // the two half-waves hold disjoint keys of the same query

  // ---- phase B: P' = 2^(t - max) -> fp16 / bf16 through the wave's LDS tile, whole 128-byte row segments to HBM; row sums
  uint16_t* Tw = &Pst[wave][0];
  float l_run = 0.f;
  typedef prob_t pt4 __attribute__((ext_vector_type(4)));
  pt4 h;
  int col0 = 4 * g;
  auto sink_exp = [&](int e, float sv) __attribute__((always_inline)) {
    const float ex = __builtin_amdgcn_exp2f(sv - m_run);
    h[e & 3] = (prob_t)ex;
    l_run += ex;
    if ((e & 3) == 3 && (!(CRAFT_AW_DBG & 4) || ex == 123.f)) *reinterpret_cast<pt4*>(&Tw[r * AW_PLD + col0 + 8 * (e >> 2)]) = h;
  };
  fetch_half(fa0, 0);
  fetch_half(fa1, 1);
  mma_half(fa0, acc0, std::true_type{});
  __builtin_amdgcn_sched_barrier(0);
  fetch_half(fa0, 2);
  const int srow = lane >> 4, sch = lane & 15;       // store mapping: 4 rows x 16 chunks of 16 bytes per instruction (256-byte row segments)
  // (tiled: band qg of entry z, whose padded row count is 32 * ceil(N / 32); chunk sch -> 64-key half sch >> 3, 16 bytes (sch & 7))
  prob_t* Pw = TILED ? P + ((long)z * ((N + 31) >> 5) + qg) * 32 * ldp + (sch >> 3) * 2048 + (sch & 7) * 8 : P + ((long)z * N + q0) * ldp + sch * 8;
  for (int t32 = 0; t32 < nt32; t32 += 4) {          // nt32 is a multiple of 4: one 128-key tile of P' per iteration
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {
      col0 = 64 * hp + 4 * g;
      stage(fa1, acc1, acc0, t32 + 2 * hp, std::integral_constant<int, 9>{}, std::true_type{}, sink_exp);
      __builtin_amdgcn_sched_barrier(0);
      fetch_half(fa1, t32 + 2 * hp + 3);
      __builtin_amdgcn_sched_barrier(0);
      col0 = 64 * hp + 32 + 4 * g;
      stage(fa0, acc0, acc1, t32 + 2 * hp + 1, std::integral_constant<int, 9>{}, std::true_type{}, sink_exp);
      __builtin_amdgcn_sched_barrier(0);
      fetch_half(fa0, t32 + 2 * hp + 4);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int j0 = t32 * 32;
    const bool jok = j0 + sch * 8 < ldp;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 4 + srow;
      const u32x4 v = *reinterpret_cast<const u32x4*>(&Tw[row * AW_PLD + sch * 8]);
      if (jok && q0 + row < N && !(CRAFT_AW_DBG & 1)) {
        // non-temporal: P' (1.6 GB at 448x1024, batch 4) is read back long after this kernel; measured 0.72 -> 0.625 ms
        prob_t* dst = TILED ? Pw + (long)(t32 >> 2) * 4096 + row * 64 : Pw + (long)row * ldp + j0;
        if (CRAFT_AW_DBG & 8) *reinterpret_cast<u32x4*>(dst) = v;
        else __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const float l = l_run + __shfl_xor(l_run, 32);
  if (lane < 32 && qcol < N) p.rowsum[(long)z * N + qcol] = l;
}

int launch_attn_probs_fused(const ScoreParams& p, void* P, long ldp, void* ws, int p_prec, int prec, int tiled, hipStream_t s) {
  if (p.d != 32 || p.rb_h != nullptr || p.rowsum == nullptr || ws == nullptr) return CRAFT_ERR_UNSUPPORTED;
  if (prec != CRAFT_PREC_F16X3 && prec != CRAFT_PREC_F16) return CRAFT_ERR_UNSUPPORTED;
  if (p_prec != CRAFT_PREC_F16 && p_prec != CRAFT_PREC_BF16) return CRAFT_ERR_UNSUPPORTED;
  if (p.N >= 65536 || p.N < 1) return CRAFT_ERR_UNSUPPORTED;                       // (umulhi division of key indices)
  if (ldp % (tiled ? 64 : 32) || ldp < p.N || (p.ldq & 3) || (p.ldk & 3) || (p.q_bs & 3) || (p.k_bs & 3)) return CRAFT_ERR_ALIGN;
  if (p.mask_radius > 15 || (p.pos_tab && p.R > 15)) return CRAFT_ERR_UNSUPPORTED;
  const int nt32 = (p.N + 127) / 128 * 4, BM = p.B * p.M;
  _Float16* Kp = reinterpret_cast<_Float16*>(ws);
  const long total = (long)BM * nt32 * 128;
  hipLaunchKernelGGL(k_pack_keys32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p.Kf, p.ldk, p.k_bs, p.N, p.M, nt32, Kp, total);
  const unsigned magic = (unsigned)((0x100000000ull + (unsigned)p.W8 - 1) / (unsigned)p.W8);
  const int nqg = (p.N + 31) / 32;
  dim3 grid((unsigned)(((nqg + 3) / 4) * BM));
#define GO1(X3, PT, TL) hipLaunchKernelGGL((k_attn_probs_w<X3, PT, TL>), grid, dim3(256), 0, s, p, Kp, reinterpret_cast<PT*>(P), ldp, nt32, magic)
#define GO(X3, PT) do { if (tiled) GO1(X3, PT, true); else GO1(X3, PT, false); } while (0)
  if (prec == CRAFT_PREC_F16X3) { if (p_prec == CRAFT_PREC_F16) GO(true, _Float16); else GO(true, __bf16); }
  else { if (p_prec == CRAFT_PREC_F16) GO(false, _Float16); else GO(false, __bf16); }
#undef GO
#undef GO1
  return (int)hipGetLastError();
}

}  // namespace craft
