// Attention probabilities  P[b][m][i][j] = softmax_j( clamp?(Q_m(i).K_m(j) * scale) + pw*pb(i,j) + mask )
// (CrossAttFeatTrans up to the softmax, setrans.py:507-557; mask of SelfAttVisPosTrans :580-584).
//
// Included once per per-mode width D (ATTN_D = 32 / 64 / 128) by kernels_attn_d*.hip so the
// instantiations compile in parallel.
//
// grid (query tiles of 128, B*M), 256 threads.  "Swapped" product: the key tile is the MFMA row operand
// and the query tile the column operand, so each lane owns ONE query (col = lane & 31) and 64 of the 128
// keys of a tile: the softmax row statistics stay lane-local (one exchange between the half-waves at the
// end of the first pass).  The query tile is staged into LDS once; key tiles stream through a single LDS
// buffer with register prefetch (the next tile's global loads are in flight during MFMA + epilogue of the
// current one).  Two passes over the keys: (row max, row sum) online, then the normalised write — P is
// written exactly once and the scores are never materialised.  Tiles that no query of the block can see
// through the (2R+1)^2 positional window skip the bias code (block-uniform test on image rows).
#include "gemm_engine.hpp"
#include "launch.hpp"

namespace craft {

template <int PREC> struct ProbT;
template <> struct ProbT<CRAFT_PREC_F32> { typedef float t; };
template <> struct ProbT<CRAFT_PREC_BF16> { typedef __bf16 t; };
template <> struct ProbT<CRAFT_PREC_F16> { typedef _Float16 t; };

template <int PREC, int D> struct RowTile {
  typedef typename PrecT<PREC>::lds_t lds_t;
  static constexpr int LD = (PREC == CRAFT_PREC_F32) ? D + 4 : D + 8;   // row stride (elements), 16-B aligned
  static constexpr int PL = Planes<PREC>::N;
  static constexpr int ELEMS = PL * 128 * LD;
  static constexpr int NCH = D / 4;          // float4 chunks per row
  static constexpr int RPP = 256 / NCH;      // rows covered per pass of the 256 threads
  static constexpr int NPASS = 128 / RPP;    // float4 per thread per 128-row tile

  // global -> registers: rows [row0, row0+128) of a [*, ld] fp32 matrix, columns [0, D); rows >= nrows are zero
  static __device__ __forceinline__ void fetch(float4 (&r)[NPASS], const float* base, long ld, int row0, int nrows, int tid) {
    const int ch = tid % NCH, rr = tid / NCH;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      // unconditional load (rows beyond the matrix re-read the last row: such keys are masked to -inf by the
      // caller, such queries are never written) -- a guarded load would serialise on vmcnt(0)
      const int row = min(row0 + rr + RPP * i, nrows - 1);
      r[i] = *reinterpret_cast<const float4*>(base + (long)row * ld + ch * 4);
    }
  }
  // registers -> LDS (convert / split as the precision requires)
  static __device__ __forceinline__ void store(lds_t* S, const float4 (&r)[NPASS], int tid, float mul = 1.f) {
    const int ch = tid % NCH, rr = tid / NCH;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int row = rr + RPP * i;
      const float4 v = make_float4(r[i].x * mul, r[i].y * mul, r[i].z * mul, r[i].w * mul);
      if constexpr (PREC == CRAFT_PREC_F32) {
        *reinterpret_cast<float4*>(&S[row * LD + ch * 4]) = v;
      } else if constexpr (PREC == CRAFT_PREC_BF16) {
        bf16x4 h;
        h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4*>(&S[row * LD + ch * 4]) = h;
      } else {
        if constexpr (PREC == CRAFT_PREC_F16X3) {
          f16x4 h, l;
          split_f16x3(v, h, l);
          *reinterpret_cast<f16x4*>(&S[row * LD + ch * 4]) = h;
          *reinterpret_cast<f16x4*>(&S[(128 + row) * LD + ch * 4]) = l;
        } else {
          f16x4 h;
          h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
          *reinterpret_cast<f16x4*>(&S[row * LD + ch * 4]) = h;
        }
      }
    }
  }
};

// acc[mt] += Ktile(rows mt*32.., all D) . Qtile(rows qrow0.., all D)^T   for one wave (128 keys x 32 queries)
template <int PREC, int D>
__device__ __forceinline__ void mma_rows(const typename PrecT<PREC>::lds_t* Ks, const typename PrecT<PREC>::lds_t* Qs,
                                         int qrow0, int lane, f32x16 (&acc)[4]) {
  constexpr int LD = RowTile<PREC, D>::LD;
  const int r = lane & 31, g = lane >> 5;
  if constexpr (PREC == CRAFT_PREC_F32) {
#pragma unroll
    for (int kk = 0; kk < D / 8; ++kk) {
      const float4 b = *reinterpret_cast<const float4*>(&Qs[(qrow0 + r) * LD + kk * 8 + g * 4]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const float4 a = *reinterpret_cast<const float4*>(&Ks[(mt * 32 + r) * LD + kk * 8 + g * 4]);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[mt], 0, 0, 0);
      }
    }
  } else if constexpr (PREC == CRAFT_PREC_BF16) {
#pragma unroll
    for (int kk = 0; kk < D / 16; ++kk) {
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(&Qs[(qrow0 + r) * LD + kk * 16 + g * 8]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(&Ks[(mt * 32 + r) * LD + kk * 16 + g * 8]);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[mt], 0, 0, 0);
      }
    }
  } else {
#pragma unroll
    for (int kk = 0; kk < D / 16; ++kk) {
      const f16x8 bh = *reinterpret_cast<const f16x8*>(&Qs[(qrow0 + r) * LD + kk * 16 + g * 8]);
      f16x8 bl;
      if constexpr (PREC == CRAFT_PREC_F16X3) bl = *reinterpret_cast<const f16x8*>(&Qs[(128 + qrow0 + r) * LD + kk * 16 + g * 8]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(&Ks[(mt * 32 + r) * LD + kk * 16 + g * 8]);
        if constexpr (PREC == CRAFT_PREC_F16X3) {
          const f16x8 al = *reinterpret_cast<const f16x8*>(&Ks[(128 + mt * 32 + r) * LD + kk * 16 + g * 8]);
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[mt], 0, 0, 0);
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[mt], 0, 0, 0);
        }
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[mt], 0, 0, 0);
      }
    }
  }
}

// MODE 0: classic two-pass softmax of one 128-query tile over all keys -> normalised P.
// Deferred normalisation (p.rowsum != NULL): P'[i][j] = 2^(t_ij - max_i) in (0, 1] plus the row sums l_i; the consumer
// (craft_attn_apply) divides its output rows by l_i.  Finding max_i needs no exponentials, and the two steps have no
// sequential dependency inside a block any more, so they run as two launches over (query tile, batch*mode, KEY CHUNK):
//   MODE 1: row maxima of the chunk -> atomicMax into p.rowmax (ordered uints),
//   MODE 2: P' of the chunk -> global, partial row sums -> p.rowsum[2 + chunk] (summed by k_rowsum_reduce).
// The finer work items matter as much as the saved exponentials: at 448x1024 the classic grid is 896 blocks for 768
// resident slots (1.17 rounds: the second round runs 17 % full), the chunked grids have thousands of blocks.
constexpr int ATTN_KC = 8;             // key tiles (of 128 keys) per work item in MODE 1 / 2
constexpr int ATTN_TABW = 33;          // bias table width: 2 * 15 + 3

template <int PREC, int PT, int D, int MODE>
__global__ __launch_bounds__(NTHREADS) void k_attn_probs(ScoreParams p, void* __restrict__ Pout, long ldp) {
  typedef RowTile<PREC, D> T;
  typedef typename T::lds_t lds_t;
  typedef typename ProbT<PT>::t prob_t;
  constexpr bool kExact = (PREC == CRAFT_PREC_F32);
  constexpr bool DEFER = MODE != 0;
  __shared__ __attribute__((aligned(16))) lds_t Qs[T::ELEMS];
  __shared__ __attribute__((aligned(16))) lds_t Ks[T::ELEMS];
  __shared__ int s_kh[128], s_kw[128];
  __shared__ float s_tab[ATTN_TABW * ATTN_TABW];
  // 16-bit P: each wave transposes its 32-query x 128-key tile through LDS so that the global stores are whole
  // 256-byte row segments.  In the accumulator layout a store instruction touches 32 rows x 16 bytes, and the
  // kernel that writes P is bound by exactly that request stream (1.6 GB of P went out at 1.8 TB/s).
  constexpr int PLD = 128 + 8;
  constexpr bool kStage = MODE != 1 && PT != CRAFT_PREC_F32 && (2 * T::ELEMS * (int)sizeof(lds_t) + 4 * 32 * PLD * 2 <= 150 * 1024);
  __shared__ __attribute__((aligned(16))) uint16_t Pst[kStage ? 4 * 32 * PLD : 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 128, z = blockIdx.y;
  const int b = z / p.M, m = z - b * p.M;
  const int N = p.N, W8 = p.W8, R = p.R;
  const int qcol = n0 + wave * 32 + (lane & 31);
  const int h1 = qcol / W8, w1 = qcol - h1 * W8;
  const int rh4 = 4 * (lane >> 5);
  const bool clamp = p.clamp_ord != nullptr && ord2f(*p.clamp_ord) > CRAFT_ATTN_CLIP;
  const int nkt = (N + 127) / 128;
  const int kt0 = MODE == 0 ? 0 : blockIdx.z * ATTN_KC, kt1 = MODE == 0 ? nkt : min(nkt, kt0 + ATTN_KC);
  const int ntile = kt1 - kt0;
  const int q_hmin = n0 / W8, q_hmax = min(n0 + 127, N - 1) / W8;
  // Everything runs in the base-2 log domain: the query tile is pre-multiplied by scale*log2(e) when it is staged
  // (once per block), so a logit costs no multiply and the softmax uses v_exp_f32 (2^x) directly.
  // Positional bias and Chebyshev mask come from ONE table over (dh, dw) in [-Re-1, Re+1]^2, Re = max(R, mask_radius):
  // bias * pos_w * log2(e) inside the bias window, -1e9 outside the mask radius, and a border ring that stands for
  // "anything farther" (0, or -1e9 with a mask).  Per element: two unsigned clamps, one mad, one LDS read, one add.
  constexpr float LOG2E = 1.4426950408889634f;
  const int mr = p.mask_radius > 0 ? p.mask_radius : 0;
  const int Re = max(p.pos_tab ? R : 0, mr), TW = 2 * Re + 3;
  for (int i = tid; i < TW * TW; i += NTHREADS) {
    const int dh = i / TW - Re - 1, dw = i - (i / TW) * TW - Re - 1;
    float v = 0.f;
    if (p.pos_tab && abs(dh) <= R && abs(dw) <= R) v = p.pos_tab[(dh + R) * (2 * R + 1) + dw + R] * (p.pos_w * LOG2E);
    if (mr > 0 && max(abs(dh), abs(dw)) > mr) v += -1e9f;
    s_tab[i] = v;
  }
  const float clipv = clamp ? CRAFT_ATTN_CLIP * LOG2E : 3.0e38f;
  const int ch = Re + 1 - h1, cw = Re + 1 - w1;       // (unsigned)(kh + ch) = dh + Re + 1, clamped to [0, 2Re+2]
  const unsigned umax = 2 * Re + 2;

  // relative-position rows of this lane's query (gma.RelPosEmb): base-2 domain like everything else
  const bool has_rb = p.rb_h != nullptr;
  const long qrow = (long)z * N + min(qcol, N - 1);
  const float* rbh = has_rb ? p.rb_h + qrow * p.ld_rbh + (p.H8 - 1 - h1) : nullptr;      // indexed by the key's row kh
  const float* rbw = has_rb ? p.rb_wd + qrow * p.ld_rbw + (W8 - 1 - w1) : nullptr;       // indexed by the key's column kw
  const float rbs = p.rb_w * LOG2E;

  const float* qbase = p.Q + (long)b * p.q_bs + (long)m * D;
  const float* kbase = p.Kf + (long)b * p.k_bs + (long)m * D;
  prob_t* Prow = reinterpret_cast<prob_t*>(Pout) + ((long)z * N + qcol) * ldp;

  float4 rk[T::NPASS];
  T::fetch(rk, qbase, p.ldq, n0, N, tid);
  T::store(Qs, rk, tid, p.scale * LOG2E);
  T::fetch(rk, kbase, p.ldk, kt0 * 128, N, tid);
  T::store(Ks, rk, tid);
  if (tid < 128) { const int j = kt0 * 128 + tid; s_kh[tid] = j / W8; s_kw[tid] = j - (j / W8) * W8; }
  __syncthreads();

  float m_run = -INFINITY, l_run = 0.f, inv_l = 1.f;
  if constexpr (MODE == 2) m_run = ord2f(p.rowmax[(long)z * N + min(qcol, N - 1)]);
  const int T2 = MODE == 0 ? 2 * ntile : ntile;
  for (int t = 0; t < T2; ++t) {
    const bool pass1 = MODE == 2 || (MODE == 0 && t >= ntile);       // true: the tile is written out
    const int jt = kt0 + ((MODE == 0 && t >= ntile) ? t - ntile : t);
    const int tn = t + 1 < T2 ? t + 1 : 0;                          // (after the last tile: a harmless refetch)
    const int jn = kt0 + ((MODE == 0 && tn >= ntile) ? tn - ntile : tn);
    T::fetch(rk, kbase, p.ldk, jn * 128, N, tid);
    __builtin_amdgcn_sched_barrier(0);     // loads stay above the MFMAs + epilogue they are meant to hide behind

    f32x16 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;
    mma_rows<PREC, D>(Ks, Qs, wave * 32, lane, acc);

    // ---- logits (base-2 domain) of this lane: key row r = mt*32 + 8*(e>>2) + rh4 + (e&3).  ONE uniform branch per
    // tile picks the path: bare (no bias window / mask / clamp), table (branch-free), or generic (ragged last tile).
    const int j0 = jt * 128;
    const int k_hmin = j0 / W8, k_hmax = min(j0 + 127, N - 1) / W8;
    const bool need_tab = has_rb || clamp || mr > 0 || (p.pos_tab != nullptr && k_hmax >= q_hmin - R && k_hmin <= q_hmax + R);
    const bool ragged = j0 + 128 > N;
    float tmax = -INFINITY;
    if (ragged) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = mt * 32 + (e & 3) + 8 * (e >> 2) + rh4;
          const unsigned u = min((unsigned)(s_kh[r] + ch), umax), v = min((unsigned)(s_kw[r] + cw), umax);
          float s = __builtin_amdgcn_fmed3f(acc[mt][e], -clipv, clipv) + s_tab[u * TW + v];
          if (has_rb && j0 + r < N) s += rbs * (rbh[s_kh[r]] + rbw[s_kw[r]]);
          if (j0 + r >= N) s = -INFINITY;
          acc[mt][e] = s;
          tmax = fmaxf(tmax, s);
        }
    } else if (need_tab) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = mt * 32 + (e & 3) + 8 * (e >> 2) + rh4;
          const unsigned u = min((unsigned)(s_kh[r] + ch), umax), v = min((unsigned)(s_kw[r] + cw), umax);
          float s = __builtin_amdgcn_fmed3f(acc[mt][e], -clipv, clipv) + s_tab[u * TW + v];
          if (has_rb) s += rbs * (rbh[s_kh[r]] + rbw[s_kw[r]]);
          acc[mt][e] = s;
          tmax = fmaxf(tmax, s);
        }
    } else {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, acc[mt][e]);
    }
    if (!pass1 && DEFER) {
      m_run = fmaxf(m_run, tmax);
    } else if (!pass1) {
      const float m_new = fmaxf(m_run, tmax);
      if (m_new > -INFINITY) {
        float add = 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int e = 0; e < 16; ++e) add += kExact ? exp2f(acc[mt][e] - m_new) : __builtin_amdgcn_exp2f(acc[mt][e] - m_new);
        l_run = l_run * ((m_run > -INFINITY) ? exp2f(m_run - m_new) : 0.f) + add;
        m_run = m_new;
      }
      if (t == ntile - 1) {   // merge the two half-waves (same query, disjoint keys)
        const float m_o = __shfl_xor(m_run, 32), l_o = __shfl_xor(l_run, 32);
        const float m_f = fmaxf(m_run, m_o);
        const float la_ = (m_run > -INFINITY) ? l_run * exp2f(m_run - m_f) : 0.f;
        const float lo_ = (m_o > -INFINITY) ? l_o * exp2f(m_o - m_f) : 0.f;
        m_run = m_f;
        inv_l = 1.f / (la_ + lo_);
      }
    } else if constexpr (kStage) {
      uint16_t* Tw = &Pst[wave * 32 * PLD];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          typedef prob_t pt4 __attribute__((ext_vector_type(4)));
          pt4 h;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float ex = __builtin_amdgcn_exp2f(acc[mt][4 * q + i] - m_run);
            if constexpr (DEFER) { h[i] = (prob_t)ex; l_run += ex; } else h[i] = (prob_t)(ex * inv_l);
          }
          *reinterpret_cast<pt4*>(&Tw[(lane & 31) * PLD + mt * 32 + 8 * q + rh4]) = h;
        }
      // rows of the staged tile -> global: lane = (row within a group of 4, 16-byte chunk of the 256-byte segment)
      const int ch16 = lane & 15, rsub = lane >> 4;
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      prob_t* Pw = reinterpret_cast<prob_t*>(Pout) + ((long)z * N + n0 + wave * 32) * ldp + j0 + ch16 * 8;
      const bool jok = j0 + ch16 * 8 < ldp;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + rsub;
        const u32x4 v = *reinterpret_cast<const u32x4*>(&Tw[row * PLD + ch16 * 8]);
        if (jok && n0 + wave * 32 + row < N) *reinterpret_cast<u32x4*>(Pw + (long)row * ldp) = v;
      }
    } else if (qcol < N) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = j0 + mt * 32 + 8 * q + rh4;
          if (j < ldp) {
            float pv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float d = acc[mt][4 * q + i] - m_run;
              const float ex = kExact ? exp2f(d) : __builtin_amdgcn_exp2f(d);
              if constexpr (DEFER) { pv[i] = ex; l_run += ex; } else pv[i] = ex * inv_l;
            }
            if constexpr (PT == CRAFT_PREC_F32) {
              *reinterpret_cast<float4*>(Prow + j) = make_float4(pv[0], pv[1], pv[2], pv[3]);
            } else {
              typedef prob_t pt4 __attribute__((ext_vector_type(4)));
              pt4 h;
              h[0] = (prob_t)pv[0]; h[1] = (prob_t)pv[1]; h[2] = (prob_t)pv[2]; h[3] = (prob_t)pv[3];
              *reinterpret_cast<pt4*>(Prow + j) = h;
            }
          }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                 // every wave is done with Ks / s_kh / s_kw of tile jt
    T::store(Ks, rk, tid);
    if (tid < 128) { const int j = jn * 128 + tid; s_kh[tid] = j / W8; s_kw[tid] = j - (j / W8) * W8; }
    __syncthreads();
  }
  if constexpr (MODE == 1) {
    const float mm = fmaxf(m_run, __shfl_xor(m_run, 32));       // the two half-waves hold disjoint keys of the same query
    if (lane < 32 && qcol < N) atomicMax(&p.rowmax[(long)z * N + qcol], f2ord(mm));
  }
  if constexpr (MODE == 2) {
    // per-chunk partial sums, added up in a fixed order by k_rowsum_reduce: run-to-run deterministic (no float atomics)
    const float l = l_run + __shfl_xor(l_run, 32);
    if (lane < 32 && qcol < N) p.rowsum[((long)(2 + blockIdx.z) * gridDim.y + z) * N + qcol] = l;
  }
}

// row sums = sum over key chunks of the partial sums (fixed order)
static __global__ void k_rowsum_reduce(float* __restrict__ rs, long n, int nchunk) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = 0.f;
  for (int c = 0; c < nchunk; ++c) a += rs[(2 + c) * n + i];
  rs[i] = a;
}

template <int PREC, int D> static int launch_pt(const ScoreParams& p, void* P, long ldp, int p_prec, dim3 grid, hipStream_t s) {
  if (p.rowsum) {       // deferred normalisation: maxima, then P' + row sums, both over key chunks
    const int nkt = (p.N + 127) / 128;
    dim3 g3(grid.x, grid.y, (nkt + ATTN_KC - 1) / ATTN_KC);
    const long bmn = (long)p.B * p.M * p.N;
    hipError_t me = hipMemsetAsync(p.rowmax, 0, sizeof(unsigned) * (size_t)bmn, s);     // ordered-uint maxima: 0 < everything
    if (me != hipSuccess) return (int)me;
#define GO2(PTV) do { hipLaunchKernelGGL((k_attn_probs<PREC, PTV, D, 1>), g3, dim3(NTHREADS), 0, s, p, P, ldp); \
                      hipLaunchKernelGGL((k_attn_probs<PREC, PTV, D, 2>), g3, dim3(NTHREADS), 0, s, p, P, ldp); } while (0)
    if (p_prec == CRAFT_PREC_F32) GO2(CRAFT_PREC_F32);
    else if (p_prec == CRAFT_PREC_BF16) GO2(CRAFT_PREC_BF16);
    else if (p_prec == CRAFT_PREC_F16) GO2(CRAFT_PREC_F16);
    else return CRAFT_ERR_ARG;
#undef GO2
    hipLaunchKernelGGL(k_rowsum_reduce, dim3((unsigned)((bmn + 255) / 256)), dim3(256), 0, s, p.rowsum, bmn, (int)g3.z);
    return (int)hipGetLastError();
  }
  if (p_prec == CRAFT_PREC_F32) hipLaunchKernelGGL((k_attn_probs<PREC, CRAFT_PREC_F32, D, 0>), grid, dim3(NTHREADS), 0, s, p, P, ldp);
  else if (p_prec == CRAFT_PREC_BF16) hipLaunchKernelGGL((k_attn_probs<PREC, CRAFT_PREC_BF16, D, 0>), grid, dim3(NTHREADS), 0, s, p, P, ldp);
  else if (p_prec == CRAFT_PREC_F16) hipLaunchKernelGGL((k_attn_probs<PREC, CRAFT_PREC_F16, D, 0>), grid, dim3(NTHREADS), 0, s, p, P, ldp);
  else return CRAFT_ERR_ARG;
  return (int)hipGetLastError();
}

template <int D> int launch_attn_probs_d(const ScoreParams& p, void* P, long ldp, int p_prec, int prec, hipStream_t s) {
  dim3 grid((p.N + 127) / 128, p.B * p.M, 1);
  if (prec == CRAFT_PREC_F32) return launch_pt<CRAFT_PREC_F32, D>(p, P, ldp, p_prec, grid, s);
  if (prec == CRAFT_PREC_BF16) return launch_pt<CRAFT_PREC_BF16, D>(p, P, ldp, p_prec, grid, s);
  if (prec == CRAFT_PREC_F16) return launch_pt<CRAFT_PREC_F16, D>(p, P, ldp, p_prec, grid, s);
  if (prec == CRAFT_PREC_F16X3) return launch_pt<CRAFT_PREC_F16X3, D>(p, P, ldp, p_prec, grid, s);
  return CRAFT_ERR_ARG;
}

}  // namespace craft
