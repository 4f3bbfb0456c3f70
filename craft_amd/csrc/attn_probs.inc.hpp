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
        f16x4 h;
        h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
        *reinterpret_cast<f16x4*>(&S[row * LD + ch * 4]) = h;
        if constexpr (PREC == CRAFT_PREC_F16X3) {
          f16x4 l;
          l[0] = (_Float16)(v.x - (float)h[0]); l[1] = (_Float16)(v.y - (float)h[1]);
          l[2] = (_Float16)(v.z - (float)h[2]); l[3] = (_Float16)(v.w - (float)h[3]);
          *reinterpret_cast<f16x4*>(&S[(128 + row) * LD + ch * 4]) = l;
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

// DEFER: the probabilities are stored UN-normalised, P'[i][j] = 2^(t_ij - max_i) in (0, 1], together with the row sums
// l_i (p.rowsum); the consumer (craft_attn_apply) divides its output rows by l_i.  Pass 1 then only needs the row
// maxima -- no exponentials -- which removes ~40 % of the VALU work that bounds this kernel.
template <int PREC, int PT, int D, bool DEFER>
__global__ __launch_bounds__(NTHREADS) void k_attn_probs(ScoreParams p, void* __restrict__ Pout, long ldp) {
  typedef RowTile<PREC, D> T;
  typedef typename T::lds_t lds_t;
  typedef typename ProbT<PT>::t prob_t;
  constexpr bool kExact = (PREC == CRAFT_PREC_F32);
  __shared__ __attribute__((aligned(16))) lds_t Qs[T::ELEMS];
  __shared__ __attribute__((aligned(16))) lds_t Ks[T::ELEMS];
  __shared__ int s_kh[128], s_kw[128];
  __shared__ float s_tab[961];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 128, z = blockIdx.y;
  const int b = z / p.M, m = z - b * p.M;
  const int N = p.N, W8 = p.W8, R = p.R;
  const int qcol = n0 + wave * 32 + (lane & 31);
  const int h1 = qcol / W8, w1 = qcol - h1 * W8;
  const int rh4 = 4 * (lane >> 5);
  const bool clamp = p.clamp_ord != nullptr && ord2f(*p.clamp_ord) > CRAFT_ATTN_CLIP;
  const int nkt = (N + 127) / 128;
  const int q_hmin = n0 / W8, q_hmax = min(n0 + 127, N - 1) / W8;
  // Everything runs in the base-2 log domain: the query tile is pre-multiplied by scale*log2(e) when it is staged
  // (once per block), the positional table by pos_w*log2(e), so a logit costs no multiply and the softmax uses
  // v_exp_f32 (2^x) directly.  P = 2^(t - max) / sum is the same number as exp(s - max) / sum.
  constexpr float LOG2E = 1.4426950408889634f;
  if (p.pos_tab) { const int TT = (2 * R + 1) * (2 * R + 1); for (int i = tid; i < TT; i += NTHREADS) s_tab[i] = p.pos_tab[i] * (p.pos_w * LOG2E); }
  const float clip2 = CRAFT_ATTN_CLIP * LOG2E;

  const float* qbase = p.Q + (long)b * p.q_bs + (long)m * D;
  const float* kbase = p.Kf + (long)b * p.k_bs + (long)m * D;
  prob_t* Prow = reinterpret_cast<prob_t*>(Pout) + ((long)z * N + qcol) * ldp;

  float4 rk[T::NPASS];
  T::fetch(rk, qbase, p.ldq, n0, N, tid);
  T::store(Qs, rk, tid, p.scale * LOG2E);
  T::fetch(rk, kbase, p.ldk, 0, N, tid);
  T::store(Ks, rk, tid);
  if (tid < 128) { s_kh[tid] = tid / W8; s_kw[tid] = tid - (tid / W8) * W8; }
  __syncthreads();

  float m_run = -INFINITY, l_run = 0.f, inv_l = 0.f;
  const int T2 = 2 * nkt;
  for (int t = 0; t < T2; ++t) {
    const bool pass1 = t >= nkt;
    const int jt = pass1 ? t - nkt : t;
    const int jn = (t + 1 < T2) ? ((t + 1 >= nkt) ? t + 1 - nkt : t + 1) : 0;   // (after the last tile: a harmless refetch)
    T::fetch(rk, kbase, p.ldk, jn * 128, N, tid);
    __builtin_amdgcn_sched_barrier(0);     // loads stay above the MFMAs + epilogue they are meant to hide behind

    f32x16 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;
    mma_rows<PREC, D>(Ks, Qs, wave * 32, lane, acc);

    // ---- logits (base-2 domain) of this lane: key row r = mt*32 + 8*(e>>2) + rh4 + (e&3).  ONE uniform branch per
    // tile selects the full path (clamp / positional window / mask / ragged tail) or the bare fast path.
    const int j0 = jt * 128;
    const int k_hmin = j0 / W8, k_hmax = min(j0 + 127, N - 1) / W8;
    const bool has_bias = p.pos_tab != nullptr && k_hmax >= q_hmin - R && k_hmin <= q_hmax + R;
    const bool ragged = j0 + 128 > N;
    float tmax = -INFINITY;
    if (has_bias || ragged || clamp || p.mask_radius > 0) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = mt * 32 + (e & 3) + 8 * (e >> 2) + rh4;
          float s = acc[mt][e];
          if (clamp) s = fminf(fmaxf(s, -clip2), clip2);
          const int dh = s_kh[r] - h1, dw = s_kw[r] - w1;
          if (has_bias && dh >= -R && dh <= R && dw >= -R && dw <= R) s += s_tab[(dh + R) * (2 * R + 1) + dw + R];
          if (p.mask_radius > 0 && max(abs(dh), abs(dw)) > p.mask_radius) s += -1e9f;
          if (ragged && j0 + r >= N) s = -INFINITY;
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
      if (t == nkt - 1) { m_run = fmaxf(m_run, __shfl_xor(m_run, 32)); inv_l = 1.f; l_run = 0.f; }
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
      if (t == nkt - 1) {   // merge the two half-waves (same query, disjoint keys)
        const float m_o = __shfl_xor(m_run, 32), l_o = __shfl_xor(l_run, 32);
        const float m_f = fmaxf(m_run, m_o);
        const float la_ = (m_run > -INFINITY) ? l_run * exp2f(m_run - m_f) : 0.f;
        const float lo_ = (m_o > -INFINITY) ? l_o * exp2f(m_o - m_f) : 0.f;
        m_run = m_f;
        inv_l = 1.f / (la_ + lo_);
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
  if constexpr (DEFER) {
    const float l = l_run + __shfl_xor(l_run, 32);      // the two half-waves hold disjoint keys of the same query
    if (lane < 32 && qcol < N) p.rowsum[(long)z * N + qcol] = l;
  }
}

template <int PREC, int D> static int launch_pt(const ScoreParams& p, void* P, long ldp, int p_prec, dim3 grid, hipStream_t s) {
#define GO(PTV) do { if (p.rowsum) hipLaunchKernelGGL((k_attn_probs<PREC, PTV, D, true>), grid, dim3(NTHREADS), 0, s, p, P, ldp); \
                     else hipLaunchKernelGGL((k_attn_probs<PREC, PTV, D, false>), grid, dim3(NTHREADS), 0, s, p, P, ldp); } while (0)
  if (p_prec == CRAFT_PREC_F32) GO(CRAFT_PREC_F32);
  else if (p_prec == CRAFT_PREC_BF16) GO(CRAFT_PREC_BF16);
  else if (p_prec == CRAFT_PREC_F16) GO(CRAFT_PREC_F16);
  else return CRAFT_ERR_ARG;
#undef GO
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
